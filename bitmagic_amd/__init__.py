"""bitmagic_amd -- MI355X-native bm::bvector<> / bm::aggregator<> hot path.

Python mirror of the reference's operator interface for this path (same names,
argument meaning and error behaviour), layered on the C-ABI of include/bmx.h:

    reference (C++)                                this package
    ---------------------------------------------  -------------------------------------------
    bm::bvector<>                  src/bm.h:113     bvector
      bit_and/bit_or/bit_xor/bit_sub(a, b, opt)     bvector.bit_and(a, b, opt) ...  (static, 3-operand)
      count()                       :2431           bvector.count()
      build_rs_index(&rs)           :2531           bvector.build_rs_index() -> rs_index
      count_to / rank(n, rs)        :3120,1449      bvector.count_to(n, rs) / rank(n, rs)
      select(rank, pos, rs)         :5350           bvector.select(rank, rs) -> (found, pos)
    bm::bit_import_u32             src/bmbvimport.h bit_import_u32(ctx, words, optimize)
    bm::count_and/or/xor/sub       src/bmalgo.h     count_and(a, b) ...
    bm::aggregator<bvector<>>      src/bmaggregator.h:120
      add / reset / combine_or / combine_and /      aggregator.add / reset / combine_or /
      combine_and_sub                               combine_and / combine_and_sub
      pipeline<agg_opt_only_counts> :222            aggregator.pipeline (add().add(bv, grp), complete())
      combine_and_sub(pipe)         :1292           aggregator.combine_and_sub(pipe)

The data path is the HIP library only; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

from . import _ffi
from ._ffi import BmxError, check, lib  # noqa: F401
from .sharding import shard_range, allreduce_counts  # noqa: F401

NULL, FULL, BIT, GAP = 0, 1, 2, 3
AND, OR, XOR, SUB = 0, 1, 2, 3
BLOCK_WORDS, BLOCK_BITS = 2048, 65536
opt_none, opt_compress = 0, 3        # bvector::optmode (src/bm.h:129-135)
ID_MAX = 0xFFFFFFFF                  # bm::id_max (src/bmconst.h:109)
ID_MAX64 = 0xFFFFFFFFFFFFFFFF

__all__ = ["context", "bvector", "aggregator", "slice_scanner", "rs_index", "group", "gbvector", "gaggregator", "gpipeline", "bit_import_u32", "count_and", "count_or",
           "count_xor", "count_sub", "BmxError", "simd_version", "device_count", "agg_run_options",
           "agg_opt_only_counts", "agg_opt_bvect_and_counts", "agg_opt_disable_bvects_and_counts"]


def simd_version() -> int:
    return lib().bmx_simd_version()


def device_count() -> int:
    n = C.c_int()
    check(lib().bmx_device_count(C.byref(n)))
    return n.value


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class context:
    """One device + one HIP stream (bmx_ctx).  `stream` may be a raw hipStream_t
    (e.g. torch.cuda.current_stream().cuda_stream) so torch events see the kernels."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        check(lib().bmx_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().bmx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().bmx_ctx_synchronize(self._h))

    def set_tuning(self, key: str, value: int):
        check(lib().bmx_ctx_set_tuning(self._h, key.encode(), int(value)))

    def trim(self):
        check(lib().bmx_ctx_trim(self._h))

    # -- debug aids (include/bmx.h "debug aids") --
    def redzone_check(self) -> dict:
        """verify the red zones of every live device allocation (contexts created under BMX_DEBUG_REDZONE=1)"""
        en, hits = C.c_int32(), C.c_uint64()
        buf = C.create_string_buffer(16384)
        check(lib().bmx_debug_redzone_check(self._h, C.byref(en), C.byref(hits), buf, len(buf)))
        return {"enabled": bool(en.value), "hits": hits.value, "report": buf.value.decode(errors="replace")}

    def inject_failure(self, kind: int, after: int = 0) -> None:
        """fault injection for tests: 1 / 2 / 3 = the library entry `after` calls from now throws bad_alloc / length_error / a non-std
        exception, 4 = the device allocation `after` allocations from now fails, 5 = red-zone self-test, 0 = disarm"""
        check(lib().bmx_debug_inject_failure(self._h, int(kind), int(after)))

    # -- packed collections (include/bmx.h "packed collections"): operand sets the engine keeps column-major --
    def collection_prepare(self, vecs, role: int = 1) -> None:
        """build the packed collection of an operand list now; role: ROLE_AND (0), ROLE_OR (1), ROLE_SUB (2)"""
        vecs = list(vecs)
        check(lib().bmx_collection_prepare(self._h, _handles(vecs), len(vecs), int(role)))

    def pack_stats(self) -> dict:
        n, b, ms = C.c_uint32(), C.c_uint64(), C.c_float()
        check(lib().bmx_ctx_pack_stats(self._h, C.byref(n), C.byref(b), C.byref(ms)))
        rb = C.c_uint64()
        check(lib().bmx_ctx_pack_run_bytes(self._h, C.byref(rb)))
        return {"collections": n.value, "bytes": b.value, "run_bytes": rb.value, "last_build_ms": ms.value}

    def mem_used(self) -> int:
        b = C.c_uint64()
        check(lib().bmx_ctx_mem_used(self._h, C.byref(b)))
        return b.value

    def timer_start(self):
        check(lib().bmx_timer_start(self._h))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        check(lib().bmx_timer_stop_ms(self._h, C.byref(ms)))
        return ms.value


class bvector:
    """Device-resident, immutable bit-vector (block table + slabs in HBM)."""

    def __init__(self, ctx: context, handle: C.c_void_p):
        self.ctx, self._h = ctx, handle

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib().bmx_vec_free(self.ctx._h, self._h)
            self._h = None
        except Exception:
            pass

    # ---- construction -----------------------------------------------------
    @staticmethod
    def from_block_table(ctx: context, nbits: int, kinds, offs, bit_slab, gap_slab) -> "bvector":
        """bmx_vec_upload: the flattened walk of blocks_manager::top_blocks_root()."""
        kinds = np.ascontiguousarray(kinds, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint32)
        bit_slab = np.ascontiguousarray(bit_slab, np.uint32)
        gap_slab = np.ascontiguousarray(gap_slab, np.uint16)
        h = C.c_void_p()
        check(lib().bmx_vec_upload(ctx._h, nbits, kinds.size, _ptr(kinds), _ptr(offs),
                                   _ptr(bit_slab) if bit_slab.size else None, bit_slab.size // BLOCK_WORDS,
                                   _ptr(gap_slab) if gap_slab.size else None, gap_slab.size, C.byref(h)))
        return bvector(ctx, h)

    @staticmethod
    def generate(ctx: context, seed: int, vec_id: int, density_q16: int, nbits: int,
                 with_common: bool = False, optimize: bool = True, block_range=None) -> "bvector":
        """block_range=(nb_from, nb_to): only that block range of the logical vector (a multi-GPU shard)"""
        h = C.c_void_p()
        nb_from, nb_to = block_range if block_range is not None else (0, ID_MAX)
        check(lib().bmx_vec_generate_shard(ctx._h, seed, vec_id, int(with_common), density_q16, nbits,
                                           nb_from, nb_to, int(optimize), C.byref(h)))
        return bvector(ctx, h)

    # ---- inspection -------------------------------------------------------
    def info(self):
        nbits, nblocks, slab, gw = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        counts = (C.c_uint32 * 4)()
        check(lib().bmx_vec_info(self._h, C.byref(nbits), C.byref(nblocks), counts, C.byref(slab), C.byref(gw)))
        return {"nbits": nbits.value, "nblocks": nblocks.value, "counts": list(counts),
                "bit_slab_blocks": slab.value, "gap_words": gw.value}

    def to_indices(self, width: int = 8) -> np.ndarray:
        """the sorted positions of the set bits (device compaction; include/bmx.h bmx_vec_to_indices)"""
        n = C.c_uint64()
        cnt = self.count()
        out = np.zeros(cnt, np.uint64 if width == 8 else np.uint32)
        check(lib().bmx_vec_to_indices(self.ctx._h, self._h, width, _ptr(out) if cnt else None, cnt, C.byref(n)))
        return out[:n.value]

    def operand_bytes(self) -> int:
        """algorithmic bytes of this operand: 8,192 B per bit-block, 2 x (len + 1) B per GAP block"""
        b = C.c_uint64()
        check(lib().bmx_vec_operand_bytes(self.ctx._h, self._h, C.byref(b)))
        return b.value

    def size(self) -> int:
        return self.info()["nbits"]

    def calc_stat(self):
        """bvector::calc_stat (src/bm.h:4010): bit_blocks / gap_blocks."""
        i = self.info()
        return {"bit_blocks": i["counts"][BIT], "gap_blocks": i["counts"][GAP],
                "full_blocks": i["counts"][FULL], "null_blocks": i["counts"][NULL]}

    def block_table(self):
        """-> kinds, offs, bit_slab, gap_slab (bmx_vec_download)"""
        i = self.info()
        kinds = np.zeros(i["nblocks"], np.uint8)
        offs = np.zeros(i["nblocks"], np.uint32)
        bit_slab = np.zeros(i["bit_slab_blocks"] * BLOCK_WORDS, np.uint32)
        gap_slab = np.zeros(i["gap_words"], np.uint16)
        check(lib().bmx_vec_download(self.ctx._h, self._h, _ptr(kinds), _ptr(offs),
                                     _ptr(bit_slab) if bit_slab.size else None,
                                     _ptr(gap_slab) if gap_slab.size else None))
        return kinds, offs, bit_slab, gap_slab

    def to_words(self, nwords: int | None = None) -> np.ndarray:
        if nwords is None:
            nwords = self.info()["nblocks"] * BLOCK_WORDS
        out = np.zeros(nwords, np.uint32)
        if nwords:
            check(lib().bmx_vec_to_words(self.ctx._h, self._h, _ptr(out), nwords))
        return out

    # ---- set algebra (3-operand forms, src/bm.h:6185,5973,6072,6403) -------
    @staticmethod
    def _op2(op, a: "bvector", b: "bvector", opt_mode: int) -> "bvector":
        h = C.c_void_p()
        check(lib().bmx_op2(a.ctx._h, op, a._h, b._h, int(opt_mode == opt_compress), C.byref(h)))
        return bvector(a.ctx, h)

    @staticmethod
    def op2_count(op, a: "bvector", b: "bvector", opt_mode: int = opt_none, want_result: bool = True):
        """bmx_op2_count: the three-operand operation and the count of its result in one call -> (vector | None, count)"""
        h, c = C.c_void_p(), C.c_uint64()
        check(lib().bmx_op2_count(a.ctx._h, op, a._h, b._h, int(opt_mode == opt_compress), C.byref(h) if want_result else None, C.byref(c)))
        return (bvector(a.ctx, h) if want_result else None), c.value

    @staticmethod
    def op2_async(op, a, b) -> "pending":
        """bmx_op2_dev: the three-operand operation (opt_none) enqueued on the context's stream; a / b are vectors (any block
        kinds) or unresolved results of earlier op2_async calls; -> pending (wait() gives the vector)"""
        ctx = a.ctx
        h = C.c_void_p()
        av, ap = (a._h, None) if isinstance(a, bvector) else (None, a._h)
        bv, bp = (b._h, None) if isinstance(b, bvector) else (None, b._h)
        check(lib().bmx_op2_dev(ctx._h, op, av, ap, bv, bp, C.byref(h)))
        return pending(ctx, h)

    @staticmethod
    def bit_and(a, b, opt_mode=opt_none):
        return bvector._op2(AND, a, b, opt_mode)

    @staticmethod
    def bit_or(a, b, opt_mode=opt_none):
        return bvector._op2(OR, a, b, opt_mode)

    @staticmethod
    def bit_xor(a, b, opt_mode=opt_none):
        return bvector._op2(XOR, a, b, opt_mode)

    @staticmethod
    def bit_sub(a, b, opt_mode=opt_none):
        return bvector._op2(SUB, a, b, opt_mode)

    def count(self) -> int:
        c = C.c_uint64()
        check(lib().bmx_count(self.ctx._h, self._h, C.byref(c)))
        return c.value

    def find(self):
        """-> (found, pos) of the first set bit  (bvector::find, src/bm.h:1593)"""
        found, idx = C.c_int(), C.c_uint64()
        check(lib().bmx_find_first_and_sub(self.ctx._h, _handles([self]), 1, None, 0, C.byref(found), C.byref(idx)))
        return bool(found.value), int(idx.value)

    def any(self) -> bool:
        return self.find()[0]

    # ---- rank / select ----------------------------------------------------
    def build_rs_index(self) -> "rs_index":
        h = C.c_void_p()
        check(lib().bmx_rs_build(self.ctx._h, self._h, C.byref(h)))
        return rs_index(self, h)

    def count_to(self, n, rs: "rs_index"):
        """ones in [0..n] inclusive; scalar or array of positions"""
        arr = np.ascontiguousarray(np.atleast_1d(n), np.uint64)
        out = np.zeros(arr.shape, np.uint64)
        if arr.size:
            check(lib().bmx_rank_batch(self.ctx._h, self._h, rs._h, _ptr(arr), arr.size, _ptr(out)))
        return int(out[0]) if np.isscalar(n) else out

    rank = count_to

    def _bit_and_rank(self, n, rs):
        """(bit(n), rank(n)) for an array of positions: bit(n) = rank(n) - rank(n-1)"""
        n = np.ascontiguousarray(np.atleast_1d(n), np.uint64)
        prev = np.where(n > 0, n - np.uint64(1), np.uint64(0))
        r = self.count_to(np.concatenate([n, prev]), rs)
        rn, rp = r[:n.size], np.where(n > 0, r[n.size:], np.uint64(0))
        return (rn - rp), rn

    def rank_corrected(self, n, rs: "rs_index"):
        """rank(n) - bit(n)  (src/bm.h:3229)"""
        bit, rn = self._bit_and_rank(n, rs)
        out = rn - bit
        return int(out[0]) if np.isscalar(n) else out

    def count_to_test(self, n, rs: "rs_index"):
        """bit(n) ? rank(n) : 0  (src/bm.h:3173)"""
        bit, rn = self._bit_and_rank(n, rs)
        out = np.where(bit != 0, rn, np.uint64(0))
        return int(out[0]) if np.isscalar(n) else out

    def count_range(self, left, right, rs: "rs_index"):
        """ones in [left..right]; arguments are swapped when left > right  (src/bm.h:3548)"""
        l = np.ascontiguousarray(np.atleast_1d(left), np.uint64); r = np.ascontiguousarray(np.atleast_1d(right), np.uint64)
        lo, hi = np.minimum(l, r), np.maximum(l, r)
        prev = np.where(lo > 0, lo - np.uint64(1), np.uint64(0))
        q = self.count_to(np.concatenate([hi, prev]), rs)
        out = q[:hi.size] - np.where(lo > 0, q[hi.size:], np.uint64(0))
        return int(out[0]) if np.isscalar(left) else out

    def find_rank(self, rank, from_pos, rs: "rs_index"):
        """position of the rank-th set bit at or after from_pos  (src/bm.h:5279) -> (found, pos)"""
        rk = np.ascontiguousarray(np.atleast_1d(rank), np.uint64); fr = np.ascontiguousarray(np.atleast_1d(from_pos), np.uint64)
        before = np.where(fr > 0, self.count_to(np.where(fr > 0, fr - np.uint64(1), np.uint64(0)), rs), np.uint64(0))
        found, pos = self.select(np.where(rk > 0, rk + before, np.uint64(0)), rs)
        if np.isscalar(rank):
            return bool(found[0]), int(pos[0])
        return found, pos

    def select(self, rank, rs: "rs_index"):
        """-> (found, pos); rank is 1-based (src/bm.h:5350)"""
        arr = np.ascontiguousarray(np.atleast_1d(rank), np.uint64)
        pos = np.zeros(arr.shape, np.uint64)
        found = np.zeros(arr.shape, np.uint8)
        if arr.size:
            check(lib().bmx_select_batch(self.ctx._h, self._h, rs._h, _ptr(arr), arr.size, _ptr(pos), _ptr(found)))
        if np.isscalar(rank):
            return bool(found[0]), int(pos[0])
        return found.astype(bool), pos


class rs_index:
    """bm::rs_index (src/bmrs.h:39): built on the device, exportable in the reference's layout."""

    def __init__(self, bv: bvector, handle):
        self.bv, self.ctx, self._h = bv, bv.ctx, handle

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib().bmx_rs_free(self.ctx._h, self._h)
            self._h = None
        except Exception:
            pass

    def count(self) -> int:
        c = C.c_uint64()
        check(lib().bmx_rs_count(self._h, C.byref(c)))
        return c.value

    def info(self) -> dict:
        """device bytes of the index and whether it holds rank lines (memory policy: tuning key rs_lines)"""
        b, h = C.c_uint64(), C.c_int32()
        check(lib().bmx_rs_info(self._h, C.byref(b), C.byref(h)))
        sb, sbits = C.c_uint64(), C.c_int32()
        check(lib().bmx_rs_select_format(self._h, C.byref(sbits), C.byref(sb)))
        return {"bytes": b.value, "has_lines": bool(h.value), "select_offset_bits": sbits.value, "select_lines_bytes": sb.value}

    def export(self):
        """-> bcount[nb], sub_count[nb] (first | second<<16 | aux0<<32 | aux1<<48)"""
        n = self.bv.info()["nblocks"]
        bc = np.zeros(n, np.uint32)
        sub = np.zeros(n, np.uint64)
        if n:
            check(lib().bmx_rs_export(self.ctx._h, self._h, _ptr(bc), _ptr(sub)))
        return bc, sub


def bit_import_u32(ctx: context, words, optimize: bool = True) -> bvector:
    """bm::bit_import_u32(bv, words, nwords, optimize)  src/bmbvimport.h:46"""
    words = np.ascontiguousarray(words, np.uint32)
    h = C.c_void_p()
    check(lib().bmx_vec_import_bits(ctx._h, _ptr(words) if words.size else None, words.size, int(optimize), C.byref(h)))
    return bvector(ctx, h)


class pending:
    """an asynchronous result (bmx_op2_dev) that has not been resolved into a vector yet: usable as an operand of further
    bvector.op2_async calls, and nowhere else"""

    def __init__(self, ctx: context, h):
        self.ctx, self._h = ctx, h

    def wait(self) -> bvector:
        """bmx_pending_wait: waits for this result, -> the vector (the handle is consumed)"""
        h = C.c_void_p()
        ph, self._h = self._h, None
        check(lib().bmx_pending_wait(self.ctx._h, ph, C.byref(h)))
        return bvector(self.ctx, h)

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib().bmx_pending_free(self.ctx._h, self._h)
        except Exception:
            pass
        self._h = None


def _count_op2(op, a: bvector, b: bvector) -> int:
    c = C.c_uint64()
    check(lib().bmx_count_op2(a.ctx._h, op, a._h, b._h, C.byref(c)))
    return c.value


def count_and(a, b): return _count_op2(AND, a, b)      # src/bmalgo.h:49
def count_or(a, b): return _count_op2(OR, a, b)        # :149
def count_xor(a, b): return _count_op2(XOR, a, b)      # :81
def count_sub(a, b): return _count_op2(SUB, a, b)      # :115


def _handles(vecs: Sequence[bvector]):
    arr = (C.c_void_p * max(len(vecs), 1))()
    for i, v in enumerate(vecs):
        arr[i] = v._h
    return arr


class arg_groups:
    """aggregator::arg_groups (src/bmaggregator.h:2925): group 0 = AND, group 1 = SUB."""

    def __init__(self):
        self.arg_bv0: list[bvector] = []
        self.arg_bv1: list[bvector] = []

    def add(self, bv: bvector | None, agr_group: int = 0) -> int:
        if agr_group > 1:
            raise BmxError(_ffi.ERR_RANGE, "BMX-03: Incorrect range or index", "agr_group > 1")   # BM_ERR_RANGE :2934
        if bv is None:                       # ignored, :2939-2947
            return 0
        (self.arg_bv1 if agr_group else self.arg_bv0).append(bv)
        return len(self.arg_bv1 if agr_group else self.arg_bv0)

    def reset(self):
        self.arg_bv0.clear()
        self.arg_bv1.clear()


class agg_run_options:
    """bm::agg_run_options<OBvects, OCounts> (src/bmaggregator.h:62-103)"""

    def __init__(self, make_results: bool = True, compute_counts: bool = False, search_masks: bool = False):
        self.make_results, self.compute_counts, self.search_masks = make_results, compute_counts, search_masks

    def is_make_results(self): return self.make_results
    def is_compute_counts(self): return self.compute_counts
    def is_masks(self): return self.search_masks                          # :78: the pipeline honours set_range_hint


agg_opt_disable_bvects_and_counts = agg_run_options(False, False)    # :84
agg_opt_only_counts = agg_run_options(False, True)                   # :92
agg_opt_bvect_and_counts = agg_run_options(True, True)               # :100


class pipeline:
    """aggregator::pipeline<Opt> (src/bmaggregator.h:222-341); default Opt = agg_opt_only_counts here
    (the headline path); pass agg_run_options() for the reference's default (result vectors, no counts)."""

    def __init__(self, ctx: context, opt: agg_run_options = agg_opt_only_counts):
        self.ctx = ctx
        self.opt = opt
        self.groups: list[arg_groups] = []
        self._h = None
        self._counts: np.ndarray | None = None
        self._results: list = []
        self._or_target: bvector | None = None
        self._want_or_target = False
        self.search_count_limit = ID_MAX

    def set_or_target(self, bv_or: "bvector | None" = None):
        """pipeline::set_or_target (:245): all group results are OR-ed into the target; the (immutable)
        device vector passed here is the initial content, the updated one is returned by get_or_target()"""
        self._want_or_target = True
        self._or_target = bv_or

    def get_or_target(self) -> "bvector | None":
        return self._or_target

    def set_search_count_limit(self, limit: int):
        """(:255, honoured at :1365) a group needs no more than `limit` hits: "can find more, cannot find less"; the counts
        run walks the block columns in ascending launch windows and stops launching once every group has enough"""
        self.search_count_limit = limit
        if self._h:
            # (bm::id_max = "no limit", as at construction: the library maps 0 / id_max / the 48-bit id_max to one plain run)
            check(lib().bmx_pipeline_set_search_count_limit(self.ctx._h, self._h, 0 if limit in (ID_MAX, ID_MAX64) else min(int(limit), ID_MAX64)))

    def last_windows(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().bmx_pipeline_last_windows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_window_groups(self) -> list:
        """arg-groups every launched window of the last counts run under a search limit ran over"""
        n = C.c_uint32()
        out = (C.c_uint32 * 64)()
        check(lib().bmx_pipeline_last_window_groups(self._h, out, 64, C.byref(n)))
        return [int(out[i]) for i in range(min(n.value, 64))]

    def get_bv_res_vector(self) -> list:
        return self._results

    def add(self) -> arg_groups:
        if self._h:
            raise RuntimeError("pipeline already complete()")
        g = arg_groups()
        self.groups.append(g)
        return g

    def size(self) -> int:
        return len(self.groups)

    def complete(self):
        and_list = [v for g in self.groups for v in g.arg_bv0]
        sub_list = [v for g in self.groups for v in g.arg_bv1]
        and_n = (C.c_uint32 * max(len(self.groups), 1))(*[len(g.arg_bv0) for g in self.groups])
        sub_n = (C.c_uint32 * max(len(self.groups), 1))(*[len(g.arg_bv1) for g in self.groups])
        h = C.c_void_p()
        check(lib().bmx_pipeline_create(self.ctx._h, _handles(and_list), and_n, _handles(sub_list), sub_n,
                                        len(self.groups), C.byref(h)))
        self._h = h
        if self.search_count_limit not in (ID_MAX, ID_MAX64):
            check(lib().bmx_pipeline_set_search_count_limit(self.ctx._h, self._h, int(self.search_count_limit)))

    def is_complete(self) -> bool:
        return self._h is not None

    def get_bv_count_vector(self) -> np.ndarray:
        return self._counts

    def operand_bytes(self, nb_from: int = 0, nb_to: int = ID_MAX) -> int:
        b = C.c_uint64()
        check(lib().bmx_pipeline_operand_bytes(self.ctx._h, self._h, nb_from, nb_to, C.byref(b)))
        return b.value

    def describe(self, nb_from: int = 0, nb_to: int = ID_MAX) -> str:
        """kernel + launch plan a counts run over the block range takes"""
        buf = C.create_string_buffer(256)
        check(lib().bmx_pipeline_describe(self.ctx._h, self._h, nb_from, nb_to, buf, 256, None))
        return buf.value.decode()

    def launches(self, nb_from: int = 0, nb_to: int = ID_MAX) -> int:
        buf = C.create_string_buffer(256); n = C.c_uint32()
        check(lib().bmx_pipeline_describe(self.ctx._h, self._h, nb_from, nb_to, buf, 256, C.byref(n)))
        return n.value

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib().bmx_pipeline_destroy(self.ctx._h, self._h)
            self._h = None
        except Exception:
            pass


class aggregator:
    """bm::aggregator<bvector<>> (src/bmaggregator.h:120)."""

    pipeline = pipeline

    def __init__(self, ctx: context):
        self.ctx = ctx
        self.ag = arg_groups()
        self.opt_mode = False            # opt_none, :917
        self.compute_count = False
        self._count = 0
        self._range = None               # set_range_hint, :837-839

    def set_range_hint(self, frm: int, to: int) -> bool:                 # :481,974
        """where results need to be searched: find_first_and_sub visits the block columns of the range only (a
        one-block range is also bit-masked); combine_and_sub(pipe) honours it when the pipeline options enable
        search masks.  -> True if the range is one-block bound"""
        self._range = (int(frm), int(to))
        return (int(frm) >> 16) == (int(to) >> 16)

    def reset_range_hint(self):                                          # :486,962
        self._range = None

    def set_optimization(self, opt: bool = True):                        # :359
        self.opt_mode = bool(opt)

    def set_compute_count(self, count_mode: bool):                       # :363
        self.compute_count = bool(count_mode)
        self._count = 0

    def count(self) -> int:                                              # :488
        return self._count

    def add(self, bv: bvector | None, agr_group: int = 0) -> int:       # :1013
        return self.ag.add(bv, agr_group)

    def reset(self):                                                     # :941
        self.ag.reset()

    def combine_or(self, bv_src: Iterable[bvector] | None = None) -> bvector:         # :1021 / :1101
        src = list(bv_src) if bv_src is not None else list(self.ag.arg_bv0)
        self.ag.reset()              # the reference clears the member arg-groups here (src/bmaggregator.h:1110)
        h = C.c_void_p()
        check(lib().bmx_agg_or_opt(self.ctx._h, _handles(src), len(src), int(self.opt_mode), C.byref(h)))
        return bvector(self.ctx, h)

    def combine_and(self, bv_src: Iterable[bvector] | None = None) -> bvector:        # :1030 == combine_and_sub w/o SUB
        src = list(bv_src) if bv_src is not None else self.ag.arg_bv0
        return self.combine_and_sub(src, [])[0]

    def combine_and_sub(self, bv_src_and=None, bv_src_sub=None):                         # :1044 / :1162 / :1292
        """-> (target, any)  |  with a pipeline argument: runs it according to its options; results in
        pipe.get_bv_res_vector() / get_bv_count_vector() / get_or_target()"""
        if isinstance(bv_src_and, pipeline):
            pipe = bv_src_and
            if pipe.opt.is_make_results() or pipe._want_or_target:
                return self._run_pipeline_results(pipe)
            if pipe.opt.is_compute_counts():
                return self._run_pipeline(pipe)
            return None
        a = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        s = list(bv_src_sub) if bv_src_sub is not None else self.ag.arg_bv1
        h = C.c_void_p()
        any_ = C.c_int()
        check(lib().bmx_agg_and_sub(self.ctx._h, _handles(a), len(a), _handles(s), len(s), C.byref(h), C.byref(any_)))
        return bvector(self.ctx, h), bool(any_.value)

    def combine_and_sub_bi(self, bv_src_and=None, bv_src_sub=None, width: int = 8) -> np.ndarray:       # :450,533,1068,1226
        """the AND-SUB result as SORTED positions (what the reference feeds into a back-insert iterator)"""
        a = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        s = list(bv_src_sub) if bv_src_sub is not None else self.ag.arg_bv1
        t, any_ = self.combine_and_sub(a, s)
        return t.to_indices(width) if any_ else np.zeros(0, np.uint64 if width == 8 else np.uint32)

    def combine_shift_right_and(self, bv_src_and=None, any: bool = False):               # :473,1089 / :552,2494
        """-> (target, found).  T_0 = src[0], T_k = (T_{k-1} >> 1) & src[k]; target stored with the
        aggregator's optimisation mode.  Under set_compute_count(True) nothing is stored (:2593):
        -> (None, found) and count() holds the population."""
        src = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        self._count = 0
        if self.compute_count:
            c = C.c_uint64()
            check(lib().bmx_agg_shift_right_and_count(self.ctx._h, _handles(src), len(src), C.byref(c)))
            self._count = int(c.value)
            return None, self._count != 0
        h, found = C.c_void_p(), C.c_int()
        check(lib().bmx_agg_shift_right_and(self.ctx._h, _handles(src), len(src), int(self.opt_mode), int(any),
                                            C.byref(h), C.byref(found)))
        return bvector(self.ctx, h), bool(found.value)

    def find_first_and_sub(self, bv_src_and=None, bv_src_sub=None):                      # :1079 / :1458
        """-> (found, idx): first set bit of AND(group 0) AND NOT OR(group 1)"""
        a = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        s = list(bv_src_sub) if bv_src_sub is not None else self.ag.arg_bv1
        found, idx = C.c_int(), C.c_uint64()
        if self._range is not None:
            check(lib().bmx_find_first_and_sub_range(self.ctx._h, _handles(a), len(a), _handles(s), len(s),
                                                     self._range[0], self._range[1], C.byref(found), C.byref(idx)))
        else:
            check(lib().bmx_find_first_and_sub(self.ctx._h, _handles(a), len(a), _handles(s), len(s), C.byref(found), C.byref(idx)))
        return bool(found.value), int(idx.value)

    def _hint_blocks(self, pipe: pipeline):
        if self._range is not None and pipe.opt.is_masks():
            return self._range[0] >> 16, (self._range[1] >> 16) + 1
        return 0, ID_MAX

    def _run_pipeline(self, pipe: pipeline, nb_from: int | None = None, nb_to: int | None = None):
        if not pipe.is_complete():
            raise RuntimeError("pipeline is not complete()")
        out = np.zeros(max(pipe.size(), 1), np.uint64)
        if nb_from is None and self._range is not None and pipe.opt.is_masks():
            # set_range_hint: block columns of the hint; a one-block hint is bit-masked as well (:980-988)
            check(lib().bmx_pipeline_run_results_hint(self.ctx._h, pipe._h, self._range[0], self._range[1], None,
                                                      out.ctypes.data_as(C.POINTER(C.c_uint64)), None, None))
            pipe._counts = out[:pipe.size()]
            return pipe._counts
        if nb_from is None:
            nb_from, nb_to = 0, ID_MAX
        check(lib().bmx_pipeline_run_counts(self.ctx._h, pipe._h, nb_from, nb_to,
                                            out.ctypes.data_as(C.POINTER(C.c_uint64))))
        pipe._counts = out[:pipe.size()]
        return pipe._counts

    def _run_pipeline_results(self, pipe: pipeline):
        if not pipe.is_complete():
            raise RuntimeError("pipeline is not complete()")
        n = pipe.size()
        res = (C.c_void_p * max(n, 1))()
        cnt = np.zeros(max(n, 1), np.uint64)
        ort = C.c_void_p()
        want_res = pipe.opt.is_make_results()
        want_cnt = pipe.opt.is_compute_counts()
        cnt_p = cnt.ctypes.data_as(C.POINTER(C.c_uint64)) if (want_cnt and want_res) else None
        ort_in = pipe._or_target._h if (pipe._want_or_target and pipe._or_target is not None) else None
        if self._range is not None and pipe.opt.is_masks():
            check(lib().bmx_pipeline_run_results_hint(self.ctx._h, pipe._h, self._range[0], self._range[1], res if want_res else None,
                                                      cnt_p, ort_in, C.byref(ort) if pipe._want_or_target else None))
        else:
            check(lib().bmx_pipeline_run_results_range(self.ctx._h, pipe._h, 0, ID_MAX, res if want_res else None, cnt_p, ort_in,
                                                       C.byref(ort) if pipe._want_or_target else None))
        pipe._results = [bvector(self.ctx, C.c_void_p(res[g])) if (want_res and res[g]) else None for g in range(n)]
        if pipe._want_or_target:
            pipe._or_target = bvector(self.ctx, ort)
        if want_cnt:
            pipe._counts = cnt[:n] if want_res else self._run_pipeline(pipe)
        return pipe._results

    def run_counts_dev(self, pipe: pipeline, d_counts_ptr: int, nb_from: int = 0, nb_to: int = ID_MAX):
        """asynchronous run; counts land in device memory (e.g. a torch tensor's data_ptr())"""
        check(lib().bmx_pipeline_run_counts_dev(self.ctx._h, pipe._h, nb_from, nb_to, C.c_void_p(d_counts_ptr)))


CMP_GT, CMP_GE, CMP_LT, CMP_LE, CMP_RANGE, CMP_EQ, CMP_ZERO, CMP_NONZERO = range(8)


class slice_scanner:
    """Bit-sliced search over device-resident slices: the aggregator call pattern of
    bm::sparse_vector_scanner<SV> (src/bmsparsevec_algo.h): find_eq :1083,2387 (group rule
    prepare_and_sub_aggregator :2593-2640), find_gt/ge/lt/le/find_range :1135-1174, find_zero :2290,
    find_nonzero :4464.  slices[i] holds bit i of every element (None = plane absent); len(slices) plays
    effective_slices(); size = sv.size() (rows; default: the longest slice); not_null = sv.get_null_bvector().
    A batch of equality searches is one counts-only pipeline launch; a comparison search is one pass over the planes."""

    def __init__(self, ctx: context, slices: Sequence[bvector | None], size: int | None = None, not_null: bvector | None = None,
                 signed: bool = False):
        """signed: the planes of a bm::sparse_vector<int, ..> -- slices[0] is the sign, slices[1..] the magnitude
        (base_sparse_vector::s2u, src/bmbmatrix.h:2536): the comparison searches then take signed bounds
        (find_gt_horizontal_s, src/bmsparsevec_algo.h:1484,3033)"""
        self.ctx, self.slices = ctx, list(slices)
        self.agg = aggregator(ctx)
        self.not_null = not_null
        self._size = size
        self.signed = bool(signed)

    def size(self) -> int:
        if self._size is None:
            self._size = max([p.size() for p in self.slices if p is not None], default=0)
        return self._size

    def _new_pipeline(self):
        return pipeline(self.ctx)

    def _can_transpose(self) -> bool:
        return len(self.slices) <= 32

    def _eq_counts_call(self, arr, vals, out):
        check(lib().bmx_slice_eq_counts(self.ctx._h, arr, len(self.slices), _ptr(vals), vals.size, self.size(),
                                        self.not_null._h if self.not_null is not None else None, _ptr(out)))

    def _groups(self, value: int):
        if value <= 0:
            raise BmxError(_ffi.ERR_BADARG, "Invalid argument", "value 0 has no AND group: find_eq(0) goes through the comparison kernel")
        a = []
        for bit in range(value.bit_length() - 1, -1, -1):               # backward order (:2614)
            if (value >> bit) & 1:
                if bit >= len(self.slices) or self.slices[bit] is None:
                    return None                                          # a set bit without a plane: nothing matches (:2621)
                a.append(self.slices[bit])
        s = [p for i, p in enumerate(self.slices) if p is not None and not (value >> i) & 1]
        return a, s

    def _compare(self, pred: int, v0: int = 0, v1: int = 0, count_only: bool = False):
        arr = (C.c_void_p * max(len(self.slices), 1))()
        for i, p in enumerate(self.slices):
            arr[i] = p._h if p is not None else None
        h, cnt = C.c_void_p(), C.c_uint64()
        nn = self.not_null._h if self.not_null is not None else None
        if self.signed:
            if not (-(1 << 63) <= v0 < (1 << 63) and -(1 << 63) <= v1 < (1 << 63)):
                raise BmxError(_ffi.ERR_RANGE, "Incorrect range or index", "signed 64-bit values only")
            check(lib().bmx_slice_compare_signed(self.ctx._h, arr, len(self.slices), pred, v0, v1, self.size(), nn,
                                                 None if count_only else C.byref(h), C.byref(cnt)))
            return cnt.value if count_only else bvector(self.ctx, h)
        if v0 < 0 or v1 < 0 or v0 >= 1 << 64 or v1 >= 1 << 64:
            raise BmxError(_ffi.ERR_RANGE, "Incorrect range or index", "unsigned 64-bit values only")
        check(lib().bmx_slice_compare(self.ctx._h, arr, len(self.slices), pred, v0, v1, self.size(), nn,
                                      None if count_only else C.byref(h), C.byref(cnt)))
        return cnt.value if count_only else bvector(self.ctx, h)

    def compare_stat(self, pred: int, v0: int = 0, v1: int = 0):
        """-> (count, plane_bytes): the count of a comparison search and the plane bytes its walk had to read"""
        arr = (C.c_void_p * max(len(self.slices), 1))()
        for i, p in enumerate(self.slices):
            arr[i] = p._h if p is not None else None
        cnt, pb = C.c_uint64(), C.c_uint64()
        check(lib().bmx_slice_compare_stat(self.ctx._h, arr, len(self.slices), pred, v0, v1, self.size(),
                                           self.not_null._h if self.not_null is not None else None, C.byref(cnt), C.byref(pb)))
        return cnt.value, pb.value

    def find_eq_in(self, values, bv_out: "bvector | None" = None) -> "bvector":
        """find_eq(sv, start, end, bv_out)  src/bmsparsevec_algo.h:1399: rows whose value is IN the list, OR-ed into
        bv_out (as the reference does); one pipeline run with an OR target (one AND-SUB group per distinct value)"""
        vals = sorted(set(int(v) for v in values))
        acc = bv_out
        if 0 in vals:
            z = self._compare(CMP_EQ, 0)
            acc = z if acc is None else bvector.bit_or(acc, z, opt_compress)
        pipe = pipeline(self.ctx, agg_opt_disable_bvects_and_counts)
        pipe.set_or_target(acc)
        for v in vals:
            if v == 0:
                continue
            g = self._groups(v)
            if g is None:
                continue
            ag = pipe.add()
            for x in g[0]: ag.add(x, 0)
            for x in g[1]: ag.add(x, 1)
        if pipe.size():
            pipe.complete()
            self.agg.combine_and_sub(pipe)
            return pipe.get_or_target()
        return acc if acc is not None else self._compare(CMP_GT, (1 << 64) - 1)       # nothing to look for: empty result

    def invert(self, bv: "bvector") -> "bvector":
        """scanner.invert(sv, bv)  src/bmsparsevec_algo.h:2321: the other rows of [0, size), NULL rows excluded"""
        zero_planes = slice_scanner(self.ctx, [], size=self.size(), not_null=self.not_null)
        rows = zero_planes._compare(CMP_ZERO)            # no planes: every (not NULL) row of [0, size)
        return bvector.bit_sub(rows, bv, opt_compress)

    def find_gt(self, value: int) -> bvector: return self._compare(CMP_GT, value)              # :2690
    def find_ge(self, value: int) -> bvector: return self._compare(CMP_GE, value)              # :2717
    def find_lt(self, value: int) -> bvector: return self._compare(CMP_LT, value)              # :2790
    def find_le(self, value: int) -> bvector: return self._compare(CMP_LE, value)              # :2824
    def find_range(self, lo: int, hi: int) -> bvector: return self._compare(CMP_RANGE, lo, hi) # :2862
    def find_zero(self) -> bvector: return self._compare(CMP_ZERO)                             # :2290 (null_correct = true)
    def find_nonzero(self) -> bvector: return self._compare(CMP_NONZERO)                       # :4464
    def count(self, pred: int, v0: int = 0, v1: int = 0) -> int:
        """popcount of a comparison search without materialising it"""
        return self._compare(pred, v0, v1, count_only=True)

    def find_eq(self, value: int):
        """-> (bv_out or None, found)"""
        if int(value) == 0:                                              # find_eq(sv, 0, ..) == find_zero (:4366)
            t = self._compare(CMP_EQ, 0)
            return t, t.any()
        g = self._groups(int(value))
        if g is None:
            return None, False
        return self.agg.combine_and_sub(g[0], g[1])

    def find_eq_indices(self, value: int) -> np.ndarray:
        """find_eq(sv, value, BII)  src/bmsparsevec_algo.h:1096: matching rows as sorted indices"""
        t, found = self.find_eq(value)
        return t.to_indices() if (found and t is not None) else np.zeros(0, np.uint64)

    def find_first_eq(self, value: int):
        g = self._groups(int(value))
        if g is None:
            return False, 0
        return self.agg.find_first_and_sub(g[0], g[1])

    def find_eq_counts(self, values, method: str = "auto") -> np.ndarray:
        """counts[q] = rows equal to values[q].  method "transpose" (default when it applies: <= 32 planes, one device):
        one pass over the planes whatever the number of queries (bmx_slice_eq_counts); "pipeline": one AND-SUB group per
        query, the reference's formulation (prepare_and_sub_aggregator + pipeline)."""
        out = np.zeros(len(values), np.uint64)
        if method == "auto":
            method = "transpose" if self._can_transpose() else "pipeline"
        if method == "transpose":
            vals = np.ascontiguousarray([int(v) for v in values], np.uint64)
            arr = (C.c_void_p * max(len(self.slices), 1))()
            for i, p in enumerate(self.slices):
                arr[i] = p._h if p is not None else None
            if vals.size:
                self._eq_counts_call(arr, vals, out)
            return out
        pipe = self._new_pipeline()
        slot = []
        for q, v in enumerate(values):
            if int(v) == 0:
                out[q] = self.count(CMP_EQ, 0); slot.append(-1); continue
            g = self._groups(int(v))
            if g is None:
                slot.append(-1); continue
            ag = pipe.add()
            for x in g[0]: ag.add(x, 0)
            for x in g[1]: ag.add(x, 1)
            slot.append(pipe.size() - 1)
        if pipe.size():
            pipe.complete()
            cnt = self.agg.combine_and_sub(pipe)
            for q, sl in enumerate(slot):
                if sl >= 0: out[q] = cnt[sl]
        return out


# ---------------------------------------------------------------------------------------------------
# multi-GPU: device groups (include/bmx.h "device groups")
# ---------------------------------------------------------------------------------------------------
GROUP_HOST_SUM, GROUP_RCCL = 0, 1
ROLE_AND, ROLE_OR, ROLE_SUB = 0, 1, 2


class group:
    """n devices driven by one host thread (bmx_group): every vector is sharded by block range over the
    members (column independence, src/bmaggregator.h:1184-1218), counts are summed on the host or by an
    in-library RCCL all-reduce (flags=GROUP_RCCL).  A device may be listed several times."""

    def __init__(self, devices: Sequence[int], flags: int = GROUP_HOST_SUM):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        check(lib().bmx_group_create(arr, len(devices), flags, C.byref(self._h)))
        self.devices = list(devices)

    def close(self):
        if self._h:
            lib().bmx_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        n = C.c_int()
        check(lib().bmx_group_size(self._h, C.byref(n)))
        return n.value

    def collection_prepare(self, vecs, role: int):
        """bmx_gcollection_prepare: every member transposes its block range of the (sharded) vectors"""
        check(lib().bmx_gcollection_prepare(self._h, _handles(vecs), len(vecs), role))

    def shard_range(self, nblocks: int, member: int) -> tuple[int, int]:
        lo, hi = C.c_uint32(), C.c_uint32()
        check(lib().bmx_group_shard_range(self._h, nblocks, member, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def set_tuning(self, key: str, value: int):
        for m in range(self.size()):
            c = C.c_void_p()
            check(lib().bmx_group_ctx(self._h, m, C.byref(c)))
            check(lib().bmx_ctx_set_tuning(c, key.encode(), int(value)))

    def member_pack_stats(self, member: int) -> dict:
        """bmx_ctx_pack_stats of one member's context: the collections it holds for its block range"""
        c = C.c_void_p()
        check(lib().bmx_group_ctx(self._h, member, C.byref(c)))
        n, b, ms = C.c_uint32(), C.c_uint64(), C.c_float()
        check(lib().bmx_ctx_pack_stats(c, C.byref(n), C.byref(b), C.byref(ms)))
        return {"collections": n.value, "bytes": b.value, "last_build_ms": ms.value}

    # -- byte-weighted shard borders (src/bmblocks.h:556-564: NULL ranges cost nothing) --
    def set_partition(self, nblocks: int, bounds) -> None:
        b = np.ascontiguousarray(bounds, np.uint32)
        if b.size != self.size() + 1:
            raise ValueError("need size() + 1 borders")
        check(lib().bmx_group_set_partition(self._h, nblocks, _ptr(b)))

    def partition_by_weight(self, weight) -> np.ndarray:
        """cut [0, len(weight)) where the running weight reaches m/n of the total; -> the n + 1 borders"""
        w = np.ascontiguousarray(weight, np.uint64)
        out = np.zeros(self.size() + 1, np.uint32)
        check(lib().bmx_group_partition_by_weight(self._h, w.size, _ptr(w) if w.size else None, _ptr(out)))
        return out

    def partition_for_tables(self, tables) -> np.ndarray:
        """borders from the algorithmic operand bytes of a collection of host block tables
        [(kinds, offs, bit_slab, gap_slab), ...] of equal length (bmx_block_table_weights per table)"""
        tables = list(tables)
        nblocks = len(tables[0][0])
        w = np.zeros(nblocks, np.uint64)
        for kinds, offs, _bits, gaps in tables:
            kinds = np.ascontiguousarray(kinds, np.uint8); offs = np.ascontiguousarray(offs, np.uint32)
            gaps = np.ascontiguousarray(gaps, np.uint16)
            if kinds.size != nblocks:
                raise ValueError("block tables of one partition must have the same length")
            check(lib().bmx_block_table_weights(nblocks, _ptr(kinds), _ptr(offs), _ptr(gaps) if gaps.size else None, gaps.size, _ptr(w)))
        return self.partition_by_weight(w)

    def rccl_ranks(self) -> int:
        n = C.c_int()
        check(lib().bmx_group_rccl_ranks(self._h, C.byref(n)))
        return n.value


class grs_index:
    """rank-select index of a sharded vector: one bm::rs_index twin per shard + the ones before each shard"""

    def __init__(self, bv: "gbvector", handle):
        self.bv, self.grp, self._h = bv, bv.grp, handle

    def __del__(self):
        try:
            if self._h and self.grp._h:
                lib().bmx_grs_free(self.grp._h, self._h)
            self._h = None
        except Exception:
            pass

    def count(self) -> int:
        c = C.c_uint64()
        check(lib().bmx_grs_count(self._h, C.byref(c)))
        return c.value


class gbvector:
    """bit-vector sharded by block range over the members of a group (bmx_gvec)"""

    def __init__(self, grp: group, handle):
        self.grp, self._h = grp, handle

    def __del__(self):
        try:
            if self._h and self.grp._h:
                lib().bmx_gvec_free(self.grp._h, self._h)
            self._h = None
        except Exception:
            pass

    @staticmethod
    def from_block_table(grp: group, nbits: int, kinds, offs, bit_slab, gap_slab) -> "gbvector":
        kinds = np.ascontiguousarray(kinds, np.uint8); offs = np.ascontiguousarray(offs, np.uint32)
        bit_slab = np.ascontiguousarray(bit_slab, np.uint32); gap_slab = np.ascontiguousarray(gap_slab, np.uint16)
        h = C.c_void_p()
        check(lib().bmx_gvec_upload(grp._h, nbits, kinds.size, _ptr(kinds), _ptr(offs),
                                    _ptr(bit_slab) if bit_slab.size else None, bit_slab.size // BLOCK_WORDS,
                                    _ptr(gap_slab) if gap_slab.size else None, gap_slab.size, C.byref(h)))
        return gbvector(grp, h)

    @staticmethod
    def generate(grp: group, seed: int, vec_id: int, density_q16: int, nbits: int,
                 with_common: bool = False, optimize: bool = True) -> "gbvector":
        h = C.c_void_p()
        check(lib().bmx_gvec_generate(grp._h, seed, vec_id, int(with_common), density_q16, nbits, int(optimize), C.byref(h)))
        return gbvector(grp, h)

    def info(self):
        nbits, nblocks, slab, gw = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        counts = (C.c_uint32 * 4)()
        check(lib().bmx_gvec_info(self._h, C.byref(nbits), C.byref(nblocks), counts, C.byref(slab), C.byref(gw)))
        return {"nbits": nbits.value, "nblocks": nblocks.value, "counts": list(counts),
                "bit_slab_blocks": slab.value, "gap_words": gw.value}

    def shard_info(self, member: int):
        """bmx_vec_info of member m's shard (what that member holds of this vector)"""
        sh = C.c_void_p()
        check(lib().bmx_gvec_shard(self._h, member, C.byref(sh)))
        nbits, nblocks, slab, gw = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        counts = (C.c_uint32 * 4)()
        check(lib().bmx_vec_info(sh, C.byref(nbits), C.byref(nblocks), counts, C.byref(slab), C.byref(gw)))
        return {"nbits": nbits.value, "nblocks": nblocks.value, "counts": list(counts), "bit_slab_blocks": slab.value, "gap_words": gw.value}

    def block_table(self):
        i = self.info()
        kinds = np.zeros(i["nblocks"], np.uint8); offs = np.zeros(i["nblocks"], np.uint32)
        bit_slab = np.zeros(i["bit_slab_blocks"] * BLOCK_WORDS, np.uint32); gap_slab = np.zeros(i["gap_words"], np.uint16)
        check(lib().bmx_gvec_download(self.grp._h, self._h, _ptr(kinds), _ptr(offs),
                                      _ptr(bit_slab) if bit_slab.size else None, _ptr(gap_slab) if gap_slab.size else None))
        return kinds, offs, bit_slab, gap_slab

    def count(self) -> int:
        c = C.c_uint64()
        check(lib().bmx_gvec_count(self.grp._h, self._h, C.byref(c)))
        return c.value

    # -- rank / select: per-shard index, queries routed to the owning member (bmx_grs_*) --
    def build_rs_index(self) -> "grs_index":
        h = C.c_void_p()
        check(lib().bmx_grs_build(self.grp._h, self._h, C.byref(h)))
        return grs_index(self, h)

    def count_to(self, n, rs: "grs_index"):
        """ones in [0..n] inclusive; scalar or array of positions (src/bm.h:3120)"""
        arr = np.ascontiguousarray(np.atleast_1d(n), np.uint64)
        out = np.zeros(arr.shape, np.uint64)
        if arr.size:
            check(lib().bmx_grank_batch(self.grp._h, self._h, rs._h, _ptr(arr), arr.size, _ptr(out)))
        return int(out[0]) if np.isscalar(n) else out

    rank = count_to

    def select(self, rank, rs: "grs_index"):
        """-> (found, pos); rank is 1-based (src/bm.h:5350)"""
        arr = np.ascontiguousarray(np.atleast_1d(rank), np.uint64)
        pos = np.zeros(arr.shape, np.uint64)
        found = np.zeros(arr.shape, np.uint8)
        if arr.size:
            check(lib().bmx_gselect_batch(self.grp._h, self._h, rs._h, _ptr(arr), arr.size, _ptr(pos), _ptr(found)))
        if np.isscalar(rank):
            return bool(found[0]), int(pos[0])
        return found.astype(bool), pos

    @staticmethod
    def _op2(op, a: "gbvector", b: "gbvector", opt_mode: int = opt_none) -> "gbvector":
        h = C.c_void_p()
        check(lib().bmx_gvec_op2(a.grp._h, op, a._h, b._h, int(opt_mode == opt_compress), C.byref(h)))
        return gbvector(a.grp, h)

    @staticmethod
    def bit_and(a, b, opt_mode=opt_none): return gbvector._op2(AND, a, b, opt_mode)
    @staticmethod
    def bit_or(a, b, opt_mode=opt_none): return gbvector._op2(OR, a, b, opt_mode)
    @staticmethod
    def bit_xor(a, b, opt_mode=opt_none): return gbvector._op2(XOR, a, b, opt_mode)
    @staticmethod
    def bit_sub(a, b, opt_mode=opt_none): return gbvector._op2(SUB, a, b, opt_mode)

    @staticmethod
    def count_op2(op, a: "gbvector", b: "gbvector") -> int:
        c = C.c_uint64()
        check(lib().bmx_gvec_count_op2(a.grp._h, op, a._h, b._h, C.byref(c)))
        return c.value


class gpipeline:
    """aggregator::pipeline<agg_opt_only_counts> over sharded vectors (bmx_gpipeline)"""

    def __init__(self, grp: group):
        self.grp = grp
        self.groups: list[arg_groups] = []
        self._h = None
        self._counts = None

    def add(self) -> arg_groups:
        if self._h:
            raise RuntimeError("pipeline already complete()")
        g = arg_groups()
        self.groups.append(g)
        return g

    def size(self) -> int:
        return len(self.groups)

    def complete(self):
        and_list = [v for g in self.groups for v in g.arg_bv0]
        sub_list = [v for g in self.groups for v in g.arg_bv1]
        and_n = (C.c_uint32 * max(len(self.groups), 1))(*[len(g.arg_bv0) for g in self.groups])
        sub_n = (C.c_uint32 * max(len(self.groups), 1))(*[len(g.arg_bv1) for g in self.groups])
        h = C.c_void_p()
        check(lib().bmx_gpipeline_create(self.grp._h, _handles(and_list), and_n, _handles(sub_list), sub_n,
                                         len(self.groups), C.byref(h)))
        self._h = h
        if getattr(self, "search_count_limit", None) is not None:
            check(lib().bmx_gpipeline_set_search_count_limit(self.grp._h, self._h, min(int(self.search_count_limit), ID_MAX64)))

    def set_search_count_limit(self, limit: int):
        """pipeline::set_search_count_limit (src/bmaggregator.h:255) over shards: every member searches under the same limit"""
        self.search_count_limit = limit
        if self._h:
            check(lib().bmx_gpipeline_set_search_count_limit(self.grp._h, self._h, min(int(limit), ID_MAX64)))

    def is_complete(self) -> bool:
        return self._h is not None

    def get_bv_count_vector(self):
        return self._counts

    def last_ms(self) -> list:
        ms = (C.c_float * self.grp.size())()
        check(lib().bmx_gpipeline_last_ms(self.grp._h, self._h, ms))
        return list(ms)

    def last_exchange_ms(self) -> list:
        ms = (C.c_float * self.grp.size())()
        check(lib().bmx_gpipeline_last_exchange_ms(self.grp._h, self._h, ms))
        return list(ms)

    def operand_bytes(self) -> list:
        b = (C.c_uint64 * self.grp.size())()
        check(lib().bmx_gpipeline_operand_bytes(self.grp._h, self._h, b))
        return list(b)

    def describe(self, member: int = 0):
        buf = C.create_string_buffer(512)
        nl = C.c_uint32()
        check(lib().bmx_gpipeline_describe(self.grp._h, self._h, member, buf, 512, C.byref(nl)))
        return buf.value.decode(), nl.value

    def __del__(self):
        try:
            if self._h and self.grp._h:
                lib().bmx_gpipeline_destroy(self.grp._h, self._h)
            self._h = None
        except Exception:
            pass


class gaggregator:
    """bm::aggregator over the sharded vectors of a group: same calls, n GPUs"""

    pipeline = gpipeline

    def __init__(self, grp: group):
        self.grp = grp
        self.ag = arg_groups()
        self.opt_mode = False

    def set_optimization(self, opt: bool = True):
        self.opt_mode = bool(opt)

    def add(self, bv: gbvector | None, agr_group: int = 0) -> int:
        return self.ag.add(bv, agr_group)

    def reset(self):
        self.ag.reset()

    def combine_or(self, bv_src=None) -> gbvector:
        src = list(bv_src) if bv_src is not None else list(self.ag.arg_bv0)
        self.ag.reset()
        h = C.c_void_p()
        check(lib().bmx_gagg_or(self.grp._h, _handles(src), len(src), int(self.opt_mode), C.byref(h)))
        return gbvector(self.grp, h)

    def combine_and(self, bv_src=None) -> gbvector:
        src = list(bv_src) if bv_src is not None else self.ag.arg_bv0
        return self.combine_and_sub(src, [])[0]

    def combine_and_sub(self, bv_src_and=None, bv_src_sub=None):
        if isinstance(bv_src_and, gpipeline):
            pipe = bv_src_and
            if not pipe.is_complete():
                raise RuntimeError("pipeline is not complete()")
            out = np.zeros(max(pipe.size(), 1), np.uint64)
            check(lib().bmx_gpipeline_run_counts(self.grp._h, pipe._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
            pipe._counts = out[:pipe.size()]
            return pipe._counts
        a = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        s = list(bv_src_sub) if bv_src_sub is not None else self.ag.arg_bv1
        h, any_ = C.c_void_p(), C.c_int()
        check(lib().bmx_gagg_and_sub(self.grp._h, _handles(a), len(a), _handles(s), len(s), C.byref(h), C.byref(any_)))
        return gbvector(self.grp, h), bool(any_.value)

    def find_first_and_sub(self, bv_src_and=None, bv_src_sub=None):                      # :1079 / :1458
        """-> (found, idx): first set bit of AND(group 0) AND NOT OR(group 1); every member searches its shard"""
        a = list(bv_src_and) if bv_src_and is not None else self.ag.arg_bv0
        s = list(bv_src_sub) if bv_src_sub is not None else self.ag.arg_bv1
        found, idx = C.c_int(), C.c_uint64()
        check(lib().bmx_gfind_first_and_sub(self.grp._h, _handles(a), len(a), _handles(s), len(s), C.byref(found), C.byref(idx)))
        return bool(found.value), int(idx.value)


class gslice_scanner(slice_scanner):
    """slice_scanner over SHARDED bit-planes (gbvector): every member searches its own rows; results stay sharded,
    counts are summed (bmx_gslice_compare, gaggregator, gpipeline).  Planes, not_null and size must span the same
    block range (upload the planes with the same number of blocks)."""

    def __init__(self, grp: group, slices, size: int | None = None, not_null: gbvector | None = None):
        self.grp, self.slices = grp, list(slices)
        self.agg = gaggregator(grp)
        self.not_null = not_null
        self._size = size

    def size(self) -> int:
        if self._size is None:
            self._size = max([p.info()["nbits"] for p in self.slices if p is not None], default=0)
        return self._size

    def _new_pipeline(self):
        return gpipeline(self.grp)

    def _eq_counts_call(self, arr, vals, out):
        check(lib().bmx_gslice_eq_counts(self.grp._h, arr, len(self.slices), _ptr(vals), vals.size, self.size(),
                                         self.not_null._h if self.not_null is not None else None, _ptr(out)))

    def _compare(self, pred: int, v0: int = 0, v1: int = 0, count_only: bool = False):
        if v0 < 0 or v1 < 0 or v0 >= 1 << 64 or v1 >= 1 << 64:
            raise BmxError(_ffi.ERR_RANGE, "Incorrect range or index", "unsigned 64-bit values only")
        arr = (C.c_void_p * max(len(self.slices), 1))()
        for i, p in enumerate(self.slices):
            arr[i] = p._h if p is not None else None
        h, cnt = C.c_void_p(), C.c_uint64()
        check(lib().bmx_gslice_compare(self.grp._h, arr, len(self.slices), pred, v0, v1, self.size(),
                                       self.not_null._h if self.not_null is not None else None,
                                       None if count_only else C.byref(h), C.byref(cnt)))
        return cnt.value if count_only else gbvector(self.grp, h)

    def find_eq(self, value: int):
        """-> (bv_out or None, found)"""
        if int(value) == 0:
            t = self._compare(CMP_EQ, 0)
            return t, t.count() != 0
        g = self._groups(int(value))
        if g is None:
            return None, False
        return self.agg.combine_and_sub(g[0], g[1])
