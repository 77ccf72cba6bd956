"""oracle -- CPU checkers for the bvector/aggregator hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product (bitmagic_amd/) never does.

Two interchangeable back-ends behind one Python surface:

* ``port()``            -- oracle/bmx_oracle.c, the plain-C restatement (kind "port").
* ``reference(flavour)`` -- oracle/_ref/libbmref_{scalar,avx2}.so, the UNMODIFIED
  reference compiled from /root/reference/src by oracle/Makefile (kind "reference").
  Only present where it was built (the build container; the .so travels to the GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NULL, FULL, BIT, GAP = 0, 1, 2, 3
AND, OR, XOR, SUB = 0, 1, 2, 3
BLOCK_WORDS = 2048
BLOCK_BITS = 65536
COMMON_ID = 0xFFFFFFFF


def build(with_ref: bool = True) -> None:
    """Compile the C restatement (and the reference shim when the sources exist)."""
    target = "all" if with_ref else "_build/libbmx_oracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u16p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Vec:
    """Handle to a vector living inside one of the oracle libraries."""

    def __init__(self, orc: "Oracle", handle, nbits: int):
        self.orc, self.h, self.nbits = orc, handle, int(nbits)

    @property
    def nblocks(self) -> int:
        return (self.nbits + BLOCK_BITS - 1) // BLOCK_BITS

    def __del__(self):
        try:
            if self.h:
                self.orc._f("vec_free")(self.h)
                self.h = None
        except Exception:
            pass

    # -- inspection -------------------------------------------------------
    def count(self) -> int:
        return int(self.orc._f("vec_count")(self.h))

    def stat(self):
        counts = (C.c_uint32 * 4)()
        gw = C.c_uint64()
        if self.orc.is_ref:
            self.orc._f("vec_stat")(self.h, C.c_uint32(self.nblocks), counts, C.byref(gw))
        else:
            self.orc._f("vec_stat")(self.h, counts, C.byref(gw))
        return list(counts), int(gw.value)

    def flatten(self):
        """-> (kinds u8[nb], offs u32[nb], bit_slab u32[nbit*2048], gap_slab u16[gw])"""
        counts, gw = self.stat()
        nb = self.nblocks
        kinds = np.zeros(nb, np.uint8)
        offs = np.zeros(nb, np.uint32)
        bit_slab = np.zeros(counts[BIT] * BLOCK_WORDS, np.uint32)
        gap_slab = np.zeros(max(gw, 1), np.uint16)
        if self.orc.is_ref:
            self.orc._f("vec_flatten")(self.h, C.c_uint32(nb), _u8p(kinds), _u32p(offs), _u32p(bit_slab), _u16p(gap_slab))
        else:
            self.orc._f("vec_flatten")(self.h, _u8p(kinds), _u32p(offs), _u32p(bit_slab), _u16p(gap_slab))
        return kinds, offs, bit_slab, gap_slab[:gw]

    def to_words(self, nwords: int | None = None) -> np.ndarray:
        if nwords is None:
            nwords = self.nblocks * BLOCK_WORDS
        out = np.zeros(nwords, np.uint32)
        self.orc._f("vec_to_words")(self.h, _u32p(out), C.c_uint64(nwords))
        return out

    def get_bit(self, n: int) -> int:
        return int(self.orc._f("vec_get_bit")(self.h, C.c_uint64(n)))

    # -- mutation (test helpers for the known-answer cases) -----------------
    def set_bit(self, n: int):
        self.orc._f("vec_set_bit")(self.h, C.c_uint64(n))

    def set_range(self, l: int, r: int):
        self.orc._f("vec_set_range")(self.h, C.c_uint64(l), C.c_uint64(r))

    def optimize(self):
        self.orc._f("vec_optimize")(self.h)


class RS:
    def __init__(self, orc, handle, vec):
        self.orc, self.h, self.vec = orc, handle, vec

    def __del__(self):
        try:
            if self.h:
                self.orc._f("rs_free")(self.h)
                self.h = None
        except Exception:
            pass

    def count(self) -> int:
        return int(self.orc._f("rs_count")(self.h))

    def total_blocks(self) -> int:
        return int(self.orc._f("rs_total_blocks")(self.h))

    def export(self, nblocks=None):
        n = self.vec.nblocks if nblocks is None else nblocks
        bc = np.zeros(n, np.uint32)
        sub = np.zeros(n, np.uint64)
        if self.orc.is_ref:
            self.orc._f("rs_export")(self.h, C.c_uint32(n), _u32p(bc), _u64p(sub))
        else:
            n = min(n, self.total_blocks())
            bc = bc[:n]; sub = sub[:n]
            self.orc._f("rs_export")(self.h, _u32p(bc), _u64p(sub))
        return bc, sub

    def count_range(self, l: int, r: int) -> int:
        return int(self.orc._f("count_range")(self.vec.h, self.h, C.c_uint64(l), C.c_uint64(r)))

    def rank_corrected(self, n: int) -> int:
        return int(self.orc._f("rank_corrected")(self.vec.h, self.h, C.c_uint64(n)))

    def count_to_test(self, n: int) -> int:
        return int(self.orc._f("count_to_test")(self.vec.h, self.h, C.c_uint64(n)))

    def find_rank(self, rank: int, frm: int):
        pos = C.c_uint64()
        f = self.orc._f("find_rank")(self.vec.h, self.h, C.c_uint64(rank), C.c_uint64(frm), C.byref(pos))
        return bool(f), int(pos.value)

    def rank(self, n) -> np.ndarray:
        n = np.ascontiguousarray(n, np.uint64)
        out = np.zeros(n.shape, np.uint64)
        self.orc._f("rank_batch")(self.vec.h, self.h, _u64p(n), C.c_size_t(n.size), _u64p(out))
        return out

    def select(self, r):
        r = np.ascontiguousarray(r, np.uint64)
        pos = np.zeros(r.shape, np.uint64)
        found = np.zeros(r.shape, np.uint8)
        self.orc._f("select_batch")(self.vec.h, self.h, _u64p(r), C.c_size_t(r.size), _u64p(pos), _u8p(found))
        return pos, found.astype(bool)


class SV:
    """bm::sparse_vector<unsigned, bvector<>> living inside the reference library + bm::sparse_vector_scanner<> calls"""

    def __init__(self, orc, handle, n):
        self.orc, self.h, self.n = orc, handle, int(n)

    def __del__(self):
        try:
            if self.h:
                self.orc.lib.ref_sv_free(self.h)
                self.h = None
        except Exception:
            pass

    def size(self) -> int:
        return int(self.orc.lib.ref_sv_size(self.h))

    def effective_slices(self) -> int:
        return int(self.orc.lib.ref_sv_effective_slices(self.h))

    def slice(self, i: int):
        h = self.orc.lib.ref_sv_slice(self.h, C.c_uint32(i))
        return Vec(self.orc, h, self.n) if h else None

    def not_null(self):
        h = self.orc.lib.ref_sv_not_null(self.h)
        return Vec(self.orc, h, self.n) if h else None

    def compare(self, pred: int, v0: int = 0, v1: int = 0) -> Vec:
        """pred = BMX_CMP_*: find_gt / ge / lt / le / range / eq / zero / nonzero"""
        return Vec(self.orc, self.orc.lib.ref_sv_compare(self.h, C.c_int(pred), C.c_uint32(v0), C.c_uint32(v1)), self.n)

    def find_first_eq(self, v: int):
        pos = C.c_uint64()
        f = self.orc.lib.ref_sv_find_first_eq(self.h, C.c_uint32(v), C.byref(pos))
        return bool(f), int(pos.value)

    def find_eq_in(self, values) -> Vec:
        """find_eq(sv, start, end, bv_out): rows whose value is IN the list (bmsparsevec_algo.h:1399)"""
        v = np.ascontiguousarray(values, np.uint32)
        return Vec(self.orc, self.orc.lib.ref_sv_find_eq_in(self.h, _u32p(v), C.c_size_t(v.size)), self.n)

    def invert(self, bv: Vec) -> Vec:
        """scanner.invert(sv, bv) on a copy of bv (bmsparsevec_algo.h:2321)"""
        return Vec(self.orc, self.orc.lib.ref_sv_invert(self.h, bv.h), self.n)


class SVS:
    """bm::sparse_vector<int, bvector<>> (signed) + scanner calls"""

    def __init__(self, orc, handle, n):
        self.orc, self.h, self.n = orc, handle, int(n)

    def __del__(self):
        try:
            if self.h:
                self.orc.lib.ref_svs_free(self.h)
                self.h = None
        except Exception:
            pass

    def size(self) -> int:
        return int(self.orc.lib.ref_svs_size(self.h))

    def effective_slices(self) -> int:
        return int(self.orc.lib.ref_svs_effective_slices(self.h))

    def slice(self, i: int):
        h = self.orc.lib.ref_svs_slice(self.h, C.c_uint32(i))
        return Vec(self.orc, h, self.n) if h else None

    def compare(self, pred: int, v0: int = 0, v1: int = 0) -> Vec:
        return Vec(self.orc, self.orc.lib.ref_svs_compare(self.h, C.c_int(pred), C.c_int32(v0), C.c_int32(v1)), self.n)


class Oracle:
    def __init__(self, path: str, prefix: str, kind: str, name: str):
        self.lib = C.CDLL(path)
        self.prefix, self.kind, self.name = prefix, kind, name
        self.is_ref = prefix == "ref_"
        L = self.lib
        vp = C.c_void_p
        for fn, res in [("vec_import", vp), ("vec_new", vp), ("op2", vp), ("agg_or", vp), ("agg_and_sub", vp),
                        ("rs_build", vp), ("vec_count", C.c_uint64), ("count_op2", C.c_uint64),
                        ("rs_count", C.c_uint64), ("rs_total_blocks", C.c_uint32), ("vec_get_bit", C.c_int)]:
            getattr(L, prefix + fn).restype = res
        for fn in ["vec_free", "vec_stat", "vec_flatten", "vec_to_words", "vec_set_bit", "vec_set_range",
                   "vec_optimize", "rs_free", "rs_export", "rank_batch", "select_batch", "agg_pipeline_counts"]:
            getattr(L, prefix + fn).restype = None
        for fn in ["vec_free", "vec_count", "vec_optimize", "rs_build", "rs_free", "rs_count", "rs_total_blocks"]:
            getattr(L, prefix + fn).argtypes = [vp]
        for fn in ["vec_set_bit", "vec_get_bit"]:
            getattr(L, prefix + fn).argtypes = [vp, C.c_uint64]
        getattr(L, prefix + "vec_set_range").argtypes = [vp, C.c_uint64, C.c_uint64]
        getattr(L, prefix + "vec_to_words").argtypes = [vp, C.POINTER(C.c_uint32), C.c_uint64]
        getattr(L, prefix + "op2").argtypes = [C.c_int, vp, vp, C.c_int]
        getattr(L, prefix + "count_op2").argtypes = [C.c_int, vp, vp]
        getattr(L, prefix + "agg_or").argtypes = [C.POINTER(vp), C.c_size_t]
        getattr(L, prefix + "agg_and_sub").argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(vp), C.c_size_t]
        getattr(L, prefix + "rank_batch").argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64)]
        getattr(L, prefix + "select_batch").argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_size_t,
                                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
        for fn in ["count_range", "rank_corrected", "count_to_test"]:
            getattr(L, prefix + fn).restype = C.c_uint64
        getattr(L, prefix + "count_range").argtypes = [vp, vp, C.c_uint64, C.c_uint64]
        getattr(L, prefix + "rank_corrected").argtypes = [vp, vp, C.c_uint64]
        getattr(L, prefix + "count_to_test").argtypes = [vp, vp, C.c_uint64]
        getattr(L, prefix + "find_rank").restype = C.c_int
        getattr(L, prefix + "find_rank").argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        getattr(L, prefix + "agg_pipeline_results").restype = None
        getattr(L, prefix + "agg_pipeline_results").argtypes = [
            C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32), C.c_size_t,
            C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp)]
        getattr(L, prefix + "agg_or_opt").restype = vp
        getattr(L, prefix + "agg_or_opt").argtypes = [C.POINTER(vp), C.c_size_t, C.c_int]
        getattr(L, prefix + "agg_shift_right_and").restype = vp
        getattr(L, prefix + "agg_shift_right_and").argtypes = [C.POINTER(vp), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
        getattr(L, prefix + "agg_shift_right_and_count").restype = C.c_uint64
        getattr(L, prefix + "agg_shift_right_and_count").argtypes = [C.POINTER(vp), C.c_size_t]
        getattr(L, prefix + "vec_find_first").restype = C.c_int
        getattr(L, prefix + "vec_find_first").argtypes = [vp, C.POINTER(C.c_uint64)]
        getattr(L, prefix + "find_first_and_sub").restype = C.c_int
        getattr(L, prefix + "find_first_and_sub").argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(vp), C.c_size_t, C.POINTER(C.c_uint64)]
        if self.is_ref:
            L.ref_vec_new.argtypes = []
            L.ref_vec_stat.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
            L.ref_vec_flatten.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
            L.ref_rs_export.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
            L.ref_agg_member.restype = vp
            L.ref_agg_member.argtypes = [C.c_int, C.POINTER(vp), C.c_size_t]
            if hasattr(L, "ref_sv_new"):       # round-3 additions of the shim: range hint, masks pipeline, sparse_vector scanner
                L.ref_find_first_and_sub_range.restype = C.c_int
                L.ref_find_first_and_sub_range.argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(vp), C.c_size_t, C.c_uint64, C.c_uint64,
                                                           C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
                L.ref_agg_pipeline_masks.restype = None
                L.ref_agg_pipeline_masks.argtypes = [C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32), C.c_size_t,
                                                     C.c_uint64, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_uint64)]
                L.ref_sv_new.restype = vp
                L.ref_sv_new.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_uint64]
                L.ref_sv_free.restype = None; L.ref_sv_free.argtypes = [vp]
                L.ref_sv_size.restype = C.c_uint64; L.ref_sv_size.argtypes = [vp]
                L.ref_sv_effective_slices.restype = C.c_uint32; L.ref_sv_effective_slices.argtypes = [vp]
                L.ref_sv_slice.restype = vp; L.ref_sv_slice.argtypes = [vp, C.c_uint32]
                L.ref_sv_not_null.restype = vp; L.ref_sv_not_null.argtypes = [vp]
                L.ref_sv_compare.restype = vp; L.ref_sv_compare.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32]
                L.ref_sv_find_first_eq.restype = C.c_int; L.ref_sv_find_first_eq.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
            if hasattr(L, "ref_svs_new"):
                L.ref_svs_new.restype = vp; L.ref_svs_new.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_uint64]
                L.ref_svs_free.restype = None; L.ref_svs_free.argtypes = [vp]
                L.ref_svs_size.restype = C.c_uint64; L.ref_svs_size.argtypes = [vp]
                L.ref_svs_effective_slices.restype = C.c_uint32; L.ref_svs_effective_slices.argtypes = [vp]
                L.ref_svs_slice.restype = vp; L.ref_svs_slice.argtypes = [vp, C.c_uint32]
                L.ref_svs_compare.restype = vp; L.ref_svs_compare.argtypes = [vp, C.c_int, C.c_int32, C.c_int32]
                L.ref_sv_find_eq_in.restype = vp; L.ref_sv_find_eq_in.argtypes = [vp, C.POINTER(C.c_uint32), C.c_size_t]
                L.ref_sv_invert.restype = vp; L.ref_sv_invert.argtypes = [vp, vp]
        else:
            L.bmo_vec_new.argtypes = [C.c_uint64]
            L.bmo_vec_stat.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
            L.bmo_vec_flatten.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
            L.bmo_rs_export.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
            L.bmo_vec_from_table.restype = vp
            L.bmo_vec_from_table.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
            L.bmo_gen_word64.restype = C.c_uint64
            L.bmo_gen_word64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
            L.bmo_gen_words.restype = None
            L.bmo_gen_words.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64,
                                        C.c_uint64, C.POINTER(C.c_uint32)]
        getattr(L, prefix + "vec_import").argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_int]
        getattr(L, prefix + "agg_pipeline_counts").argtypes = [
            C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32), C.c_size_t,
            C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # -- construction -----------------------------------------------------
    def new(self, nbits: int) -> Vec:
        h = self.lib.ref_vec_new() if self.is_ref else self.lib.bmo_vec_new(C.c_uint64(nbits))
        return Vec(self, h, nbits)

    def import_words(self, words: np.ndarray, optimize: bool = True, nbits: int | None = None) -> Vec:
        words = np.ascontiguousarray(words, np.uint32)
        h = self._f("vec_import")(_u32p(words), C.c_uint64(words.size), C.c_int(int(optimize)))
        return Vec(self, h, words.size * 32 if nbits is None else nbits)

    def from_table(self, nbits, kinds, offs, bit_slab, gap_slab) -> Vec:
        assert not self.is_ref
        kinds = np.ascontiguousarray(kinds, np.uint8); offs = np.ascontiguousarray(offs, np.uint32)
        bit_slab = np.ascontiguousarray(bit_slab, np.uint32); gap_slab = np.ascontiguousarray(gap_slab, np.uint16)
        if gap_slab.size == 0:
            gap_slab = np.zeros(1, np.uint16)
        if bit_slab.size == 0:
            bit_slab = np.zeros(1, np.uint32)
        h = self.lib.bmo_vec_from_table(C.c_uint64(nbits), C.c_uint32(kinds.size), _u8p(kinds), _u32p(offs),
                                        _u32p(bit_slab), _u16p(gap_slab))
        return Vec(self, h, nbits)

    # -- operations -------------------------------------------------------
    def op2(self, op: int, a: Vec, b: Vec, opt_compress: bool = False) -> Vec:
        h = self._f("op2")(op, a.h, b.h, int(opt_compress))
        return Vec(self, h, max(a.nbits, b.nbits))

    def count_op2(self, op: int, a: Vec, b: Vec) -> int:
        return int(self._f("count_op2")(op, a.h, b.h))

    @staticmethod
    def _ptrs(vecs):
        arr = (C.c_void_p * max(len(vecs), 1))()
        for i, v in enumerate(vecs):
            arr[i] = v.h
        return arr

    def agg_or(self, vecs, opt_compress: bool = False) -> Vec:
        h = self._f("agg_or_opt")(self._ptrs(vecs), len(vecs), int(opt_compress))
        return Vec(self, h, max([v.nbits for v in vecs], default=0))

    def agg_and_sub(self, and_vecs, sub_vecs=()) -> Vec:
        h = self._f("agg_and_sub")(self._ptrs(and_vecs), len(and_vecs), self._ptrs(sub_vecs), len(sub_vecs))
        return Vec(self, h, max([v.nbits for v in list(and_vecs) + list(sub_vecs)], default=0))

    def agg_shift_right_and(self, vecs, opt_compress: bool = False, any: bool = False):
        """-> (target Vec, found)  aggregator::combine_shift_right_and  src/bmaggregator.h:2494"""
        f = C.c_int()
        h = self._f("agg_shift_right_and")(self._ptrs(vecs), len(vecs), int(opt_compress), int(any), C.byref(f))
        return Vec(self, h, max([v.nbits for v in vecs], default=0)), bool(f.value)

    def agg_shift_right_and_count(self, vecs) -> int:
        return int(self._f("agg_shift_right_and_count")(self._ptrs(vecs), len(vecs)))

    def agg_member(self, kind: str, vecs) -> Vec:
        """reference only: add()+combine_and()/combine_or() member API"""
        assert self.is_ref
        h = self.lib.ref_agg_member(0 if kind == "and" else 1, self._ptrs(vecs), len(vecs))
        return Vec(self, h, max([v.nbits for v in vecs], default=0))

    def pipeline_counts(self, groups, nb_from: int = 0, nb_to: int | None = None) -> np.ndarray:
        """groups: list of (and_vecs, sub_vecs).  Block range only honoured by the port."""
        and_list = [v for g in groups for v in g[0]]
        sub_list = [v for g in groups for v in g[1]]
        and_n = np.array([len(g[0]) for g in groups], np.uint32)
        sub_n = np.array([len(g[1]) for g in groups], np.uint32)
        if nb_to is None:
            nb_to = max([v.nblocks for v in and_list + sub_list], default=0)
        out = np.zeros(len(groups), np.uint64)
        self._f("agg_pipeline_counts")(self._ptrs(and_list), _u32p(and_n), self._ptrs(sub_list), _u32p(sub_n),
                                       len(groups), C.c_uint32(nb_from), C.c_uint32(nb_to), _u64p(out))
        return out

    def pipeline_counts_limit(self, groups, limit: int) -> np.ndarray:
        """pipeline::set_search_count_limit(limit) + counts-only run.  The reference (src/bmaggregator.h:255,1361-1367) stops
        evaluating a group on the blocks that follow the one where its count reached the limit; the port restates exactly that
        rule from per-block counts (the block range form of its pipeline)."""
        if self.is_ref:
            and_list = [v for g in groups for v in g[0]]
            sub_list = [v for g in groups for v in g[1]]
            and_n = np.array([len(g[0]) for g in groups], np.uint32)
            sub_n = np.array([len(g[1]) for g in groups], np.uint32)
            out = np.zeros(len(groups), np.uint64)
            fn = self.lib.ref_agg_pipeline_counts_limit
            fn.restype = None
            fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_size_t,
                           C.c_uint64, C.POINTER(C.c_uint64)]
            fn(self._ptrs(and_list), _u32p(and_n), self._ptrs(sub_list), _u32p(sub_n), len(groups), C.c_uint64(limit), _u64p(out))
            return out
        nb = max([v.nblocks for g in groups for v in g[0] + g[1]], default=0)
        out = np.zeros(len(groups), np.uint64)
        for b in range(nb):
            live = out < np.uint64(limit)
            if not live.any(): break
            per = self.pipeline_counts(groups, b, b + 1)
            out[live] += per[live]
        return out

    def pipeline_results(self, groups):
        """-> (results: list[Vec|None], counts: np.ndarray, or_target: Vec)"""
        and_list = [v for g in groups for v in g[0]]
        sub_list = [v for g in groups for v in g[1]]
        and_n = np.array([len(g[0]) for g in groups], np.uint32)
        sub_n = np.array([len(g[1]) for g in groups], np.uint32)
        nbits = max([v.nbits for v in and_list + sub_list], default=0)
        res = (C.c_void_p * max(len(groups), 1))()
        cnt = np.zeros(len(groups), np.uint64)
        ort = C.c_void_p()
        self._f("agg_pipeline_results")(self._ptrs(and_list), _u32p(and_n), self._ptrs(sub_list), _u32p(sub_n),
                                        len(groups), res, _u64p(cnt), C.byref(ort))
        return [Vec(self, res[g], nbits) if res[g] else None for g in range(len(groups))], cnt, Vec(self, ort, nbits)

    def find_first(self, v: Vec):
        pos = C.c_uint64()
        f = self._f("vec_find_first")(v.h, C.byref(pos))
        return bool(f), int(pos.value)

    def find_first_and_sub(self, and_vecs, sub_vecs=()):
        idx = C.c_uint64()
        f = self._f("find_first_and_sub")(self._ptrs(and_vecs), len(and_vecs), self._ptrs(sub_vecs), len(sub_vecs), C.byref(idx))
        return bool(f), int(idx.value)

    # -- reference only: range hint, masks pipeline, bm::sparse_vector_scanner (the generators of the round-3 fixtures) --
    def find_first_and_sub_range(self, and_vecs, sub_vecs, frm: int, to: int):
        """-> (hint_accepted, found, idx): aggregator::set_range_hint(frm, to) + find_first_and_sub (bmaggregator.h:481,974,1458)"""
        assert self.is_ref
        idx, ok = C.c_uint64(), C.c_int()
        f = self.lib.ref_find_first_and_sub_range(self._ptrs(and_vecs), len(and_vecs), self._ptrs(sub_vecs), len(sub_vecs),
                                                  C.c_uint64(frm), C.c_uint64(to), C.byref(idx), C.byref(ok))
        return bool(ok.value), bool(f), int(idx.value)

    def pipeline_masks(self, groups, frm: int, to: int):
        """pipeline<agg_run_options<true, true, true>> under set_range_hint(frm, to) -> (results: list[Vec|None], counts)"""
        assert self.is_ref
        and_list = [v for g in groups for v in g[0]]
        sub_list = [v for g in groups for v in g[1]]
        and_n = np.array([len(g[0]) for g in groups], np.uint32)
        sub_n = np.array([len(g[1]) for g in groups], np.uint32)
        nbits = max([v.nbits for v in and_list + sub_list], default=0)
        res = (C.c_void_p * max(len(groups), 1))()
        cnt = np.zeros(len(groups), np.uint64)
        self.lib.ref_agg_pipeline_masks(self._ptrs(and_list), _u32p(and_n), self._ptrs(sub_list), _u32p(sub_n), len(groups),
                                        C.c_uint64(frm), C.c_uint64(to), res, _u64p(cnt))
        return [Vec(self, res[g], nbits) if res[g] else None for g in range(len(groups))], cnt

    def sparse_vector(self, values, is_null=None) -> "SV":
        assert self.is_ref
        values = np.ascontiguousarray(values, np.uint32)
        nn = None if is_null is None else np.ascontiguousarray(is_null, np.uint8)
        h = self.lib.ref_sv_new(_u32p(values), _u8p(nn) if nn is not None else None, C.c_uint64(values.size))
        return SV(self, h, values.size)

    def sparse_vector_signed(self, values, is_null=None) -> "SVS":
        assert self.is_ref
        values = np.ascontiguousarray(values, np.int32)
        nn = None if is_null is None else np.ascontiguousarray(is_null, np.uint8)
        h = self.lib.ref_svs_new(values.ctypes.data_as(C.POINTER(C.c_int32)), _u8p(nn) if nn is not None else None, C.c_uint64(values.size))
        return SVS(self, h, values.size)

    def rs_build(self, v: Vec) -> RS:
        return RS(self, self._f("rs_build")(v.h), v)

    # -- synthetic generator (port only; normative spec shared with the HIP kernel) --
    def gen_words(self, seed: int, vec_id: int, density_q16: int, nbits: int, with_common: bool = False,
                  word_off: int = 0, nwords: int | None = None) -> np.ndarray:
        assert not self.is_ref
        if nwords is None:
            nwords = ((nbits + 63) // 64) * 2
        out = np.zeros(nwords, np.uint32)
        self.lib.bmo_gen_words(C.c_uint64(seed), C.c_uint32(vec_id), C.c_int(int(with_common)),
                               C.c_uint32(density_q16), C.c_uint64(nbits), C.c_uint64(word_off),
                               C.c_uint64(nwords), _u32p(out))
        return out


_cache: dict = {}


def port() -> Oracle:
    if "port" not in _cache:
        path = os.path.join(_HERE, "_build", "libbmx_oracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        _cache["port"] = Oracle(path, "bmo_", "port", "bmx_oracle.c")
    return _cache["port"]


def reference_path(flavour: str = "avx2") -> str:
    return os.path.join(_HERE, "_ref", f"libbmref_{flavour}.so")


def have_reference(flavour: str = "avx2") -> bool:
    return os.path.exists(reference_path(flavour))


def reference(flavour: str = "avx2") -> Oracle:
    key = "ref_" + flavour
    if key not in _cache:
        _cache[key] = Oracle(reference_path(flavour), "ref_", "reference", f"BitMagic 9.2.1 ({flavour})")
    return _cache[key]
