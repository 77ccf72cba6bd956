"""ctypes binding of the C-ABI declared in include/bmx.h (bitmagic_amd/lib/libbmx.so).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is
usable, every entry point raises.  (The CPU oracle lives under oracle/ and is
test infrastructure only -- this package never imports it.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BMX_LIB selects another build of the same ABI (the tuning build lib/libbmx_tune.so: extra launch shapes + diagnostics)
LIB_PATH = os.environ.get("BMX_LIB") or os.path.join(_HERE, "lib", "libbmx.so")

OK, ERR_BADALLOC, ERR_BADARG, ERR_RANGE, ERR_DEVICE = 0, 1, 2, 3, 4


class BmxError(RuntimeError):
    def __init__(self, status: int, msg: str, detail: str):
        super().__init__(f"{msg}" + (f" [{detail}]" if detail else ""))
        self.status = status


_lib = None


def lib() -> C.CDLL:
    """Load libbmx.so (once).  torch is imported first when available so that the
    process holds a single HIP runtime (torch bundles its own libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C bitmagic_amd/csrc)")
    try:  # plumbing only: share torch's HIP runtime / streams when torch is around
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    P = C.POINTER
    sig = {
        "bmx_error_msg": (C.c_char_p, [i32]),
        "bmx_last_error": (C.c_char_p, []),
        "bmx_simd_version": (i32, []),
        "bmx_device_count": (i32, [P(i32)]),
        "bmx_ctx_create": (i32, [i32, vp, P(vp)]),
        "bmx_ctx_destroy": (i32, [vp]),
        "bmx_ctx_synchronize": (i32, [vp]),
        "bmx_ctx_set_tuning": (i32, [vp, C.c_char_p, i32]),
        "bmx_ctx_mem_used": (i32, [vp, P(u64)]),
        "bmx_ctx_trim": (i32, [vp]),
        "bmx_vec_upload": (i32, [vp, u64, u32, vp, vp, vp, u32, vp, u64, P(vp)]),
        "bmx_vec_import_bits": (i32, [vp, vp, u64, i32, P(vp)]),
        "bmx_vec_generate": (i32, [vp, u64, u32, i32, u32, u64, i32, P(vp)]),
        "bmx_vec_generate_shard": (i32, [vp, u64, u32, i32, u32, u64, u32, u32, i32, P(vp)]),
        "bmx_vec_free": (i32, [vp, vp]),
        "bmx_vec_info": (i32, [vp, P(u64), P(u32), P(u32), P(u32), P(u64)]),
        "bmx_vec_operand_bytes": (i32, [vp, vp, P(u64)]),
        "bmx_vec_download": (i32, [vp, vp, vp, vp, vp, vp]),
        "bmx_vec_to_indices": (i32, [vp, vp, i32, vp, u64, P(u64)]),
        "bmx_vec_to_indices_dev": (i32, [vp, vp, i32, vp, u64, P(u64)]),
        "bmx_agg_and_sub_indices": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, i32, vp, u64, P(u64)]),
        "bmx_vec_to_words": (i32, [vp, vp, vp, u64]),
        "bmx_count": (i32, [vp, vp, P(u64)]),
        "bmx_op2": (i32, [vp, i32, vp, vp, i32, P(vp)]),
        "bmx_op2_count": (i32, [vp, i32, vp, vp, i32, P(vp), P(u64)]),
        "bmx_op2_dev": (i32, [vp, i32, vp, vp, vp, vp, P(vp)]),
        "bmx_pending_wait": (i32, [vp, vp, P(vp)]),
        "bmx_pending_free": (i32, [vp, vp]),
        "bmx_count_op2": (i32, [vp, i32, vp, vp, P(u64)]),
        "bmx_count_op2_dev": (i32, [vp, i32, vp, vp, vp]),
        "bmx_agg_or": (i32, [vp, P(vp), C.c_size_t, P(vp)]),
        "bmx_agg_or_opt": (i32, [vp, P(vp), C.c_size_t, i32, P(vp)]),
        "bmx_agg_and_sub": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, P(vp), P(i32)]),
        "bmx_find_first_and_sub": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, P(i32), P(u64)]),
        "bmx_find_first_and_sub_range": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, u64, u64, P(i32), P(u64)]),
        "bmx_agg_shift_right_and": (i32, [vp, P(vp), C.c_size_t, i32, i32, P(vp), P(i32)]),
        "bmx_agg_shift_right_and_count": (i32, [vp, P(vp), C.c_size_t, P(u64)]),
        "bmx_slice_compare": (i32, [vp, P(vp), C.c_size_t, i32, u64, u64, u64, vp, P(vp), P(u64)]),
        "bmx_slice_compare_signed": (i32, [vp, P(vp), C.c_size_t, i32, C.c_int64, C.c_int64, u64, vp, P(vp), P(u64)]),
        "bmx_slice_compare_stat": (i32, [vp, P(vp), C.c_size_t, i32, u64, u64, u64, vp, P(u64), P(u64)]),
        "bmx_slice_eq_counts": (i32, [vp, P(vp), C.c_size_t, vp, C.c_size_t, u64, vp, vp]),
        "bmx_collection_prepare": (i32, [vp, P(vp), C.c_size_t, i32]),
        "bmx_ctx_pack_stats": (i32, [vp, P(u32), P(u64), P(C.c_float)]),
        "bmx_ctx_pack_run_bytes": (i32, [vp, P(u64)]),
        "bmx_pipeline_create": (i32, [vp, P(vp), P(u32), P(vp), P(u32), C.c_size_t, P(vp)]),
        "bmx_pipeline_destroy": (i32, [vp, vp]),
        "bmx_pipeline_set_search_count_limit": (i32, [vp, vp, u64]),
        "bmx_pipeline_last_windows": (i32, [vp, P(u32), P(u32)]),
        "bmx_pipeline_last_window_groups": (i32, [vp, P(u32), u32, P(u32)]),
        "bmx_pipeline_run_counts": (i32, [vp, vp, u32, u32, P(u64)]),
        "bmx_pipeline_run_counts_dev": (i32, [vp, vp, u32, u32, vp]),
        "bmx_pipeline_run_results": (i32, [vp, vp, P(vp), P(u64), vp, P(vp)]),
        "bmx_pipeline_run_results_range": (i32, [vp, vp, u32, u32, P(vp), P(u64), vp, P(vp)]),
        "bmx_pipeline_run_results_hint": (i32, [vp, vp, u64, u64, P(vp), P(u64), vp, P(vp)]),
        "bmx_pipeline_operand_bytes": (i32, [vp, vp, u32, u32, P(u64)]),
        "bmx_pipeline_describe": (i32, [vp, vp, u32, u32, C.c_char_p, C.c_size_t, P(u32)]),
        "bmx_rs_build": (i32, [vp, vp, P(vp)]),
        "bmx_rs_free": (i32, [vp, vp]),
        "bmx_rs_info": (i32, [vp, P(u64), P(i32)]),
        "bmx_rs_select_format": (i32, [vp, P(i32), P(u64)]),
        "bmx_rs_count": (i32, [vp, P(u64)]),
        "bmx_rs_export": (i32, [vp, vp, vp, vp]),
        "bmx_rank_batch": (i32, [vp, vp, vp, vp, C.c_size_t, vp]),
        "bmx_select_batch": (i32, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
        "bmx_rank_batch_dev": (i32, [vp, vp, vp, vp, C.c_size_t, vp]),
        "bmx_select_batch_dev": (i32, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
        "bmx_group_create": (i32, [P(i32), i32, i32, P(vp)]),
        "bmx_group_destroy": (i32, [vp]),
        "bmx_group_size": (i32, [vp, P(i32)]),
        "bmx_group_ctx": (i32, [vp, i32, P(vp)]),
        "bmx_group_shard_range": (i32, [vp, u32, i32, P(u32), P(u32)]),
        "bmx_block_table_weights": (i32, [u32, vp, vp, vp, u64, vp]),
        "bmx_group_partition_by_weight": (i32, [vp, u32, vp, vp]),
        "bmx_group_set_partition": (i32, [vp, u32, vp]),
        "bmx_group_rccl_ranks": (i32, [vp, P(i32)]),
        "bmx_gvec_upload": (i32, [vp, u64, u32, vp, vp, vp, u32, vp, u64, P(vp)]),
        "bmx_gvec_generate": (i32, [vp, u64, u32, i32, u32, u64, i32, P(vp)]),
        "bmx_gvec_free": (i32, [vp, vp]),
        "bmx_gvec_info": (i32, [vp, P(u64), P(u32), P(u32), P(u32), P(u64)]),
        "bmx_gvec_shard": (i32, [vp, i32, P(vp)]),
        "bmx_gvec_download": (i32, [vp, vp, vp, vp, vp, vp]),
        "bmx_gvec_count": (i32, [vp, vp, P(u64)]),
        "bmx_gvec_count_op2": (i32, [vp, i32, vp, vp, P(u64)]),
        "bmx_gvec_op2": (i32, [vp, i32, vp, vp, i32, P(vp)]),
        "bmx_grs_build": (i32, [vp, vp, P(vp)]),
        "bmx_grs_free": (i32, [vp, vp]),
        "bmx_grs_count": (i32, [vp, P(u64)]),
        "bmx_grank_batch": (i32, [vp, vp, vp, vp, C.c_size_t, vp]),
        "bmx_gselect_batch": (i32, [vp, vp, vp, vp, C.c_size_t, vp, vp]),
        "bmx_gagg_or": (i32, [vp, P(vp), C.c_size_t, i32, P(vp)]),
        "bmx_gagg_and_sub": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, P(vp), P(i32)]),
        "bmx_gfind_first_and_sub": (i32, [vp, P(vp), C.c_size_t, P(vp), C.c_size_t, P(i32), P(u64)]),
        "bmx_gslice_compare": (i32, [vp, P(vp), C.c_size_t, i32, u64, u64, u64, vp, P(vp), P(u64)]),
        "bmx_gslice_eq_counts": (i32, [vp, P(vp), C.c_size_t, vp, C.c_size_t, u64, vp, vp]),
        "bmx_gpipeline_create": (i32, [vp, P(vp), P(u32), P(vp), P(u32), C.c_size_t, P(vp)]),
        "bmx_gpipeline_destroy": (i32, [vp, vp]),
        "bmx_gpipeline_run_counts": (i32, [vp, vp, P(u64)]),
        "bmx_gpipeline_set_search_count_limit": (i32, [vp, vp, u64]),
        "bmx_gcollection_prepare": (i32, [vp, P(vp), C.c_size_t, i32]),
        "bmx_gpipeline_last_ms": (i32, [vp, vp, P(C.c_float)]),
        "bmx_gpipeline_last_exchange_ms": (i32, [vp, vp, P(C.c_float)]),
        "bmx_gpipeline_operand_bytes": (i32, [vp, vp, P(u64)]),
        "bmx_gpipeline_describe": (i32, [vp, vp, i32, C.c_char_p, C.c_size_t, P(u32)]),
        "bmx_probe_random_lines": (i32, [vp, u64, u64, i32, P(C.c_float)]),
        "bmx_probe_stream_rw": (i32, [vp, u64, i32, i32, i32, P(C.c_float)]),
        "bmx_debug_redzone_check": (i32, [vp, P(i32), P(u64), C.c_char_p, C.c_size_t]),
        "bmx_debug_inject_failure": (i32, [vp, i32, C.c_longlong]),
        "bmx_timer_start": (i32, [vp]),
        "bmx_timer_stop_ms": (i32, [vp, P(C.c_float)]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    if missing:
        raise ImportError(f"libbmx.so does not export: {', '.join(missing)}")
    if hasattr(L, "bmx_diag_stream_read"):        # tuning build only (bitmagic_amd/csrc/bmx_diag.h)
        L.bmx_diag_stream_read.restype = i32
        L.bmx_diag_stream_read.argtypes = [vp, u64, i32, u32, i32, i32, P(C.c_float)]
    _lib = L
    return L


def exported_symbols():
    """Names include/bmx.h declares (used by the CPU-side ABI test)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "bmx.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(bmx_[a-z0-9_]+)\s*\(", txt)))


def check(status: int) -> None:
    if status != OK:
        L = lib()
        raise BmxError(status, L.bmx_error_msg(status).decode(), L.bmx_last_error().decode())
