"""CPU: the C-ABI library loads and exports every symbol include/bmx.h declares
(no compute calls without a GPU), and the host-side mirror keeps the reference's
argument / error behaviour."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bitmagic_amd import _ffi
    names = _ffi.exported_symbols()
    assert len(names) >= 30
    L = C.CDLL(_ffi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    L2 = _ffi.lib()                       # typed loader must also accept the library
    assert L2.bmx_simd_version() == 950
    assert b"BMX-00" in L2.bmx_error_msg(0) and b"BMX-03" in L2.bmx_error_msg(3)


def test_header_cites_reference_for_each_entry_point():
    txt = open(os.path.join(ROOT, "include", "bmx.h")).read()
    assert 'extern "C"' in txt
    assert "torch" not in txt.lower()
    for ref in ("src/bm.h:6185", "src/bmaggregator.h:1162", "src/bmaggregator.h:1292", "src/bm.h:2531",
                "src/bm.h:5350", "src/bmalgo.h:49", "src/bmbvimport.h:46", "libbm.h"):
        assert ref in txt, ref


def test_product_never_touches_oracle():
    """the shipped package must not import / load anything under oracle/"""
    pkg = os.path.join(ROOT, "bitmagic_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, re.M), f
                assert "libbmx_oracle" not in src and "libbmref" not in src, f


def test_arg_groups_semantics():
    """arg_groups::add: group index > 1 is BM_ERR_RANGE, nullptr is ignored
    (src/bmaggregator.h:2931-2947)"""
    import bitmagic_amd as bm
    ag = bm.arg_groups()
    assert ag.add(None, 0) == 0 and ag.arg_bv0 == []
    with pytest.raises(bm.BmxError) as e:
        ag.add(None, 2)
    assert e.value.status == 3


def test_no_device_fails_loudly():
    """on a machine without a GPU the context must raise, never fall back"""
    import bitmagic_amd as bm
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(bm.BmxError):
        bm.context(0)
