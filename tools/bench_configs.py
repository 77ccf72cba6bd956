#!/usr/bin/env python3
"""Secondary BASELINE configs on one MI355X (configs[1], [3]; [4] as a 1-GPU slice).
Prints one JSON object per measurement.  Times are HIP-event medians on the launch stream."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bitmagic_amd as bm
from bitmagic_amd import _ffi
import ctypes as C

SEED = 0xB17A61C
ap = argparse.ArgumentParser()
ap.add_argument("--which", default="2,4")
ap.add_argument("--or-vecs", type=int, default=4096)
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
L = _ffi.lib()

def timed(fn, reps=15, warm=3):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        ctx.timer_start(); fn(); ts.append(ctx.timer_stop_ms())
    return float(np.median(ts)), float(np.min(ts))

if "2" in a.which.split(","):
    nbits = 1_000_000_000
    for dq, name in [(655, "1%"), (6554, "10%"), (32768, "50%")]:
        va = bm.bvector.generate(ctx, SEED, 1, dq, nbits); vb = bm.bvector.generate(ctx, SEED, 2, dq, nbits)
        st = va.calc_stat()
        ia, ib = va.info(), vb.info()
        in_bytes = (ia["counts"][2] + ib["counts"][2]) * 8192 + (ia["gap_words"] + ib["gap_words"]) * 2
        for op, opname in enumerate(["AND", "OR", "XOR", "SUB"]):
            c = C.c_uint64()
            med, mn = timed(lambda: L.bmx_count_op2(ctx._h, op, va._h, vb._h, C.byref(c)))
            dcnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            def burst():
                for _ in range(20): L.bmx_count_op2_dev(ctx._h, op, va._h, vb._h, C.c_void_p(dcnt.data_ptr()))
            kmed, kmn = timed(burst, reps=7, warm=2)
            kmn /= 20
            print(json.dumps({"config": 2, "op": "count_" + opname.lower(), "density": name, "block_types": st,
                              "sync_call_ms": round(mn, 4), "kernel_ms": round(kmn, 4), "operand_bytes": in_bytes,
                              "kernel_GBps": round(in_bytes / kmn / 1e6, 1), "kernel_Gbit_per_s_per_operand": round(nbits / kmn / 1e6, 1),
                              "count": c.value, "note": "kernel_ms = 20 back-to-back async launches / 20 (HIP events); operands (250 MB) fit the 256 MB Infinity Cache"}))
            res = []
            def run():
                t = bm.bvector._op2(op, va, vb, bm.opt_none); res.append(t)
                if len(res) > 2: res.pop(0)
            med, mn = timed(run, reps=8, warm=2)
            out_blocks = res[-1].info()["counts"][2]
            tot = in_bytes + out_blocks * 8192
            print(json.dumps({"config": 2, "op": "bit_" + opname.lower(), "density": name, "ms_median": round(med, 4),
                              "ms_min": round(mn, 4), "bytes_in_out": tot, "GBps": round(tot / mn / 1e6, 1),
                              "note": "incl. result allocation (hipMalloc of desc+slab) and layout scan"}))
            del res
        del va, vb

if "4" in a.which.split(","):
    nbits = 4_000_000_000
    for dq, name in [(6554, "10%"), (655, "1%")]:
        v = bm.bvector.generate(ctx, SEED, 7, dq, nbits)
        t0 = time.perf_counter(); rs = v.build_rs_index(); ctx.synchronize(); t_build = time.perf_counter() - t0
        med_b, mn_b = timed(lambda: v.build_rs_index(), reps=5, warm=1)
        nq = 10_000_000
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        qn = torch.randint(0, nbits, (nq,), device="cuda", dtype=torch.int64, generator=g)
        cnt = rs.count()
        qr = torch.randint(1, cnt + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        out = torch.zeros(nq, dtype=torch.int64, device="cuda"); pos = torch.zeros(nq, dtype=torch.int64, device="cuda")
        found = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        mr, mnr = timed(lambda: _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, qn.data_ptr(), nq, out.data_ptr())))
        ms, mns = timed(lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qr.data_ptr(), nq, pos.data_ptr(), found.data_ptr())))
        torch.cuda.synchronize()
        # round trip property: rank(select(r)) == r ; select positions are set bits
        chk = torch.zeros(nq, dtype=torch.int64, device="cuda")
        _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, pos.data_ptr(), nq, chk.data_ptr())); torch.cuda.synchronize()
        ok = bool((chk == qr).all().item()) and bool(found.all().item())
        print(json.dumps({"config": 4, "density": name, "block_types": v.calc_stat(), "count": cnt,
                          "rs_build_ms": round(mn_b, 3), "rank_ms_10M": round(mnr, 4), "rank_Mq_per_s": round(nq / mnr / 1e3, 1),
                          "select_ms_10M": round(mns, 4), "select_Mq_per_s": round(nq / mns / 1e3, 1),
                          "rank_select_roundtrip_ok": ok}))
        del v, rs

if "5" in a.which.split(","):
    nbits = 4_000_000_000
    nv = a.or_vecs
    t0 = time.perf_counter()
    vecs = [bm.bvector.generate(ctx, SEED, 10000 + i, 13, nbits) for i in range(nv)]      # 13/65536 = 0.02 %
    ctx.synchronize(); t_build = time.perf_counter() - t0
    gap_bytes = sum(v.info()["gap_words"] for v in vecs) * 2
    agg = bm.aggregator(ctx)
    res = []
    def run():
        res.append(agg.combine_or(vecs));
        if len(res) > 1: res.pop(0)
    med, mn = timed(run, reps=3, warm=1)
    r = res[-1]
    print(json.dumps({"config": 5, "vectors": nv, "bits": nbits, "block_types_vec0": vecs[0].calc_stat(), "build_s": round(t_build, 1),
                      "gap_operand_bytes": gap_bytes, "ms_min": round(mn, 3), "GBps_algorithmic": round(gap_bytes / mn / 1e6, 1),
                      "result_count": r.count(), "result_types": r.calc_stat(), "hbm_used": ctx.mem_used()}))
