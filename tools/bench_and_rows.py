#!/usr/bin/env python3
"""Round 5: first-call 256-way AND+COUNT over GAP-only operands (configs[2]-shaped, sparse): k_agg_and_rows launch shapes
against the round-4 kernels on the same box, counts pipeline (ctx timer around the asynchronous run) and the materialised
combine_and host call.  One JSON line per (density, variant)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bitmagic_amd as bm

SEED = 0xB17A61C
ctx = bm.context(0)
nbits = int(os.environ.get("NBITS", 1_000_000_000)); nvec = int(os.environ.get("NVEC", 256))
dqs = [int(x) for x in os.environ.get("DQS", "197,66").split(",")]
DIAG = "tune" in os.environ.get("BMX_LIB", "")
variants = [("gapcount_r4", dict(and_rows=0))]
for nt in (0, 1):
    for wg in (128, 256, 512):
        for dp in (2, 3, 4):
            variants.append(("rows_%d_%d_nt%d" % (wg, dp, nt), dict(and_rows=1, and_rows_wg=wg, and_rows_depth=dp, and_rows_nt=nt)))
if os.environ.get("VARS"): variants = [v for v in variants if v[0] in os.environ["VARS"].split(",")]
if DIAG:
    variants = [(n + "_diag%d" % d, dict(k, diag=d)) for d in (0, 1, 3, 4, 8) for n, k in variants if n in ("rows_256_2_nt0",)]
import ctypes as C
from bitmagic_amd import _ffi
L = _ffi.lib()
d_counts = C.c_void_p()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc(C.byref(d_counts), 4096)
for dq in dqs:
    vecs = [bm.bvector.generate(ctx, SEED, v, dq, nbits, with_common=os.environ.get("COMMON", "1") != "0") for v in range(nvec)]     # COMMON=0: independent operands (the AND is empty)
    st = vecs[0].calc_stat()
    agg = bm.aggregator(ctx)
    pipe = bm.aggregator.pipeline(ctx); g = pipe.add()
    for v in vecs: g.add(v, 0)
    pipe.complete()
    alg = pipe.operand_bytes()
    ref = None
    for name, knobs in variants:
        for k, v in knobs.items():
            if k == "diag": os.environ["BMX_DIAG_AROWS"] = str(v)
            else: ctx.set_tuning(k, v)
        cnt = int(agg.combine_and_sub(pipe)[0])
        if ref is None: ref = cnt
        ts = []
        for _ in range(3): agg.run_counts_dev(pipe, d_counts.value)
        ctx.synchronize()
        for _ in range(10):
            ctx.timer_start(); agg.run_counts_dev(pipe, d_counts.value); ts.append(ctx.timer_stop_ms())
        ms = float(np.median(ts))
        os.environ["BMX_DIAG_AROWS"] = "0"
        t0 = time.perf_counter(); r, _ = agg.combine_and_sub(vecs, []); host_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); r2, _ = agg.combine_and_sub(vecs, []); host_ms = min(host_ms, (time.perf_counter() - t0) * 1e3)
        print(json.dumps({"dq": dq, "variant": name, "kernel": pipe.describe(), "stat": st, "alg_GB": round(alg / 1e9, 3), "ms": round(ms, 4),
                          "frac": round(alg / (ms * 1e-3) / 8e12, 4), "count": cnt, "count_ok": cnt == ref, "materialised_host_ms": round(host_ms, 3),
                          "materialised_count": r.count()}), flush=True)
        del r, r2
    del pipe, vecs
ctx.set_tuning('and_rows', -1)
agg = bm.aggregator(ctx)

# crossover against the wave-per-item kernels: 16 arg-groups of n operands each (distinct vectors), counts pipeline
if os.environ.get("CROSS"):
    for k, v in (("and_rows_wg", 256), ("and_rows_depth", 3), ("and_rows_nt", 0)): ctx.set_tuning(k, v)
    for dq in dqs:
        vecs = [bm.bvector.generate(ctx, SEED, v, dq, nbits, with_common=os.environ.get("COMMON", "1") != "0") for v in range(nvec)]     # COMMON=0: independent operands (the AND is empty)
        for n in (2, 4, 8, 12, 16, 24, 32, 64):
            ng = min(16, nvec // n)
            pipe = bm.aggregator.pipeline(ctx)
            for gi in range(ng):
                g = pipe.add()
                for v in vecs[gi * n:(gi + 1) * n]: g.add(v, 0)
            pipe.complete()
            out = {"dq": dq, "ops_per_group": n, "groups": ng, "alg_GB": round(pipe.operand_bytes() / 1e9, 3)}
            for name, ar, wg, ipw in (("rows128", 1, 256, 4), ("rows", 1, 256, 1), ("rows512", 1, 256, 8), ("older", 0, 256, 0)):
                ctx.set_tuning("and_rows", ar); ctx.set_tuning("and_rows_wg", wg); ctx.set_tuning("and_rows_ipw", ipw)
                cnt = agg.combine_and_sub(pipe).copy()
                for _ in range(2): agg.run_counts_dev(pipe, d_counts.value)
                ctx.synchronize(); ts = []
                for _ in range(5):
                    ctx.timer_start(); agg.run_counts_dev(pipe, d_counts.value); ts.append(ctx.timer_stop_ms())
                out[name + "_ms"] = round(float(np.median(ts)), 4); out[name + "_kernel"] = pipe.describe()[:40]; out[name + "_sum"] = int(cnt.sum())
            print(json.dumps(out), flush=True)
            del pipe
        del vecs
