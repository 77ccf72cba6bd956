#!/usr/bin/env python3
"""the sparse_vector_scanner call pattern -- many AND-SUB arg-groups over the SAME
bit-plane vectors (bit-sliced equality search, src/bmsparsevec_algo.h:2400-2630).  Measures how much of the
shared-operand traffic the per-XCD L2 absorbs (logical operand bytes / time vs the HBM roofline)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bitmagic_amd as bm

ap = argparse.ArgumentParser()
ap.add_argument("--planes", type=int, default=32)
ap.add_argument("--groups", type=int, default=512)
ap.add_argument("--nbits", type=int, default=1_000_000_000)
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
planes = [bm.bvector.generate(ctx, 0xB17A61C, 500 + i, 32768, a.nbits) for i in range(a.planes)]   # 50 % planes
rng = np.random.default_rng(3)
pipe = bm.aggregator.pipeline(ctx)
for g in range(a.groups):
    x = int(rng.integers(0, 1 << min(a.planes, 62)))
    ag = pipe.add()
    for i in range(a.planes):
        ag.add(planes[i], 0 if (x >> (i % 62)) & 1 else 1)
    if not ag.arg_bv0: ag.add(planes[0], 0)
pipe.complete()
agg = bm.aggregator(ctx)
counts = torch.zeros(a.groups, dtype=torch.int64, device="cuda")
ob = pipe.operand_bytes()
ref_counts = None
for staged, swz, slots in ((0, 1, 16), (1, 1, 16), (1, 1, 8)):
    ctx.set_tuning("xcd_swizzle", swz); ctx.set_tuning("pipe_staged", staged); ctx.set_tuning("pipe_slots", slots)
    for _ in range(2): agg.run_counts_dev(pipe, counts.data_ptr())
    ts = []
    for _ in range(5):
        ctx.timer_start(); agg.run_counts_dev(pipe, counts.data_ptr()); ts.append(ctx.timer_stop_ms())
    ms = min(ts)
    torch.cuda.synchronize()
    if ref_counts is None: ref_counts = counts.clone()
    assert bool((counts == ref_counts).all().item()), "staged / unstaged counts differ"
    print(json.dumps({"pattern": "scanner", "staged": staged, "slots": slots, "planes": a.planes, "groups": a.groups, "nbits": a.nbits, "xcd_swizzle": swz,
                      "ms": round(ms, 3), "logical_operand_GB": round(ob / 1e9, 2), "logical_TBps": round(ob / ms / 1e9, 2),
                      "unique_operand_GB": round(a.planes * a.nbits / 8e9, 2), "queries_per_s": round(a.groups / ms * 1e3, 1),
                      "nonzero_groups": int((counts > 0).sum().item())}))
