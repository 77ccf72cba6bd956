#!/usr/bin/env python3
"""Within-process interleaved A/B of the counts-pipeline launch shapes (headline workload).
usage: python tools/tune_pipe.py [--nvec 256] [--rounds 7]"""
import argparse, itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bitmagic_amd as bm

ap = argparse.ArgumentParser()
ap.add_argument("--nvec", type=int, default=256)
ap.add_argument("--nbits", type=int, default=1_000_000_000)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--variants", type=str, default="")
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, a.nbits, with_common=True) for v in range(a.nvec)]
agg = bm.aggregator(ctx); pipe = bm.aggregator.pipeline(ctx); g = pipe.add()
for v in vecs: g.add(v, 0)
pipe.complete()
counts = torch.zeros(1, dtype=torch.int64, device="cuda")
ob = pipe.operand_bytes()
if a.variants:
    variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]
else:
    variants = [(u, r, nt, wg, sw) for u in (1, 2, 4) for r in (8, 4, 2) for nt in (0, 1) for wg in (256,) for sw in (1,)]
    variants += [(1, 8, 0, 64, 1), (1, 8, 0, 128, 1), (2, 4, 0, 64, 1), (1, 8, 0, 256, 0), (2, 4, 0, 256, 0), (1, 1, 0, 256, 1), (4, 1, 0, 256, 1)]
res = {v: [] for v in variants}
ref = None
for rnd in range(a.rounds):
    for v in variants:
        u, r, nt, wg, sw = v
        for k, x in (("pipe_unroll", u), ("pipe_rows", r), ("pipe_nt", nt), ("pipe_wg", wg), ("xcd_swizzle", sw)):
            ctx.set_tuning(k, x)
        agg.run_counts_dev(pipe, counts.data_ptr())
        torch.cuda.synchronize()
        c = int(counts.item())
        if ref is None: ref = c
        assert c == ref, (v, c, ref)
        ctx.timer_start()
        for _ in range(a.iters): agg.run_counts_dev(pipe, counts.data_ptr())
        res[v].append(ctx.timer_stop_ms() / a.iters)
rows = []
for v, t in res.items():
    t = np.array(t)
    rows.append((float(np.median(t)), float(t.min()), v))
rows.sort()
print("count", ref, "operand GB", ob / 1e9)
for med, mn, v in rows:
    print(f"U={v[0]} rows={v[1]} nt={v[2]} wg={v[3]} swz={v[4]}  median {med:.4f} ms  min {mn:.4f} ms  {ob/med/1e6:.0f} GB/s  frac {ob/med/1e6/8000:.3f}")
