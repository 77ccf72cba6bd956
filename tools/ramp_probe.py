#!/usr/bin/env python3
"""per-launch time of the headline kernel right after process start (clock ramp / DVFS probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, 1_000_000_000, with_common=True) for v in range(256)]
agg = bm.aggregator(ctx); pipe = bm.aggregator.pipeline(ctx); g = pipe.add()
for v in vecs: g.add(v, 0)
pipe.complete()
counts = torch.zeros(1, dtype=torch.int64, device="cuda")
ts = []
t0 = time.perf_counter()
for i in range(400):
    ctx.timer_start(); agg.run_counts_dev(pipe, counts.data_ptr()); ts.append(ctx.timer_stop_ms())
print("wall", time.perf_counter() - t0)
for i in range(0, 400, 10): print(i, " ".join(f"{x:.3f}" for x in ts[i:i+10]))
# back-to-back batches of 20
for rep in range(5):
    ctx.timer_start()
    for _ in range(20): agg.run_counts_dev(pipe, counts.data_ptr())
    print("batch20", ctx.timer_stop_ms() / 20)
