#!/usr/bin/env python3
"""Full-size headline pass issued as a sequence of block-range launches (chunk columns each): does keeping
every wave in flight on the SAME contiguous column window beat one launch over all 15,259 columns?
BMX_LIB=bitmagic_amd/lib/libbmx_tune.so python tools/chunk_probe.py"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bitmagic_amd as bm
ap = argparse.ArgumentParser()
ap.add_argument("--nvec", type=int, default=256)
ap.add_argument("--chunks", default="0,1024,1536,1908,2048,2560,3072,4096,6144")
ap.add_argument("--wgs", default="256,384,192,128")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, 1_000_000_000, with_common=True) for v in range(a.nvec)]
agg = bm.aggregator(ctx); pipe = bm.aggregator.pipeline(ctx); g = pipe.add()
for v in vecs: g.add(v, 0)
pipe.complete()
nb = vecs[0].info()["nblocks"]
ob = pipe.operand_bytes()
counts = torch.zeros(64, dtype=torch.int64, device="cuda")
ctx.set_tuning("pipe_rows", 8); ctx.set_tuning("pipe_unroll", 4)
for wg in [int(x) for x in a.wgs.split(",")]:
    ctx.set_tuning("pipe_wg", wg)
    for ch in [int(x) for x in a.chunks.split(",")]:
        ranges = [(0, nb)] if ch == 0 else [(lo, min(nb, lo + ch)) for lo in range(0, nb, ch)]
        def run():
            for i, (lo, hi) in enumerate(ranges):
                agg.run_counts_dev(pipe, counts.data_ptr() + 8 * (i % 64), lo, hi)
        try:
            run()
        except bm.BmxError as e:
            print("skip wg", wg, e); break
        ctx.synchronize()
        ts = []
        for _ in range(a.rounds):
            ctx.timer_start()
            for _ in range(3): run()
            ts.append(ctx.timer_stop_ms() / 3)
        t = float(np.median(ts))
        print(f"wg={wg} chunk={ch} launches={len(ranges)}  {t:.4f} ms  {ob / t / 1e6:.0f} GB/s  frac {ob / t / 8e9:.4f}", flush=True)
