#!/usr/bin/env python3
"""aggregator::combine_shift_right_and (sequence search) on 1e9-bit vectors.
 (a) 'dna': 4 symbol vectors at 25 %, random pattern of length n -- dies within ~8 operands (early exit);
 (b) 'dense': 90 % vectors, nothing dies: every operand block is read (n x 125 MB), the bandwidth case.
Times the host call (count-only mode: kernel + fan-in + 8-byte readback; materialised: + result vector)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bitmagic_amd as bm

ap = argparse.ArgumentParser()
ap.add_argument("--nbits", type=int, default=1_000_000_000)
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
agg = bm.aggregator(ctx)
rng = np.random.default_rng(11)
sym = [bm.bvector.generate(ctx, 0xB17A61C, 900 + i, 16384, a.nbits) for i in range(4)]
dense = [bm.bvector.generate(ctx, 0xB17A61C, 950 + i, 58982, a.nbits) for i in range(4)]
for name, vecs, lens in (("dna", sym, (8, 16, 32, 64)), ("dense", dense, (2, 8, 16, 32, 33, 64))):
    for n in lens:
        src = [vecs[int(x)] for x in rng.integers(0, 4, size=n)]
        res = {}
        for mode in ("count", "target"):
            agg.set_compute_count(mode == "count")
            for _ in range(2): t, f = agg.combine_shift_right_and(src)
            ts = []
            for _ in range(5):
                ctx.timer_start(); t, f = agg.combine_shift_right_and(src); ts.append(ctx.timer_stop_ms())
            res[mode] = min(ts)
            if mode == "count": cnt = agg.count()
            else: assert t.count() == cnt
        agg.set_compute_count(False)
        ob = n * ((a.nbits + 65535) // 65536) * 8192
        print(json.dumps({"pattern": name, "n": n, "nbits": a.nbits, "count": cnt, "ms_count": round(res["count"], 3),
                          "ms_target": round(res["target"], 3), "full_read_GB": round(ob / 1e9, 2),
                          "TBps_if_full_read": round(ob / res["count"] / 1e9, 2)}))
