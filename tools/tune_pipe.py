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
    variants = [v if len(v) >= 6 else v + (1,) for v in variants]
    variants = [v if len(v) == 7 else v + (0,) for v in variants]
else:
    variants = [(u, 8, nt, 256, 1, ver, 0) for u in (1, 2, 4) for nt in (0, 1) for ver in (1, 2)] + [(2, 4, 1, 256, 1, 1, 0), (4, 8, 1, 256, 0, 2, 0)]
    variants += [(4, 8, 1, wg, 1, 2, 0) for wg in (192, 320, 384, 448, 512, 640, 768)]
res = {v: [] for v in variants}
ref = None
for rnd in range(a.rounds):
    for v in variants:
        u, r, nt, wg, sw, ver, ldsb = v
        for k, x in (("pipe_unroll", u), ("pipe_rows", r), ("pipe_nt", nt), ("pipe_wg", wg), ("xcd_swizzle", sw), ("pipe_ver", ver), ("pipe_lds", ldsb)):
            ctx.set_tuning(k, x)
        agg.run_counts_dev(pipe, counts.data_ptr())
        torch.cuda.synchronize()
        c = int(counts.item())
        if ref is None: ref = c
        assert c == ref, (v, c, ref)
        ctx.timer_start()
        for _ in range(a.iters): agg.run_counts_dev(pipe, counts.data_ptr())
        res[v].append(ctx.timer_stop_ms() / a.iters)
import ctypes as C
from bitmagic_amd import _ffi
for swz in (1, 0):
  ctx.set_tuning("xcd_swizzle", swz)
  for pattern in (0, 1):
    for bpw in (1, 16, 256):
        ms = C.c_float()
        _ffi.check(_ffi.lib().bmx_diag_stream_read(ctx._h, 16 << 30, 1, bpw, pattern, 5, C.byref(ms)))
        print(f"stream_read 16 GiB nt=1 swz={swz} pattern={pattern} blocks_per_wave={bpw}: {ms.value:.4f} ms  {(16 << 30) / ms.value / 1e6:.0f} GB/s  frac {(16 << 30) / ms.value / 1e6 / 8000:.3f}")
rows = []
for v, t in res.items():
    t = np.array(t)
    rows.append((float(np.median(t)), float(t.min()), v))
rows.sort()
print("count", ref, "operand GB", ob / 1e9)
for med, mn, v in rows:
    print(f"ver={v[5]} U={v[0]} rows={v[1]} nt={v[2]} wg={v[3]} swz={v[4]} lds={v[6]}  median {med:.4f} ms  min {mn:.4f} ms  {ob/med/1e6:.0f} GB/s  frac {ob/med/1e6/8000:.3f}")
