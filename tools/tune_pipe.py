#!/usr/bin/env python3
"""Within-process interleaved A/B of the counts-pipeline launch shapes (headline workload or a block range of it).
Needs the tuning build for shapes outside the default set:  make -C bitmagic_amd/csrc tune;
  BMX_LIB=bitmagic_amd/lib/libbmx_tune.so python tools/tune_pipe.py [--nvec 256] [--rounds 7] [--shard 8]
variant = rows:unroll:nt:wg:swz:window[:lds]   (rows/unroll/wg/window 0 = plan default, window -1 = single launch)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bitmagic_amd as bm

ap = argparse.ArgumentParser()
ap.add_argument("--nvec", type=int, default=256)
ap.add_argument("--nbits", type=int, default=1_000_000_000)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--shard", type=int, default=1, help="run block columns shard_range(nblocks, 0, SHARD) only (what one of SHARD GPUs runs)")
ap.add_argument("--variants", type=str, default="")
ap.add_argument("--stream", action="store_true", help="also measure the plain streaming-read ceiling (tuning build only)")
a = ap.parse_args()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, a.nbits, with_common=True) for v in range(a.nvec)]
agg = bm.aggregator(ctx); pipe = bm.aggregator.pipeline(ctx); g = pipe.add()
for v in vecs: g.add(v, 0)
pipe.complete()
nblocks = vecs[0].info()["nblocks"]
lo, hi = bm.shard_range(nblocks, 0, a.shard)
counts = torch.zeros(1, dtype=torch.int64, device="cuda")
ob = pipe.operand_bytes(lo, hi)
if a.variants:
    variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]
    variants = [v if len(v) == 7 else v + (0,) for v in variants]
else:
    variants = [(0, 0, 1, 0, 1, 0, 0), (8, 4, 1, 384, 1, -1, 0), (8, 4, 1, 256, 1, -1, 0)]
    variants += [(8, 4, 1, wg, 1, w, 0) for wg, ws in ((256, (1024, 2048, 3072)), (384, (1536, 3072)), (192, (1536, 2304, 3072)), (512, (2048,)), (320, (1280, 2560)))
                 for w in ws]
    variants += [(8, 2, 1, 256, 1, 2048, 0), (8, 2, 1, 384, 1, 3072, 0), (4, 4, 1, 256, 1, 1024, 0), (4, 8, 1, 256, 1, 1024, 0)]
res = {v: [] for v in variants}
ref = None
for rnd in range(a.rounds):
    for v in variants:
        r, u, nt, wg, sw, win, ldsb = v
        for k, x in (("pipe_unroll", u), ("pipe_rows", r), ("pipe_nt", nt), ("pipe_wg", wg), ("xcd_swizzle", sw), ("pipe_window", win), ("pipe_lds", ldsb)):
            ctx.set_tuning(k, x)
        try:
            agg.run_counts_dev(pipe, counts.data_ptr(), lo, hi)
        except bm.BmxError as e:
            if rnd == 0: print("skip", v, e)
            res.pop(v, None); continue
        torch.cuda.synchronize()
        c = int(counts.item())
        if ref is None: ref = c
        assert c == ref, (v, c, ref)
        ctx.timer_start()
        for _ in range(a.iters): agg.run_counts_dev(pipe, counts.data_ptr(), lo, hi)
        res[v].append(ctx.timer_stop_ms() / a.iters)
    variants = [v for v in variants if v in res]
if a.stream:
    import ctypes as C
    from bitmagic_amd import _ffi
    if hasattr(_ffi.lib(), "bmx_diag_stream_read"):
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
print(f"blocks [{lo}, {hi}) of {nblocks}  count {ref}  operand GB {ob / 1e9:.3f}")
for med, mn, v in rows:
    print(f"rows={v[0]} U={v[1]} nt={v[2]} wg={v[3]} swz={v[4]} window={v[5]} lds={v[6]}  median {med:.4f} ms  min {mn:.4f} ms  {ob/med/1e6:.0f} GB/s  frac {ob/med/1e6/8000:.3f}")
