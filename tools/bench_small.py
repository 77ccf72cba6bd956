#!/usr/bin/env python3
"""Latency of whole host calls on SMALL collections (BASELINE configs[0] scale: 1 Mbit vectors), where launch
and row-building latency dominate: combine_and_sub / combine_or / find_first_and_sub over 256 vectors, pairwise ops."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bitmagic_amd as bm
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
def t(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 4)
for nbits in (1_000_000, 16_000_000, 100_000_000):
    vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, nbits, with_common=True) for v in range(256)]
    agg = bm.aggregator(ctx)
    out = {"nbits": nbits, "nvec": 256}
    out["combine_and_ms"] = t(lambda: agg.combine_and_sub(vecs, []))
    out["combine_or_ms"] = t(lambda: agg.combine_or(vecs))
    out["find_first_ms"] = t(lambda: agg.find_first_and_sub(vecs, []))
    own = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, nbits, with_common=False) for v in range(256)]   # AND dies after a few operands
    out["combine_and_disjoint_ms"] = t(lambda: agg.combine_and_sub(own, []))
    out["find_first_disjoint_ms"] = t(lambda: agg.find_first_and_sub(own, []))
    del own
    def mk():
        p = bm.aggregator.pipeline(ctx); g = p.add()
        for v in vecs: g.add(v, 0)
        p.complete(); return p
    out["pipeline_complete_ms"] = t(mk, reps=10, warm=2)
    out["bit_and_ms"] = t(lambda: bm.bvector.bit_and(vecs[0], vecs[1]))
    out["count_and_ms"] = t(lambda: bm.count_and(vecs[0], vecs[1]))
    print(json.dumps(out))
# find_first_and_sub at full size: the first hit decides the cost (ascending launch windows), 1e9-bit vectors
for nv, label in ((256, "256 operands"), (4, "4 operands")):
    nbits = 1_000_000_000
    vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, nbits, with_common=True) for v in range(nv)]
    agg = bm.aggregator(ctx)
    out = {"find_first_1e9": label}
    out["hit_in_block_0_ms"] = t(lambda: agg.find_first_and_sub(vecs, []), reps=10, warm=2)
    agg.set_range_hint(7000 * 65536, nbits - 1)
    out["hit_in_block_7000_of_15259_ms"] = t(lambda: agg.find_first_and_sub(vecs, []), reps=10, warm=2)
    agg.reset_range_hint()
    ctx.set_tuning("ff_window", -1)
    out["one_launch_hit_in_block_0_ms"] = t(lambda: agg.find_first_and_sub(vecs, []), reps=10, warm=2)
    ctx.set_tuning("ff_window", 0)
    out["combine_and_ms"] = t(lambda: agg.combine_and_sub(vecs, []), reps=5, warm=1)
    print(json.dumps(out))
    del vecs
# short operand lists (one wave per column straight from the descriptor tables) against the row-table pipeline
for nbits in (1_000_000, 1_000_000_000):
    for nv in (2, 4, 16):
        vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, nbits, with_common=True) for v in range(nv)]
        agg = bm.aggregator(ctx)
        out = {"short_lists": nv, "nbits": nbits}
        for name, dc in (("direct", 384), ("rows", 0)):
            ctx.set_tuning("direct_cols", dc)
            out["combine_and_%s_ms" % name] = t(lambda: agg.combine_and_sub(vecs, []), reps=10, warm=2)
            out["combine_or_%s_ms" % name] = t(lambda: agg.combine_or(vecs), reps=10, warm=2)
        ctx.set_tuning("direct_cols", 384)
        print(json.dumps(out))
        del vecs
# materialised combine_and at full size: windowed one-workgroup-per-CU launches against one launch of 256-thread workgroups
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 6554, 1_000_000_000, with_common=True) for v in range(256)]
agg = bm.aggregator(ctx)
out = {"combine_and_256x1e9": "materialised"}
for name, win in (("windows_ms", 0), ("one_launch_ms", -1)):
    ctx.set_tuning("pipe_window", win)
    out[name] = t(lambda: agg.combine_and_sub(vecs, []), reps=8, warm=2)
ctx.set_tuning("pipe_window", 0)
print(json.dumps(out))
del vecs
# materialised combine_and over 256 GAP-only vectors (0.3 %): counting formulation against the run-by-run kernel
vecs = [bm.bvector.generate(ctx, 0xB17A61C, v, 197, 1_000_000_000, with_common=True) for v in range(256)]
agg = bm.aggregator(ctx)
out = {"combine_and_256x1e9_gap_only": "materialised"}
for name, gc in (("counting_ms", -1), ("run_by_run_ms", 0)):
    ctx.set_tuning("gap_count", gc)
    out[name] = t(lambda: agg.combine_and_sub(vecs, []), reps=8, warm=2)
    out[name.replace("_ms", "_count")] = agg.combine_and_sub(vecs, [])[0].count()
ctx.set_tuning("gap_count", -1)
print(json.dumps(out))
del vecs
