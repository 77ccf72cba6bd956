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
xs = []
for g in range(a.groups):
    x = int(rng.integers(0, 1 << min(a.planes, 62)))
    xs.append(x)
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

# ---- the same batch of equality searches in ONE pass over the planes: bit-matrix transposition + hash lookup ----
import time
sc = bm.slice_scanner(ctx, planes, size=a.nbits)
if a.planes <= 32:
    for nq, eq_big, shape in ((a.groups, 0, 1), (a.groups, -1, 2), (2048, 0, 1), (2048, 1, 1), (2048, 1, 2), (4096, -1, 1), (4096, -1, 2), (8192, 0, 1), (8192, -1, 0), (8192, -1, 1), (8192, -1, 2), (16384, -1, 1), (16384, -1, 2)):
        ctx.set_tuning("eq_big", eq_big); ctx.set_tuning("eq_big_shape", shape)
        q = xs[:nq] if nq <= len(xs) else xs + [int(v) for v in rng.integers(1, 1 << a.planes, size=nq - len(xs))]
        got = sc.find_eq_counts(q)
        ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); got = sc.find_eq_counts(q); ts.append((time.perf_counter() - t0) * 1e3)
        ok = bool((np.asarray(got[:len(xs)], np.int64) == ref_counts.cpu().numpy()[:min(nq, len(xs))]).all()) if all(x > 0 for x in xs) else None
        nu = len(set(q))
        big = eq_big == 1 or (eq_big < 0 and (nu > 2048 or shape == 2))
        passes = -(-nu // ((8704 if shape == 2 else 9216) if big else 2048))
        print(json.dumps({"pattern": "scanner_transposed", "planes": a.planes, "queries": nq, "unique": nu, "nbits": a.nbits,
                          "table": ("k_slice_eq_counts_big<%s>" % ("18,512,768 threads,3 waves" if shape == 2 else "18,512" if shape else "17,1024")) if big else "k_slice_eq_counts", "passes_over_the_planes": passes,
                          "host_call_ms": round(min(ts), 3), "queries_per_s": round(nq / min(ts) * 1e3, 1), "plane_GB": round(a.planes * a.nbits / 8e9, 2),
                          "plane_TBps": round(a.planes * a.nbits / 8 / min(ts) / 1e9 * passes, 2), "counts_equal_pipeline": ok}))
    ctx.set_tuning("eq_big", -1); ctx.set_tuning("eq_big_shape", 2)

# a 12-plane container (values < 4096, every row non-zero): the 16-plane instantiation
if a.planes >= 12:
    sc12 = bm.slice_scanner(ctx, planes[:12], size=a.nbits)
    q = [int(v) for v in rng.integers(1, 1 << 12, size=2048)]
    for eq_big, shape in ((-1, 2), (1, 2)):
        ctx.set_tuning("eq_big", eq_big); ctx.set_tuning("eq_big_shape", shape)
        got = sc12.find_eq_counts(q); ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); got = sc12.find_eq_counts(q); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"pattern": "scanner_transposed", "planes": 12, "queries": 2048, "nbits": a.nbits,
                          "table": "k_slice_eq_counts<16>" if eq_big <= 0 else "k_slice_eq_counts_big<16,18,512,768 threads,3 waves>",
                          "host_call_ms": round(min(ts), 3), "sum_counts": int(np.asarray(got, np.int64).sum())}))
    ctx.set_tuning("eq_big", -1); ctx.set_tuning("eq_big_shape", 2)

# ---- range search (find_gt / find_le / find_range / find_zero): one pass over the planes (bmx_slice_compare) ----
plane_bytes = a.planes * ((a.nbits + 65535) // 65536) * 8192
for name, fn, cnt_fn in (("find_gt", lambda v: sc.find_gt(v), lambda v: sc.count(bm.CMP_GT, v)),
                         ("find_le", lambda v: sc.find_le(v), lambda v: sc.count(bm.CMP_LE, v)),
                         ("find_range", lambda v: sc.find_range(v >> 1, v), lambda v: sc.count(bm.CMP_RANGE, v >> 1, v))):
    vals = [int(rng.integers(1 << (a.planes - 2), 1 << (a.planes - 1))) for _ in range(6)]
    for v in vals[:2]: cnt_fn(v)
    ctx.synchronize()
    t0 = time.perf_counter()
    cs = [cnt_fn(v) for v in vals]
    t_cnt = (time.perf_counter() - t0) / len(vals) * 1e3
    t0 = time.perf_counter()
    rs = [fn(v) for v in vals]
    ctx.synchronize()
    t_mat = (time.perf_counter() - t0) / len(vals) * 1e3
    assert [r.count() for r in rs] == cs
    # algorithmic bytes of a comparison search: the plane blocks the walk had to read before every row of a block column was
    # decided (it stops in a column once no row is "equal so far"), counted by the kernel itself (bmx_slice_compare_stat)
    if name == "find_range":
        alg = None                        # (the stat entry takes one bound; the two-bound walk reads at least as much)
    else:
        pred = bm.CMP_GT if name == "find_gt" else bm.CMP_LE
        alg = sum(sc.compare_stat(pred, v)[1] for v in vals) / len(vals)
    print(json.dumps({"pattern": "range_search", "op": name, "planes": a.planes, "rows": a.nbits, "count_only_ms": round(t_cnt, 3),
                      "materialised_ms": round(t_mat, 3), "all_planes_GB": round(plane_bytes / 1e9, 2),
                      "algorithmic_GB": None if alg is None else round(alg / 1e9, 3),
                      "GBps": None if alg is None else round(alg / t_cnt / 1e6, 1),
                      "frac_of_8TBps": None if alg is None else round(alg / t_cnt / 1e6 / 8000.0, 4), "example_count": cs[0]}))
