"""Round-5 randomized differential soak (GPU): what this round added, against the oracle on random inputs --
(F) k_agg_and_rows: counts pipelines (several arg-groups, AND + SUB lists, block-range runs), the materialising
    combine_and_sub and results pipelines over random GAP-only collections -- sparse, dense (inverted) and long blocks,
    blocks starting with a 1-run, NULL / FULL blocks, ragged operand lengths, every launch shape;
(G) pipeline::set_search_count_limit per arg-group on random pipelines (bit-block and GAP-only operands): counts within
    [min(limit, true), true], the window / group bookkeeping consistent, results + counts runs = leading columns of the
    unlimited result.
Usage: python tools/soak_r05.py [rounds] [parts]   (one FAIL line per difference, then "soak_r05 done, failures: N")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import numpy as np
import oracle, bitmagic_amd as bm
import test_gpu_parity as P

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ONLY = sys.argv[2] if len(sys.argv) > 2 else "FG"
port = oracle.port()
bad = 0
ran = {"F": 0, "G": 0}


def fail(*a):
    global bad
    bad += 1
    print("FAIL", *a, flush=True)


def kinds_equal(g, e):
    gk = g.block_table()[0].tolist()
    ek = (e.flatten()[0].tolist() + [0] * 64)[:len(gk)]
    return gk == ek


# ---------------------------------------------------------------- (F) AND rows kernel
for seed in range(ROUNDS if "F" in ONLY else 0):
    rng = np.random.default_rng(510000 + seed)
    nblk = int(rng.integers(1, 14)); nvec = int(rng.integers(3, 330))
    dq = int(rng.choice([5, 13, 66, 197, 300]))
    nbits = nblk * 65536 - int(rng.integers(0, 60000))
    words = P._sparse_collection(port, rng, nvec, nbits, dq, long_runs=bool(rng.integers(0, 2)), ragged=bool(rng.integers(0, 2)) and nblk >= 5,
                                 specials=bool(rng.integers(0, 2)))
    for v in range(0, nvec, int(rng.integers(2, 9))):                   # dense (inverted) blocks, blocks starting with a 1-run
        b0 = int(rng.integers(0, nblk)); lo, hi = b0 * 2048, min((b0 + 1) * 2048, words[v].size)
        if hi - lo == 2048 and (b0 + 1) * 65536 <= nbits and rng.integers(0, 2): words[v][lo:hi] = ~words[v][lo:hi]
        b1 = int(rng.integers(0, nblk))
        if (b1 + 1) * 2048 <= words[v].size: words[v][b1 * 2048] |= np.uint32(int(rng.integers(1, 16)))
    pv = [port.import_words(w, True, w.size * 32) for w in words]
    if any(p.flatten()[0].tolist().count(2) for p in pv):
        continue
    ran["F"] += 1
    c = bm.context(0)
    c.set_tuning("direct_cols", 0); c.set_tuning("pipe_split", 0); c.set_tuning("and_rows", 1)
    c.set_tuning("and_rows_wg", int(rng.choice([128, 256, 512]))); c.set_tuning("and_rows_depth", int(rng.choice([2, 3, 4, 8]))); c.set_tuning("and_rows_nt", int(rng.integers(0, 2)))
    gv = [bm.bvector.from_block_table(c, w.size * 32, *p.flatten()) if i % 2 else bm.bit_import_u32(c, w, True) for i, (w, p) in enumerate(zip(words, pv))]
    groups = []
    for g in range(int(rng.integers(1, 6))):
        na = int(rng.integers(1, nvec + 1)); a = rng.choice(nvec, size=na, replace=False).tolist()
        rest = [i for i in range(nvec) if i not in a]
        ns = int(rng.integers(0, min(len(rest), 40) + 1)) if rest and rng.integers(0, 2) else 0
        s_ = rng.choice(rest, size=ns, replace=False).tolist() if ns else []
        groups.append((a, s_))
    exp = port.pipeline_counts([([pv[i] for i in a], [pv[i] for i in s_]) for a, s_ in groups])
    pipe = bm.aggregator.pipeline(c)
    for a, s_ in groups:
        ag = pipe.add()
        for i in a: ag.add(gv[i], 0)
        for i in s_: ag.add(gv[i], 1)
    pipe.complete()
    agg = bm.aggregator(c)
    if "k_agg_and_rows" not in pipe.describe(): fail("F: kernel not taken", seed, pipe.describe())
    got = agg.combine_and_sub(pipe)
    if not (got == exp).all(): fail("F counts", seed, nblk, nvec, dq, got, exp)
    cut = int(rng.integers(0, nblk + 1))
    parts = agg._run_pipeline(pipe, 0, cut).astype(np.int64) + agg._run_pipeline(pipe, cut, nblk).astype(np.int64)
    if not (parts == exp.astype(np.int64)).all(): fail("F block-range parts", seed, cut, parts, exp)
    a, s_ = groups[0]
    e = port.agg_and_sub([pv[i] for i in a], [pv[i] for i in s_])
    t, any_ = agg.combine_and_sub([gv[i] for i in a], [gv[i] for i in s_])
    nw = nblk * 2048
    if not (t.to_words(nw) == e.to_words(nw)).all() or not kinds_equal(t, e) or any_ != (e.count() != 0): fail("F materialised", seed, len(a), len(s_))
    if seed % 4 == 0:
        rp = bm.aggregator.pipeline(c, bm.agg_opt_bvect_and_counts)
        for a, s_ in groups:
            ag = rp.add()
            for i in a: ag.add(gv[i], 0)
            for i in s_: ag.add(gv[i], 1)
        rp.complete()
        res = agg.combine_and_sub(rp)
        for (a, s_), r, cnt in zip(groups, res, rp.get_bv_count_vector()):
            e = port.agg_and_sub([pv[i] for i in a], [pv[i] for i in s_])
            if int(cnt) != e.count() or (r is None) != (e.count() == 0) or (r is not None and not (r.to_words(nw) == e.to_words(nw)).all()):
                fail("F results pipeline", seed, len(a), len(s_))
    del gv, pipe
    c.close()

# ---------------------------------------------------------------- (G) search count limit per arg-group
ctx = bm.context(0)
agg = bm.aggregator(ctx)
for seed in range(max(ROUNDS // 3, 3) if "G" in ONLY else 0):
    rng = np.random.default_rng(540000 + seed)
    nblk = int(rng.integers(200, 3000)); nbits = nblk * 65536
    nv = int(rng.integers(4, 10))
    gap_only = bool(seed % 2)
    dqs = [int(rng.choice([13, 66, 197])) if gap_only else int(rng.choice([655, 6554, 20000])) for _ in range(nv)]
    gv = [bm.bvector.generate(ctx, 199 + seed, 40 + i, dqs[i], nbits, with_common=True) for i in range(nv)]
    gv.append(bm.bvector.generate(ctx, 199 + seed, 99, 2, nbits))       # a nearly empty vector: groups with it stay below any limit
    groups = []
    for g in range(int(rng.integers(2, 40))):
        a = rng.choice(nv + 1, size=int(rng.integers(1, nv)), replace=False).tolist()
        if gap_only: a = a * 3                                           # (long lists: the row kernel)
        s_ = [i for i in rng.choice(nv, size=int(rng.integers(0, 3)), replace=False).tolist() if i not in a]
        groups.append((a, s_))
    def mk(limit, opt=bm.agg_opt_only_counts):
        pipe = bm.aggregator.pipeline(ctx, opt)
        for a, s_ in groups:
            ag = pipe.add()
            for i in a: ag.add(gv[i], 0)
            for i in s_: ag.add(gv[i], 1)
        if limit is not None: pipe.set_search_count_limit(limit)
        pipe.complete()
        return pipe
    ran["G"] += 1
    full = [int(x) for x in agg.combine_and_sub(mk(None))]
    for limit in (1, int(rng.integers(2, 5000)), max(full) + 1):
        p = mk(limit)
        got = [int(x) for x in agg.combine_and_sub(p)]
        launched, planned = p.last_windows(); wg = p.last_window_groups()
        if not all(min(limit, f) <= x <= f for x, f in zip(got, full)): fail("G counts", seed, limit, got, full)
        if len(wg) != launched or wg[0] != len(groups) or any(b > a for a, b in zip(wg, wg[1:])): fail("G windows", seed, limit, launched, planned, wg)
        if limit > max(full) and (launched != planned or got != full): fail("G no group satisfied", seed, launched, planned)
        # groups still running in the last launched window are exactly those not yet satisfied before it (or nothing was left to do)
        unsat = sum(1 for x in got if x < limit)
        if launched == planned and wg[-1] < unsat: fail("G active bookkeeping", seed, limit, wg, unsat)
    if seed % 3 == 0:
        limit = int(rng.integers(2, 3000))
        pr = mk(limit, bm.agg_opt_bvect_and_counts); res = agg.combine_and_sub(pr); cnt = [int(x) for x in pr.get_bv_count_vector()]
        pf = mk(None, bm.agg_opt_bvect_and_counts); rf = agg.combine_and_sub(pf)
        for g in range(len(groups)):
            if (res[g] is None) != (cnt[g] == 0): fail("G results null", seed, g); continue
            if not (min(limit, full[g]) <= cnt[g] <= full[g]): fail("G results count", seed, g, cnt[g], full[g], limit)
            if res[g] is not None and (res[g].count() != cnt[g] or bm.count_and(res[g], rf[g]) != cnt[g]): fail("G results subset", seed, g)
    del gv
ctx.close()
print("cases run:", ran)
print("soak_r05 done, failures:", bad)
sys.exit(1 if bad else 0)
