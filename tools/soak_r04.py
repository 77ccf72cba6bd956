"""Round-4 randomized differential soak (GPU): the kernels added this round against the oracle on random inputs --
(A) k_agg_or_rows (tile directories) over operands made every way a device vector can come to be (import, block-table
    upload, results of earlier operations, clones), random tile shapes, long runs, NULL / FULL blocks, ragged lengths;
(B) k_coll_members: random subsets / pipelines over prepared collections;
(C) k_op2_loop: materialised pairwise operations over long vectors of mixed block kinds, both optimisation modes;
(D) pipeline::set_search_count_limit on random pipelines;
(E) bmx_op2_dev / bmx_pending_wait: random chains over random block tables and long mixed vectors, waits in random order.
Usage: python tools/soak_r04.py [rounds]   (prints one FAIL line per difference, then "soak_r04 done, failures: N")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import numpy as np
import oracle, bitmagic_amd as bm
import test_gpu_parity as P

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 240
ONLY = sys.argv[2] if len(sys.argv) > 2 else "ABCDE"                 # parts to run
port = oracle.port()
bad = 0


def fail(*a):
    global bad
    bad += 1
    print("FAIL", *a, flush=True)


def kinds_equal(g, e):
    gk = g.block_table()[0].tolist()
    ek = (e.flatten()[0].tolist() + [0] * 16)[:len(gk)]
    return gk == ek


# ---------------------------------------------------------------- (A) row kernel
for seed in range(ROUNDS if "A" in ONLY else 0):
    rng = np.random.default_rng(410000 + seed)
    nblk = int(rng.integers(1, 48)); nvec = int(rng.integers(64, 360))
    dq = int(rng.choice([3, 13, 13, 40, 120, 260]))
    nbits = nblk * 65536 - int(rng.integers(0, 60000))
    words = P._sparse_collection(port, rng, nvec, nbits, dq, long_runs=bool(rng.integers(0, 2)), ragged=bool(rng.integers(0, 2)) and nblk >= 5,
                                 specials=bool(rng.integers(0, 2)))
    for v in range(0, nvec, int(rng.integers(3, 12))):                  # NULL stretches; blocks that start with a 1-run
        b0 = int(rng.integers(0, nblk)); words[v][b0 * 2048:(b0 + int(rng.integers(1, 4))) * 2048] = 0
        b1 = int(rng.integers(0, nblk))
        if (b1 + 1) * 2048 <= words[v].size: words[v][b1 * 2048] |= np.uint32(int(rng.integers(1, 16)))
    pv = [port.import_words(w, True, w.size * 32) for w in words]
    if any(p.flatten()[0].tolist().count(2) for p in pv):
        continue
    c = bm.context(0)
    c.set_tuning("gap_pack", 0); c.set_tuning("direct_cols", 0); c.set_tuning("pipe_split", 0); c.set_tuning("or_rows", 1)
    c.set_tuning("or_depth", 8 if seed % 3 == 0 else 4)
    gv = []
    for i, (w, p) in enumerate(zip(words, pv)):
        how = int(rng.integers(0, 4))
        if how == 0:
            k, o, b, g = p.flatten(); gv.append(bm.bvector.from_block_table(c, w.size * 32, k, o, b, g))
        elif how == 1 and i >= 2:                                       # a RESULT vector as an operand: OR of two earlier ones (opt_compress: GAP blocks again)
            j0, j1 = (int(x) for x in rng.integers(0, i, 2))
            r = bm.bvector._op2(bm.OR, gv[j0], gv[j1], bm.opt_compress)
            e = port.op2(1, pv[j0], pv[j1], True)
            if r.block_table()[0].tolist().count(2):
                gv.append(bm.bit_import_u32(c, w, True))
            else:
                gv.append(r); pv[i] = e
        else:
            gv.append(bm.bit_import_u32(c, w, True))
    nwb = (nblk + 1) * 2048
    agg = bm.aggregator(c)
    for t in range(4):
        m = int(rng.integers(64, nvec + 1))
        sel = rng.choice(nvec, size=m, replace=bool(t & 1)).tolist()
        opt = bool(rng.integers(0, 2))
        agg.set_optimization(opt)
        o = agg.combine_or([gv[i] for i in sel]); e = port.agg_or([pv[i] for i in sel], opt)
        if not ((o.to_words(nwb) == e.to_words(nwb)).all() and kinds_equal(o, e) and o.count() == e.count()):
            fail("rows", seed, t, nblk, nvec, dq, opt)
    del gv, o, agg
    c.close()
print("A done, failures so far:", bad, flush=True)

# ---------------------------------------------------------------- (B) member directory over prepared collections
for seed in range(max(ROUNDS // 2, 4) if "B" in ONLY else 0):
    rng = np.random.default_rng(420000 + seed)
    nblk = int(rng.integers(1, 20)); nvec = int(rng.integers(40, 300))
    dq = int(rng.choice([5, 13, 40, 150, 280]))
    nbits = nblk * 65536 - int(rng.integers(0, 60000))
    words = P._sparse_collection(port, rng, nvec, nbits, dq, long_runs=bool(rng.integers(0, 2)), ragged=nblk >= 5, specials=bool(rng.integers(0, 2)))
    pv = [port.import_words(w, True, w.size * 32) for w in words]
    if any(p.flatten()[0].tolist().count(2) for p in pv):
        continue
    c = bm.context(0)
    c.set_tuning("direct_cols", 0); c.set_tuning("pipe_split", 0); c.set_tuning("coll_members", 1)
    gv = [bm.bit_import_u32(c, w, True) for w in words]
    c.collection_prepare(gv, bm.ROLE_OR); c.collection_prepare(gv, bm.ROLE_AND)
    nwb = (nblk + 1) * 2048
    agg = bm.aggregator(c)
    for t in range(6):
        m = int(rng.integers(16, nvec + 1))
        sel = rng.choice(nvec, size=m, replace=bool(t % 3 == 2)).tolist()
        opt = bool(t & 1)
        agg.set_optimization(opt)
        o = agg.combine_or([gv[i] for i in sel]); e = port.agg_or([pv[i] for i in sel], opt)
        if not ((o.to_words(nwb) == e.to_words(nwb)).all() and kinds_equal(o, e)): fail("members or", seed, t, nblk, nvec, dq)
        agg.set_optimization(False)
        cut = int(rng.integers(1, max(2, m // 2)))
        r, any_ = agg.combine_and_sub([gv[i] for i in sel[:cut]], [gv[i] for i in sel[cut:]])
        e = port.agg_and_sub([pv[i] for i in sel[:cut]], [pv[i] for i in sel[cut:]])
        if not ((r.to_words(nwb) == e.to_words(nwb)).all() and kinds_equal(r, e) and any_ == (e.count() > 0)): fail("members and_sub", seed, t, nblk, nvec, dq, cut)
    groups = []
    for g in range(int(rng.integers(1, 90))):
        na = int(rng.integers(1, 50)); ns = int(rng.integers(0, 60))
        a = rng.choice(nvec, size=min(na, nvec), replace=False).tolist()
        s_ = [i for i in rng.choice(nvec, size=min(ns, nvec), replace=False).tolist() if i not in a]
        groups.append((a, s_))
    pipe = bm.aggregator.pipeline(c)
    for a, s_ in groups:
        ag = pipe.add()
        for i in a: ag.add(gv[i], 0)
        for i in s_: ag.add(gv[i], 1)
    pipe.complete()
    got = [int(x) for x in agg.combine_and_sub(pipe)]
    exp = [port.agg_and_sub([pv[i] for i in a], [pv[i] for i in s_]).count() for a, s_ in groups]
    if got != exp: fail("members pipeline", seed, nblk, nvec, dq, len(groups))
    if c.pack_stats()["collections"] != 2: fail("members: a collection was built on the side", seed)
    del gv, pipe, agg, o, r
    c.close()
print("B done, failures so far:", bad, flush=True)

# ---------------------------------------------------------------- (C) persistent materialising pairwise kernel
ctx = bm.context(0)
for seed in range(max(ROUNDS // 3, 3) if "C" in ONLY else 0):
    rng = np.random.default_rng(430000 + seed)
    nblk = int(rng.integers(2048, 2500)); nbits = nblk * 65536 - int(rng.integers(0, 60000))
    vs = []
    for v in range(3):
        dq = int(rng.choice([20, 300, 655, 655, 3000, 40000]))
        w = port.gen_words(8800 + seed, v, dq, nbits)
        for _ in range(int(rng.integers(0, 40))):                       # NULL / FULL / long-run stretches
            b = int(rng.integers(0, nblk - 1)); kind = int(rng.integers(0, 3))
            if kind == 0: w[b * 2048:(b + 1) * 2048] = 0
            elif kind == 1: w[b * 2048:(b + 1) * 2048] = 0xFFFFFFFF
            else: w[b * 2048 + 100:b * 2048 + int(rng.integers(101, 2048))] = 0xFFFFFFFF
        opt = bool(rng.integers(0, 4))
        vs.append((port.import_words(w, opt, nbits), bm.bit_import_u32(ctx, w, opt)))
    nwb = (nblk + 1) * 2048
    for op in range(4):
        i, j = (int(x) for x in rng.choice(3, size=2, replace=False))
        for oc in (False, True):
            t = bm.bvector._op2(op, vs[i][1], vs[j][1], bm.opt_compress if oc else bm.opt_none)
            e = port.op2(op, vs[i][0], vs[j][0], oc)
            if not ((t.to_words(nwb) == e.to_words(nwb)).all() and kinds_equal(t, e) and t.count() == e.count()):
                fail("op2_loop", seed, op, oc, i, j, nblk)
        if bm._count_op2(op, vs[i][1], vs[j][1]) != port.count_op2(op, vs[i][0], vs[j][0]): fail("count_op2_loop", seed, op)
    del vs, t
print("C done, failures so far:", bad, flush=True)

# ---------------------------------------------------------------- (D) search count limit
agg = bm.aggregator(ctx)
for seed in range(max(ROUNDS // 3, 3) if "D" in ONLY else 0):
    rng = np.random.default_rng(440000 + seed)
    nblk = int(rng.integers(300, 4000)); nbits = nblk * 65536
    nv = int(rng.integers(3, 9))
    gv = [bm.bvector.generate(ctx, 99 + seed, 40 + i, int(rng.choice([655, 6554, 20000])), nbits, with_common=True) for i in range(nv)]
    groups = []
    for g in range(int(rng.integers(1, 7))):
        a = rng.choice(nv, size=int(rng.integers(1, nv)), replace=False).tolist()
        s_ = [i for i in rng.choice(nv, size=int(rng.integers(0, 3)), replace=False).tolist() if i not in a]
        groups.append((a, s_))
    def run(limit):
        pipe = bm.aggregator.pipeline(ctx)
        for a, s_ in groups:
            ag = pipe.add()
            for i in a: ag.add(gv[i], 0)
            for i in s_: ag.add(gv[i], 1)
        if limit is not None: pipe.set_search_count_limit(limit)
        pipe.complete()
        return [int(x) for x in agg.combine_and_sub(pipe)], pipe.last_windows()
    full, _ = run(None)
    for limit in (1, int(rng.integers(2, 5000)), max(full) + 1):
        got, win = run(limit)
        if not all(min(limit, f) <= x <= f for x, f in zip(got, full)): fail("search limit", seed, limit, got, full, win)
    del gv
# ---------------------------------------------------------------- (E) asynchronous chains (bmx_op2_dev / bmx_pending_wait)
import test_gpu_stress as S
for seed in range(max(ROUNDS // 2, 4) if "E" in ONLY else 0):
    rng = np.random.default_rng(450000 + seed)
    nblk = int(rng.integers(1, 30)) if seed % 4 else int(rng.integers(2048, 2200))
    if nblk < 2048:
        base = [S._random_vector(rng, port, ctx, nblk, bool(rng.integers(0, 2))) for _ in range(4)]
    else:
        base = []
        for v in range(3):
            w = port.gen_words(9900 + seed, v, int(rng.choice([13, 655, 6554, 40000])), nblk * 65536)
            for _ in range(20):
                b = int(rng.integers(0, nblk)); w[b * 2048:(b + 1) * 2048] = 0 if rng.integers(0, 2) else 0xFFFFFFFF
            opt = bool(rng.integers(0, 4))
            base.append((port.import_words(w, opt, w.size * 32), bm.bit_import_u32(ctx, w, opt)))
    nw = (nblk + 1) * 2048
    exp = [b_[0] for b_ in base]; dev = [b_[1] for b_ in base]            # dev[i]: bvector or pending
    chain = []
    for step in range(int(rng.integers(2, 8))):
        i, j = (int(x) for x in rng.integers(0, len(dev), 2)); op = int(rng.integers(0, 4))
        chain.append((op, i, j))
        dev.append(bm.bvector.op2_async(op, dev[i], dev[j])); exp.append(port.op2(op, exp[i], exp[j], False))
    order = [k for k in range(len(base), len(dev))]; rng.shuffle(order)
    drop = order.pop() if len(order) > 2 and rng.integers(0, 3) == 0 else None   # one unresolved result is dropped instead of waited for
    for k in order:
        t = dev[k].wait(); e = exp[k]
        okw, okk, okc = bool((t.to_words(nw) == e.to_words(nw)).all()), kinds_equal(t, e), t.count() == e.count()
        if not (okw and okk and okc): fail("async chain", seed, k, nblk, "bits" if not okw else "", "kinds" if not okk else "", "count" if not okc else "", chain, len(base),
                                           t.block_table()[0].tolist()[:30], e.flatten()[0].tolist()[:30])
        dev[k] = t
    del dev, exp, base
print("E done, failures so far:", bad, flush=True)
print("soak_r04 done, failures:", bad, flush=True)
