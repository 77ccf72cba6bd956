mkdir -p gpurun_out
cat > /tmp/soak.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle, bitmagic_amd as bm
import test_gpu_stress as S, test_gpu_parity as P
ctx = bm.context(0); port = oracle.port()
bad = 0
for seed in range(60, 1500):
    try: S.test_random_block_tables(ctx, port, seed, "direct")
    except AssertionError as e: bad += 1; print("FAIL seed", seed, str(e)[:200])
for args in [(2, 50, 45), (17, 80, 64), (90, 60, 33), (333, 150, 70), (600, 20, 40), (250, 300, 37)]:
    try: P.test_sparse_state_of_gap_lists(ctx, port, *args, "direct")
    except AssertionError as e: bad += 1; print("FAIL sparse", args, str(e)[:200])
for dq, nv in [(7, 130), (100, 64), (65400, 40), (2000, 35)]:
    try: P.test_many_gap_operands(ctx, port, dq, nv, "direct")
    except AssertionError as e: bad += 1; print("FAIL many", dq, nv, str(e)[:200])
import numpy as np
agg = bm.aggregator(ctx)
for seed in range(400):
    rng = np.random.default_rng(50000 + seed)
    nblk = int(rng.integers(1, 7))
    vecs = [S._random_vector(rng, port, ctx, nblk, True) for _ in range(int(rng.integers(1, 6)))]
    n = int(rng.integers(1, 45))
    sel = [int(x) for x in rng.integers(0, len(vecs), n)]
    opt, any_ = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    agg.set_optimization(opt)
    t, f = agg.combine_shift_right_and([vecs[i][1] for i in sel], any=any_)
    e, ef = port.agg_shift_right_and([vecs[i][0] for i in sel], opt, any_)
    nw = nblk * 2048
    ok = f == ef and (t.to_words(nw) == e.to_words(nw)).all() and t.block_table()[0].tolist()[:nblk] == e.flatten()[0].tolist()[:nblk]
    if not ok: bad += 1; print("FAIL shift seed", seed, n, opt, any_)
agg.set_optimization(False)
# block kinds (representation) of compressed results vs the oracle
for seed in range(400):
    rng = np.random.default_rng(70000 + seed)
    nblk = int(rng.integers(1, 7)); nv = int(rng.integers(2, 7))
    vecs = [S._random_vector(rng, port, ctx, nblk, bool(rng.integers(0, 2))) for _ in range(nv)]
    pv = [v[0] for v in vecs]; gv = [v[1] for v in vecs]
    for op in range(4):
        i, j = (int(x) for x in rng.integers(0, nv, 2))
        for oc in (True, False):
            kk = bm.bvector._op2(op, gv[i], gv[j], bm.opt_compress if oc else bm.opt_none).block_table()[0].tolist()
            ek = port.op2(op, pv[i], pv[j], oc).flatten()[0].tolist()
            if kk != ek:
                bad += 1; print("FAIL kinds op2", seed, op, oc, i, j, kk, ek, pv[i].flatten()[0].tolist(), pv[j].flatten()[0].tolist())
    na = int(rng.integers(1, nv + 1))
    t, _ = agg.combine_and_sub(gv[:na], gv[na:])
    e = port.agg_and_sub(pv[:na], pv[na:])
    if t.block_table()[0].tolist() != e.flatten()[0].tolist(): bad += 1; print("FAIL kinds and_sub", seed)
    for oc in (True, False):
        agg.set_optimization(oc)
        if agg.combine_or(gv).block_table()[0].tolist() != port.agg_or(pv, oc).flatten()[0].tolist(): bad += 1; print("FAIL kinds or", seed, oc)
    agg.set_optimization(False)
    # pipeline with result vectors, counts and an OR target: presence, kinds and content of every result
    groups = []
    for _ in range(int(rng.integers(1, 5))):
        a = [int(x) for x in rng.integers(0, nv, int(rng.integers(0, nv + 1)))]
        s_ = [int(x) for x in rng.integers(0, nv, int(rng.integers(0, nv)))]
        groups.append((a, s_))
    pipe = bm.aggregator.pipeline(ctx, bm.agg_opt_bvect_and_counts)
    pipe.set_or_target(None)
    for a, s_ in groups:
        ag = pipe.add()
        for k in a: ag.add(gv[k], 0)
        for k in s_: ag.add(gv[k], 1)
    pipe.complete()
    agg.combine_and_sub(pipe)
    pres, pcnt, port_or = port.pipeline_results([([pv[k] for k in a], [pv[k] for k in s_]) for a, s_ in groups])
    res = pipe.get_bv_res_vector()
    nw = nblk * 2048
    okp = [r is None for r in res] == [r is None for r in pres] and [int(x) for x in pipe.get_bv_count_vector()] == [int(x) for x in pcnt]
    for r, e in zip(res, pres):
        if r is not None and e is not None:
            okp = okp and (r.to_words(nw) == e.to_words(nw)).all() and r.block_table()[0].tolist()[:nblk] == e.flatten()[0].tolist()[:nblk]
    go = pipe.get_or_target()
    gk = go.block_table()[0].tolist()[:nblk]; gk += [0] * (nblk - len(gk))     # (an untouched OR target has no blocks at all)
    pk = port_or.flatten()[0].tolist()[:nblk]; pk += [0] * (nblk - len(pk))
    okp = okp and (go.to_words(nw) == port_or.to_words(nw)).all() and gk == pk
    if not okp: bad += 1; print("FAIL pipeline results", seed, groups)
# small collections through the one-launch path (k_direct): 24..300 operands drawn (with repeats) from a pool of random
# vectors, 1..6 block columns: combine_and_sub / combine_or / find_first_and_sub -- content, kinds, first bit
for seed in range(300):
    rng = np.random.default_rng(90000 + seed)
    nblk = int(rng.integers(1, 7)); npool = int(rng.integers(3, 12))
    pool = [S._random_vector(rng, port, ctx, nblk, bool(rng.integers(0, 4))) for _ in range(npool)]
    n = int(rng.integers(24, 300))
    # AND groups that survive need correlated operands: mostly repeats of a few vectors
    hot = [int(x) for x in rng.integers(0, npool, int(rng.integers(1, 4)))]
    a = [hot[int(x)] for x in rng.integers(0, len(hot), n)]
    s_ = [int(x) for x in rng.integers(0, npool, int(rng.integers(0, 40)))] if rng.integers(0, 2) else []
    nw = nblk * 2048
    t, f = agg.combine_and_sub([pool[i][1] for i in a], [pool[i][1] for i in s_])
    e = port.agg_and_sub([pool[i][0] for i in a], [pool[i][0] for i in s_])
    ok = (t.to_words(nw) == e.to_words(nw)).all() and t.block_table()[0].tolist()[:nblk] == e.flatten()[0].tolist()[:nblk] and f == (e.count() != 0)
    ff = agg.find_first_and_sub([pool[i][1] for i in a], [pool[i][1] for i in s_])
    ef = port.find_first_and_sub([pool[i][0] for i in a], [pool[i][0] for i in s_])
    ok = ok and ff[0] == ef[0] and (not ef[0] or ff[1] == ef[1])
    sel = [int(x) for x in rng.integers(0, npool, n)]
    for oc in (True, False):
        agg.set_optimization(oc)
        o = agg.combine_or([pool[i][1] for i in sel]); eo = port.agg_or([pool[i][0] for i in sel], oc)
        ok = ok and (o.to_words(nw) == eo.to_words(nw)).all() and o.block_table()[0].tolist()[:nblk] == eo.flatten()[0].tolist()[:nblk]
    agg.set_optimization(False)
    if not ok: bad += 1; print("FAIL direct", seed, nblk, npool, n, len(s_))
# rank / select over sharded vectors (3 members on this GPU) vs the single-device index
grp = bm.group([0, 0, 0])
for seed in range(60):
    rng = np.random.default_rng(95000 + seed)
    nblk = int(rng.integers(1, 12))
    p, v = S._random_vector(rng, port, ctx, nblk, bool(rng.integers(0, 2)))
    k, o, b, gp = p.flatten()
    nbits = v.info()["nbits"]
    g = bm.gbvector.from_block_table(grp, nbits, k, o, b, gp)
    grs, rs = g.build_rs_index(), v.build_rs_index()
    q = rng.integers(0, nbits + 70000, size=2000).astype(np.uint64)
    c = rs.count()
    r = rng.integers(0, c + 3, size=2000).astype(np.uint64)
    gf, gpos = g.select(r, grs); sf, spos = v.select(r, rs)
    ok = grs.count() == c and (g.rank(q, grs) == v.rank(q, rs)).all() and (gf == sf).all() and (gpos[gf] == spos[sf]).all()
    if not ok: bad += 1; print("FAIL group rank/select", seed, nblk)
    del grs, g
grp.close()
# batched equality counts by transposition (bmx_slice_eq_counts) vs numpy: random plane counts, value distributions
# (dense / sparse / runs => BIT, GAP, FULL, NULL, absent planes), sizes that end inside a block, NULL elements
for seed in range(120):
    rng = np.random.default_rng(97000 + seed)
    nplanes = int(rng.integers(1, 33))
    n = int(rng.integers(1, 5 * 65536))
    kind = seed % 4
    hi = 1 << nplanes
    if kind == 0: col = rng.integers(0, hi, size=n)
    elif kind == 1: col = np.where(rng.random(n) < 0.02, rng.integers(0, hi, size=n), 0)
    elif kind == 2: col = np.repeat(rng.integers(0, hi, size=n // 700 + 1), 700)[:n]
    else: col = rng.integers(0, min(hi, 50), size=n)
    col = col.astype(np.uint64)
    notnull = rng.random(n) < 0.95
    col[~notnull] = 0
    def up(bits):
        w = np.packbits(np.concatenate([bits.astype(np.uint8), np.zeros((-n) % 32, np.uint8)]), bitorder="little").view(np.uint32)
        return bm.bit_import_u32(ctx, w, True)
    sl = []
    for b in range(nplanes):
        bits = ((col >> np.uint64(b)) & np.uint64(1)).astype(bool)
        sl.append(up(bits) if bits.any() else None)
    nn = up(notnull)
    size = n if seed % 3 else max(1, n - int(rng.integers(0, min(n, 70000))))
    with_null = bool(seed % 2)
    sc = bm.slice_scanner(ctx, sl, size=size, not_null=nn if with_null else None)
    c = col[:size]; valid = (notnull if with_null else np.ones(n, bool))[:size]
    pool = [int(x) for x in rng.choice(c, min(30, c.size))] + [0, 1, hi - 1, hi, (1 << 33) + 1] + [int(x) for x in rng.integers(0, hi, size=20)]
    exp = [int(((c == np.uint64(v)) & (valid if v == 0 else True)).sum()) for v in pool]
    got = sc.find_eq_counts(pool, method="transpose").tolist()
    if got != exp: bad += 1; print("FAIL eq_counts", seed, nplanes, n, size, kind, with_null)
# counts pipelines over GAP-only operands: counting formulation (gap_count 1) vs the run-by-run kernel vs the oracle; and the
# comparison searches in partial-block passes vs the whole-block kernel vs numpy
ctx.set_tuning("pipe_split", 0)
for seed in range(80):
    rng = np.random.default_rng(99000 + seed)
    nblk = int(rng.integers(1, 5)); nbits = nblk * 65536 - int(rng.integers(0, 3000))
    nvec = int(rng.integers(2, 300)); nsub = int(rng.integers(0, 12))
    dq = int(rng.choice([5, 40, 150, 300, 65520, 65000]))
    common = port.gen_words(7000 + seed, 0xFFFFFFFF, max(dq // 3, 2) if dq < 1000 else 65400, nbits)
    ws = []
    for v in range(nvec + nsub):
        w = port.gen_words(7000 + seed, v, dq if dq < 1000 or v % 2 else 65530, nbits)
        if v < nvec: w = w | common
        if rng.integers(0, 15) == 0: w[:2048] = 0xFFFFFFFF if v < nvec else 0
        ws.append(w)
    gvs = [bm.bit_import_u32(ctx, w, True) for w in ws]
    if any(v.calc_stat()["bit_blocks"] for v in gvs): continue
    pvs = [port.import_words(w, True, nbits) for w in ws]
    groups = [(list(range(nvec)), list(range(nvec, nvec + nsub))), (list(range(0, nvec, 3)), [])]
    exp = port.pipeline_counts([([pvs[i] for i in a], [pvs[i] for i in s_]) for a, s_ in groups])
    pipe = bm.aggregator.pipeline(ctx)
    for a, s_ in groups:
        ag = pipe.add()
        for i in a: ag.add(gvs[i], 0)
        for i in s_: ag.add(gvs[i], 1)
    pipe.complete()
    for gc in (1, 0):
        ctx.set_tuning("gap_count", gc)
        got = agg.combine_and_sub(pipe)
        if not (got == exp).all(): bad += 1; print("FAIL gapcount", seed, gc, nvec, nsub, dq, got, exp)
ctx.set_tuning("gap_count", -1); ctx.set_tuning("pipe_split", -1)
for seed in range(60):
    rng = np.random.default_rng(99500 + seed)
    nplanes = int(rng.integers(1, 20)); n = int(rng.integers(1, 4 * 65536)); hi = 1 << nplanes
    col = (rng.integers(0, hi, size=n) if seed % 2 else np.repeat(rng.integers(0, hi, size=n // 300 + 1), 300)[:n]).astype(np.uint64)
    def up(bits):
        w = np.packbits(np.concatenate([bits.astype(np.uint8), np.zeros((-n) % 32, np.uint8)]), bitorder="little").view(np.uint32)
        return bm.bit_import_u32(ctx, w, True)
    sl = []
    for b in range(nplanes):
        bits = ((col >> np.uint64(b)) & np.uint64(1)).astype(bool)
        sl.append(up(bits) if bits.any() else None)
    sc = bm.slice_scanner(ctx, sl, size=n)
    for _ in range(6):
        v0, v1 = sorted(int(x) for x in rng.integers(0, hi + 3, size=2))
        exp = [int((col > np.uint64(v0)).sum()), int((col <= np.uint64(v0)).sum()), int(((col >= np.uint64(v0)) & (col <= np.uint64(v1))).sum())]
        for halves in (1, 0):
            ctx.set_tuning("range_halves", halves)
            got = [sc.count(bm.CMP_GT, v0), sc.find_le(v0).count(), sc.count(bm.CMP_RANGE, v0, v1)]
            if got != exp: bad += 1; print("FAIL compare", seed, halves, v0, v1, got, exp)
ctx.set_tuning("range_halves", 1)
print("soak done, failures:", bad)
PY
timeout 1200 python /tmp/soak.py > gpurun_out/soak.log 2>&1; tail -6 gpurun_out/soak.log
