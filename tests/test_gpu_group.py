"""GPU: the multi-device group layer of the C-ABI (include/bmx.h "device groups") on ONE MI355X -- a group may
list a device several times, so the 1-GPU box runs 3- and 8-member groups (one stream per member).  Everything a
group computes must equal what a single context computes and what the oracle says: block columns are independent
(src/bmaggregator.h:1184-1218), so block-range shards add up / concatenate exactly."""
import numpy as np
import pytest

import bitmagic_amd as bm

pytestmark = pytest.mark.gpu

SEED = 0xB17A61C


def _vectors(port, nvec, nbits, dens, seed0=0, common=False):
    words = [port.gen_words(SEED + 17, seed0 + v, dens[v % len(dens)], nbits, with_common=common) for v in range(nvec)]
    return words, [port.import_words(w, True, nbits) for w in words]


@pytest.mark.parametrize("members", [1, 3, 8])
def test_group_matches_single_context_and_oracle(ctx, port, members):
    nbits = 37 * 65536 + 4321                     # 38 blocks: uneven shards for 3 and 8 members
    words, pv = _vectors(port, 9, nbits, [6554, 300, 30000, 2, 655])     # bit, GAP, dense, nearly empty, mixed
    grp = bm.group([0] * members)
    assert grp.size() == members
    r = [grp.shard_range(38, m) for m in range(members)]
    assert r[0][0] == 0 and r[-1][1] == 38 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
    sv = [bm.bvector.from_block_table(ctx, nbits, *p.flatten()) for p in pv]
    for g, s, p in zip(gv, sv, pv):
        assert g.count() == s.count() == p.count()
        assert g.info()["counts"] == s.info()["counts"]
        k, o, b, gp = g.block_table()
        ek, eo, eb, egp = p.flatten()
        assert k.tolist() == ek.tolist()
        # content: install the gathered table into a single-context vector and compare words
        back = bm.bvector.from_block_table(ctx, nbits, k, o, b, gp)
        assert (back.to_words() == p.to_words()).all()
    for op in range(4):
        for i, j in ((0, 1), (1, 2), (4, 0), (3, 3 - 1)):
            assert bm.gbvector.count_op2(op, gv[i], gv[j]) == port.count_op2(op, pv[i], pv[j])
            for oc in (bm.opt_none, bm.opt_compress):
                t = bm.gbvector._op2(op, gv[i], gv[j], oc)
                e = port.op2(op, pv[i], pv[j], oc == bm.opt_compress)
                assert t.count() == e.count()
                assert t.block_table()[0].tolist() == e.flatten()[0].tolist(), (op, i, j, oc)
    agg = bm.gaggregator(grp)
    t, any_ = agg.combine_and_sub([gv[0], gv[2]], [gv[1], gv[4]])
    e = port.agg_and_sub([pv[0], pv[2]], [pv[1], pv[4]])
    assert t.count() == e.count() and any_ == (e.count() > 0)
    assert t.block_table()[0].tolist() == e.flatten()[0].tolist()
    assert agg.find_first_and_sub([gv[0], gv[2]], [gv[1], gv[4]]) == port.find_first_and_sub([pv[0], pv[2]], [pv[1], pv[4]])
    for a_, s_ in (([gv[1], gv[4]], []), ([gv[3]], [gv[0]]), ([gv[3], gv[1]], [])):
        pa = [pv[gv.index(x)] for x in a_]; ps = [pv[gv.index(x)] for x in s_]
        ef = port.find_first_and_sub(pa, ps)
        gf = agg.find_first_and_sub(a_, s_)
        assert gf[0] == ef[0] and (not ef[0] or gf[1] == ef[1])
    for oc in (False, True):
        agg.set_optimization(oc)
        o = agg.combine_or(gv)
        eo = port.agg_or(pv, oc)
        assert o.count() == eo.count() and o.block_table()[0].tolist() == eo.flatten()[0].tolist()
    groups = [(list(range(3)), []), ([0, 2], [1, 4]), ([0], []), ([2, 0], [3])]
    pipe = bm.gaggregator.pipeline(grp)
    for a, s in groups:
        ag = pipe.add()
        for i in a: ag.add(gv[i], 0)
        for i in s: ag.add(gv[i], 1)
    pipe.complete()
    got = agg.combine_and_sub(pipe)
    exp = port.pipeline_counts([([pv[i] for i in a], [pv[i] for i in s]) for a, s in groups])
    assert (got == exp).all(), (got, exp)
    ms = pipe.last_ms()
    assert len(ms) == members and all(x >= 0 for x in ms)
    # rank / select: per-shard index + the ones before each shard; queries routed to the owner
    rng = np.random.default_rng(members)
    for g, s, p in zip(gv, sv, pv):
        grs, rs, prs = g.build_rs_index(), s.build_rs_index(), port.rs_build(p)
        c = p.count()
        assert grs.count() == rs.count() == c
        borders = np.array([b * 65536 + d for b in (0, 12, 13, 25, 26, 37) for d in (-1, 0, 1, 65535) if b * 65536 + d >= 0], np.uint64)
        q = np.concatenate([rng.integers(0, nbits + 70000, size=3000).astype(np.uint64), borders])   # incl. past the end
        assert (g.rank(q, grs) == s.rank(q, rs)).all()
        assert (g.rank(q, grs) == prs.rank(q)).all()
        r = np.concatenate([rng.integers(0, c + 3, size=3000).astype(np.uint64), np.array([0, 1, max(c, 1), c + 1], np.uint64)])
        gf, gp_ = g.select(r, grs)
        sf, sp = s.select(r, rs)
        ppos, pf = prs.select(r)
        assert (gf == sf).all() and (gp_[gf] == sp[sf]).all()
        assert (gf == pf).all() and (gp_[gf] == ppos[pf]).all()
        if c:
            assert g.select(1, grs) == s.select(1, rs) and g.rank(nbits - 1, grs) == c
    grp.close()


def test_group_with_more_members_than_blocks(ctx, port):
    """8 members, 3 blocks: five shards are empty -- counts, pairwise ops, aggregation, rank / select still add up"""
    nbits = 3 * 65536 - 11
    words, pv = _vectors(port, 4, nbits, [6554, 300, 30000])
    grp = bm.group([0] * 8)
    gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
    for g, p in zip(gv, pv):
        assert g.count() == p.count()
    assert bm.gbvector.count_op2(bm.AND, gv[0], gv[2]) == port.count_op2(bm.AND, pv[0], pv[2])
    t = bm.gbvector.bit_or(gv[0], gv[1])
    assert t.count() == port.op2(bm.OR, pv[0], pv[1]).count()
    agg = bm.gaggregator(grp)
    r, any_ = agg.combine_and_sub([gv[0], gv[2]], [gv[1]])
    assert r.count() == port.agg_and_sub([pv[0], pv[2]], [pv[1]]).count()
    grs, prs = gv[2].build_rs_index(), port.rs_build(pv[2])
    q = np.arange(0, nbits + 70000, 997, dtype=np.uint64)
    assert (gv[2].rank(q, grs) == prs.rank(q)).all()
    c = pv[2].count()
    r = np.arange(0, c + 2, max(1, c // 500), dtype=np.uint64)
    gf, gp = gv[2].select(r, grs)
    ppos, pf = prs.select(r)
    assert (gf == pf).all() and (gp[gf] == ppos[pf]).all()
    grp.close()


@pytest.mark.parametrize("members", [1, 3, 8])
def test_group_slice_scanner(ctx, port, members):
    """scanner call pattern over SHARDED bit-planes (bmx_gslice_compare + group aggregator / pipeline): range and
    equality searches equal the single-device scanner and numpy, with NULL elements, an absent plane, and rows that
    end inside the last block"""
    rng = np.random.default_rng(members)
    n = 9 * 65536 + 777
    col = np.where(rng.random(n) < 0.6, rng.integers(0, 3000, size=n), 0).astype(np.uint64)
    col[2 * 65536:3 * 65536] = 1023                                     # FULL blocks in the low planes
    col &= ~np.uint64(1 << 7)                                           # plane 7 does not exist
    notnull = rng.random(n) < 0.85
    col[~notnull] = 0
    nplanes = 12
    def words(bits):
        return np.packbits(np.concatenate([bits.astype(np.uint8), np.zeros((-n) % 32, np.uint8)]), bitorder="little").view(np.uint32)
    grp = bm.group([0] * members)
    gs, ss = [], []
    for b in range(nplanes):
        bits = ((col >> np.uint64(b)) & np.uint64(1)).astype(bool)
        if not bits.any(): gs.append(None); ss.append(None); continue
        p = port.import_words(words(bits), True, n)
        gs.append(bm.gbvector.from_block_table(grp, n, *p.flatten()))
        ss.append(bm.bvector.from_block_table(ctx, n, *p.flatten()))
    assert gs[7] is None
    pn = port.import_words(words(notnull), True, n)
    gnn = bm.gbvector.from_block_table(grp, n, *pn.flatten()); snn = bm.bvector.from_block_table(ctx, n, *pn.flatten())
    def bits_of(gv):
        k, o, b, g = gv.block_table()
        back = bm.bvector.from_block_table(ctx, n, k, o, b, g)
        return np.unpackbits(back.to_words((n + 31) // 32).view(np.uint8), bitorder="little")[:n].astype(bool)
    for with_null in (False, True):
        gsc = bm.gslice_scanner(grp, gs, size=n, not_null=gnn if with_null else None)
        ssc = bm.slice_scanner(ctx, ss, size=n, not_null=snn if with_null else None)
        valid = notnull if with_null else np.ones(n, bool)
        for v in (0, 1, 100, 1023, 1024, 2999, 5000):
            V = np.uint64(v)
            for pred, exp in ((bm.CMP_GT, col > V), (bm.CMP_GE, (col >= V) & (valid if v == 0 else True)),
                              (bm.CMP_LT, (col < V) & valid), (bm.CMP_LE, (col <= V) & valid)):
                assert gsc.count(pred, v) == ssc.count(pred, v) == int(exp.sum()), (members, with_null, pred, v)
            assert (bits_of(gsc.find_le(v)) == ((col <= V) & valid)).all()
            t, f = gsc.find_eq(v)
            e = (col == V) & (valid if v == 0 else True)
            assert f == bool(e.any()) and (t is None or (bits_of(t) == e).all()), (members, with_null, v)
        assert gsc.count(bm.CMP_RANGE, 10, 2000) == int(((col >= 10) & (col <= 2000)).sum())
        assert (bits_of(gsc.find_range(0, 50)) == ((col <= 50) & valid)).all()
        assert (bits_of(gsc.find_zero()) == ((col == 0) & valid)).all() and (bits_of(gsc.find_nonzero()) == (col != 0)).all()
        vals = [int(x) for x in rng.choice(col[col > 0], 20)] + [0, 4095, 1 << 20]
        assert (gsc.find_eq_counts(vals) == ssc.find_eq_counts(vals, method="pipeline")).all()
        assert (gsc.find_eq_counts(vals, method="pipeline") == ssc.find_eq_counts(vals)).all()
        assert (gsc.find_eq_counts(vals) == np.array([int(((col == np.uint64(x)) & (valid if x == 0 else True)).sum()) for x in vals], np.uint64)).all()
        ff = gsc.find_first_eq(vals[0])
        assert ff == ssc.find_first_eq(vals[0]) == (True, int(np.flatnonzero(col == np.uint64(vals[0]))[0]))
    grp.close()


def test_group_generate_equals_single_generate(ctx, port):
    """bmx_gvec_generate: every member generates its own block range of the SAME logical vector"""
    nbits = 100 * 65536 + 99
    grp = bm.group([0, 0, 0, 0])
    for vid, dq, common in ((1, 6554, True), (2, 200, False)):
        g = bm.gbvector.generate(grp, SEED, vid, dq, nbits, with_common=common)
        s = bm.bvector.generate(ctx, SEED, vid, dq, nbits, with_common=common)
        assert g.count() == s.count()
        assert g.block_table()[0].tolist() == s.block_table()[0].tolist()
    grp.close()


def test_group_rccl_single_member(ctx, port):
    """BMX_GROUP_RCCL on the one device of this box: librccl is loaded on demand, ncclCommInitAll(1), the counts
    go through ncclAllReduce (a 1-rank all-reduce) -- the code path the 8-GPU job takes"""
    nbits = 20 * 65536
    words, pv = _vectors(port, 4, nbits, [6554, 20000])
    try:
        grp = bm.group([0], bm.GROUP_RCCL)
    except bm.BmxError as e:
        pytest.fail(f"RCCL group could not be created: {e}")
    gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
    agg = bm.gaggregator(grp)
    pipe = bm.gaggregator.pipeline(grp)
    for a, s in (([0, 1, 2], []), ([0], [3])):
        ag = pipe.add()
        for i in a: ag.add(gv[i], 0)
        for i in s: ag.add(gv[i], 1)
    pipe.complete()
    got = agg.combine_and_sub(pipe)
    exp = port.pipeline_counts([([pv[0], pv[1], pv[2]], []), ([pv[0]], [pv[3]])])
    assert (got == exp).all()
    with pytest.raises(bm.BmxError):
        bm.group([0, 0], bm.GROUP_RCCL)          # RCCL needs distinct devices
    grp.close()


def test_group_rccl_over_all_visible_devices(port):
    """the first multi-GPU lease must not also be the first test.  When >= 2 devices are visible (skipped on the
    one-GPU boxes of this pool) a BMX_GROUP_RCCL group over min(device_count, 8) DISTINCT devices must come up with that many
    RCCL ranks, its pipeline counts (summed by the in-library ncclAllReduce) and its materialised results must equal the
    single-device / oracle ones, shards must sit on their own devices, and the exchange time is reported per member."""
    n = min(bm.device_count(), 8)
    if n < 2:
        pytest.skip(f"{bm.device_count()} device visible: the multi-device RCCL path needs >= 2")
    nbits = 97 * 65536 + 4321
    words, pv = _vectors(port, 6, nbits, [6554, 20000, 655])
    grp = bm.group(list(range(n)), bm.GROUP_RCCL)
    try:
        assert grp.rccl_ranks() == n, (grp.rccl_ranks(), n)
        gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
        agg = bm.gaggregator(grp)
        groups = [([0, 1, 2], []), ([0], [3]), ([4, 5], [1]), ([0, 1, 2, 3, 4, 5], [])]
        pipe = bm.gaggregator.pipeline(grp)
        for a, s in groups:
            ag = pipe.add()
            for i in a: ag.add(gv[i], 0)
            for i in s: ag.add(gv[i], 1)
        pipe.complete()
        exp = port.pipeline_counts([([pv[i] for i in a], [pv[i] for i in s]) for a, s in groups])
        for _ in range(3):
            got = agg.combine_and_sub(pipe)
            assert (got == exp).all(), (got, exp)
        xs = pipe.last_exchange_ms(); ks = pipe.last_ms()
        assert len(xs) == n and len(ks) == n and all(x >= 0 for x in xs)
        # the same counts from ONE device (a plain context on device 0) and through the host-sum exchange
        c0 = bm.context(0)
        sv = [bm.bvector.from_block_table(c0, nbits, *p.flatten()) for p in pv]
        sp = bm.aggregator.pipeline(c0)
        for a, s in groups:
            ag = sp.add()
            for i in a: ag.add(sv[i], 0)
            for i in s: ag.add(sv[i], 1)
        sp.complete()
        assert (bm.aggregator(c0).combine_and_sub(sp) == got).all()
        del sp, sv; c0.close()
        # materialised results gathered from the shards; pairwise counts over shards
        t, _any = agg.combine_and_sub([gv[0], gv[1]], [gv[3]])
        e = port.agg_and_sub([pv[0], pv[1]], [pv[3]])
        ge = bm.gbvector.from_block_table(grp, nbits, *e.flatten())
        assert t.count() == e.count() and bm.gbvector.count_op2(bm.XOR, t, ge) == 0
        assert bm.gbvector.count_op2(bm.AND, gv[0], gv[4]) == port.count_op2(0, pv[0], pv[4])
    finally:
        grp.close()


def test_group_full_size_headline_shards(ctx):
    """BASELINE configs[2] shape through the group API at reduced width: 32 x 1e9-bit vectors, 8 members on one
    GPU; the sharded count equals the single-context count (size-independent property: shard sums = total)"""
    nbits, nvec = 1_000_000_000, 32
    grp = bm.group([0] * 8)
    gv = [bm.gbvector.generate(grp, SEED, v, 6554, nbits, with_common=True) for v in range(nvec)]
    agg = bm.gaggregator(grp)
    pipe = bm.gaggregator.pipeline(grp)
    ag = pipe.add()
    for v in gv: ag.add(v, 0)
    pipe.complete()
    got = int(agg.combine_and_sub(pipe)[0])
    del pipe, gv
    grp.close()
    sv = [bm.bvector.generate(ctx, SEED, v, 6554, nbits, with_common=True) for v in range(nvec)]
    a = bm.aggregator(ctx); p = bm.aggregator.pipeline(ctx); g = p.add()
    for v in sv: g.add(v, 0)
    p.complete()
    exp = int(a.combine_and_sub(p)[0])
    assert got == exp and got > 90_000_000


def test_byte_weighted_shard_borders(ctx, port):
    """shards weighted by non-NULL operand bytes.  A collection whose first half is NULL (empty top-level
    ranges, src/bmblocks.h:556-564): equal block counts would leave half of the members idle; the weighted cut gives
    every member the same share of operand bytes, results stay identical to the single-context ones."""
    nblk, members = 64, 4
    nbits = nblk * 65536
    tabs, pv = [], []
    for v in range(6):
        w = port.gen_words(SEED + 5, v, [6554, 300, 30000][v % 3], nbits)
        w[: (nblk // 2) * 2048] = 0                      # first half empty -> NULL blocks
        p = port.import_words(w, True, nbits)
        pv.append(p); tabs.append(p.flatten())
    grp = bm.group([0] * members)
    # default cut: equal block counts
    assert [grp.shard_range(nblk, m) for m in range(members)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    bounds = grp.partition_for_tables(tabs)
    assert bounds[0] == 0 and bounds[-1] == nblk and all(bounds[i] <= bounds[i + 1] for i in range(members))
    assert bounds[1] >= nblk // 2                         # the empty half costs nothing: it all lands on member 0
    assert [grp.shard_range(nblk, m) for m in range(members)] == [(int(bounds[m]), int(bounds[m + 1])) for m in range(members)]
    # per-member operand bytes within 10 % of each other (what decides the busy time of an HBM-bound pass)
    wsum = np.zeros(nblk, np.uint64)
    for k, o, b, g in tabs:
        for nb in range(nblk):
            wsum[nb] += 8192 if k[nb] == 2 else (2 * ((int(g[o[nb]]) >> 3) + 1) if k[nb] == 3 else 0)
    per = [int(wsum[bounds[m]:bounds[m + 1]].sum()) for m in range(members)]
    assert max(per) <= 1.10 * (sum(per) / members), per
    gv = [bm.gbvector.from_block_table(grp, nbits, *t) for t in tabs]
    # borders cannot change under live vectors
    with pytest.raises(bm.BmxError):
        grp.set_partition(nblk, [0, 16, 32, 48, 64])
    for g, p in zip(gv, pv):
        assert g.count() == p.count()
        assert g.block_table()[0].tolist() == p.flatten()[0].tolist()
    agg = bm.gaggregator(grp)
    t, _ = agg.combine_and_sub([gv[0], gv[3]], [gv[1]])
    e = port.agg_and_sub([pv[0], pv[3]], [pv[1]])
    assert t.count() == e.count() and t.block_table()[0].tolist() == e.flatten()[0].tolist()
    o = agg.combine_or(gv)
    assert o.count() == port.agg_or(pv).count()
    pipe = bm.gpipeline(grp)
    for a_, s_ in (([0, 3], []), ([2, 5], [1]), ([4], [0, 2])):
        ag = pipe.add()
        for i in a_: ag.add(gv[i], 0)
        for i in s_: ag.add(gv[i], 1)
    pipe.complete()
    got = agg.combine_and_sub(pipe)
    exp = port.pipeline_counts([([pv[i] for i in a_], [pv[i] for i in s_]) for a_, s_ in (([0, 3], []), ([2, 5], [1]), ([4], [0, 2]))])
    assert got.tolist() == [int(x) for x in exp]
    # member busy times: the pipeline's per-member operand bytes follow the cut
    pb = pipe.operand_bytes()
    assert sum(pb) > 0 and len(pipe.last_ms()) == members and len(pipe.last_exchange_ms()) == members
    rs = gv[0].build_rs_index()
    q = np.array([0, nbits // 2 - 1, nbits // 2, nbits // 2 + 70000, nbits - 1], np.uint64)
    assert (gv[0].count_to(q, rs) == port.rs_build(pv[0]).rank(q)).all()
    f, pos = gv[0].select(np.array([1, 5, pv[0].count()], np.uint64), rs)
    epos, ef = port.rs_build(pv[0]).select(np.array([1, 5, pv[0].count()], np.uint64))
    assert f.all() and (pos == epos).all()
    del pipe, rs, gv, t, o, g, ag
    import gc; gc.collect()
    grp.set_partition(nblk, [0, 16, 32, 48, 64])          # no live vectors: allowed again
    grp.close()


@pytest.mark.parametrize("members", [1, 3])
def test_group_prepared_collections_and_search_limit(port, members):
    """round 4 over shards: bmx_gcollection_prepare (every member transposes its block range; aggregations and pipelines of the
    group then use the collections member by member) and bmx_gpipeline_set_search_count_limit (every member under the same
    limit: the sum is >= min(limit, true) and <= true): results = the oracle's, = the group without collections / without a limit"""
    import test_gpu_parity as P
    rng = np.random.default_rng(31 + members)
    nblk = 23
    nbits = nblk * 65536 - 777
    words = P._sparse_collection(port, rng, 90, nbits, 40, long_runs=True, ragged=False, specials=True)
    pv = [port.import_words(w, True, nbits) for w in words]
    assert all(p.flatten()[0].tolist().count(2) == 0 for p in pv)
    grp = bm.group([0] * members)
    try:
        gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
        agg = bm.gaggregator(grp)
        grp.collection_prepare(gv, bm.ROLE_OR); grp.collection_prepare(gv, bm.ROLE_AND)
        for m in range(members):
            assert grp.member_pack_stats(m)["collections"] in (0, 2)           # (0: a member whose shard holds no GAP block)
        assert sum(grp.member_pack_stats(m)["collections"] for m in range(members)) >= 2
        nw = nblk * 2048
        for sel in (list(range(90)), rng.choice(90, size=40, replace=False).tolist(), list(range(89, 20, -1))):
            o = agg.combine_or([gv[i] for i in sel]); e = port.agg_or([pv[i] for i in sel], False)
            assert o.count() == e.count() and o.block_table()[0].tolist() == e.flatten()[0].tolist()
            t, any_ = agg.combine_and_sub([gv[i] for i in sel[:8]], [gv[i] for i in sel[8:]])
            e = port.agg_and_sub([pv[i] for i in sel[:8]], [pv[i] for i in sel[8:]])
            assert t.count() == e.count() and any_ == (e.count() > 0) and t.block_table()[0].tolist() == e.flatten()[0].tolist()
        groups = [(rng.choice(90, size=int(rng.integers(1, 40)), replace=False).tolist(), rng.choice(90, size=int(rng.integers(0, 50)), replace=False).tolist()) for _ in range(12)]
        groups = [(a, [i for i in s_ if i not in a]) for a, s_ in groups]
        def run(limit):
            pipe = bm.gaggregator.pipeline(grp)
            for a, s_ in groups:
                ag = pipe.add()
                for i in a: ag.add(gv[i], 0)
                for i in s_: ag.add(gv[i], 1)
            if limit is not None: pipe.set_search_count_limit(limit)
            pipe.complete()
            return [int(x) for x in agg.combine_and_sub(pipe)]
        true = [int(x) for x in port.pipeline_counts([([pv[i] for i in a], [pv[i] for i in s_]) for a, s_ in groups])]
        assert run(None) == true
        for limit in (1, 25, max(true) + 1):
            got = run(limit)
            assert all(min(limit, t) <= x <= t for x, t in zip(got, true)), (limit, got, true)
    finally:
        grp.close()


def test_group_workers_persist_across_calls(port):
    """the materialising group calls run on persistent per-member workers: many calls, same results, errors carried over"""
    nbits = 10 * 65536
    pv = [port.import_words(port.gen_words(SEED + 9, v, 6554, nbits), True, nbits) for v in range(3)]
    grp = bm.group([0] * 4)
    gv = [bm.gbvector.from_block_table(grp, nbits, *p.flatten()) for p in pv]
    exp = [port.op2(op, pv[0], pv[1], False).count() for op in range(4)]
    for _ in range(50):
        for op in range(4):
            assert bm.gbvector._op2(op, gv[0], gv[1]).count() == exp[op]
    other = bm.group([0] * 2)
    ov = bm.gbvector.from_block_table(other, nbits, *pv[2].flatten())
    with pytest.raises(bm.BmxError):
        bm.gbvector._op2(0, gv[0], ov)
    with pytest.raises(bm.BmxError):                     # a malformed table is refused by a worker, the text reaches the caller
        k, o, b, g = pv[0].flatten()
        bad = k.copy(); bad[7] = 9
        bm.gbvector.from_block_table(grp, nbits, bad, o, b, g)
    assert bm.gbvector._op2(0, gv[0], gv[1]).count() == exp[0]
    grp.close(); other.close()


def test_bench_plain_gpus2_uses_two_members():
    """`python bench.py --gpus 2` with no RANK in the environment must itself run 2 GPUs (here: the one-device test hook,
    two members of a bmx_group on device 0) and say so; never an n_gpus: 1 line for --gpus 2"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["BMX_BENCH_TEST_ONE_DEVICE"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"): env.pop(k, None)
    nbits = 40 * 65536 + 99
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--nvec", "16", "--nbits", str(nbits),
                          "--steps", "3", "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["mode"] == "group" and len(j["per_rank"]["kernel_ms"]) == 2
    import oracle
    P = oracle.port()
    vecs = [P.import_words(P.gen_words(SEED, v, 6554, nbits, with_common=True), True, nbits) for v in range(16)]
    assert j["config"]["result_count"] == int(P.pipeline_counts([(vecs, [])])[0])
    assert j["weak_scaling"]["result_count"] > 0
    # without the hook a 1-GPU box must refuse --gpus 2 instead of printing a 1-GPU line
    env.pop("BMX_BENCH_TEST_ONE_DEVICE")
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--nvec", "4", "--nbits", str(nbits),
                              "--steps", "1", "--warmup", "0", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode != 0 and not any(l.startswith("{") for l in out.stdout.splitlines())


def test_bench_gpus8_through_the_one_device_hook():
    """the 8-GPU invocations the driver will make, on ONE device (test hook: 8 members of a bmx_group on device 0): the headline
    `bench.py --gpus 8` and `bench.py --config 4 --gpus 8` must run, say n_gpus 8, print exchange / rccl_ranks / rccl_error
    whatever happened, balance the members and return the single-context results"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["BMX_BENCH_TEST_ONE_DEVICE"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"): env.pop(k, None)
    nbits = 203 * 65536 + 99
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--nvec", "16", "--nbits", str(nbits),
                          "--steps", "3", "--warmup", "1", "--no-cpu", "--no-weak"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 8 and j["mode"] == "group" and len(j["per_rank"]["kernel_ms"]) == 8
    for k in ("exchange", "rccl_ranks", "rccl_error", "rccl_requested"): assert k in j
    assert j["exchange"] == "host_sum" and j["rccl_ranks"] == 0                 # (one device: RCCL cannot come up; the hook says so)
    rng_ = j["config"]["member_block_ranges"]
    assert rng_[0][0] == 0 and rng_[-1][1] == 204 and all(a[1] == b[0] for a, b in zip(rng_, rng_[1:]))
    assert max(b - a for a, b in rng_) - min(b - a for a, b in rng_) <= 1
    import oracle
    P = oracle.port()
    vecs = [P.import_words(P.gen_words(SEED, v, 6554, nbits, with_common=True), True, nbits) for v in range(16)]
    assert j["config"]["result_count"] == int(P.pipeline_counts([(vecs, [])])[0])
    # configs[4] shape over 8 members
    nb4 = 333 * 65536 - 5
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "4", "--gpus", "8", "--or-vecs", "96", "--nbits", str(nb4),
                          "--steps", "2", "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j4 = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j4["n_gpus"] == 8 and j4["mode"] == "group"
    for k in ("exchange", "rccl_ranks", "rccl_error"): assert k in j4
    mb = j4["config"]["member_gap_bytes"]
    assert len(mb) == 8 and min(mb) > 0 and max(mb) <= 1.25 * min(mb), mb                 # the members hold about the same bytes
    ctx1 = bm.context(0)
    vv = [bm.bvector.generate(ctx1, SEED, 10000 + i, 13, nb4) for i in range(96)]
    assert j4["config"]["result_count"] == bm.aggregator(ctx1).combine_or(vv).count()
    del vv; ctx1.close()
