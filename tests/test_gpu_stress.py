"""GPU randomized differential stress test -- the analogue of StressTest / FillSetsRandomMethod
(tests/stress/t.cpp:917,11439): random block tables (every block independently NULL / FULL / bit / GAP with
random run structure, un-optimised representations included), random operations, compared with the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

import bitmagic_amd as bm  # noqa: E402


def _random_block(rng, kind):
    """-> 2048 uint32 words of one block of the requested flavour"""
    if kind == "zero":
        return np.zeros(2048, np.uint32)
    if kind == "ones":
        return np.full(2048, 0xFFFFFFFF, np.uint32)
    if kind == "dense":
        return rng.integers(0, 1 << 32, size=2048, dtype=np.uint64).astype(np.uint32)
    if kind == "sparse":
        w = np.zeros(2048, np.uint32)
        for p in rng.integers(0, 65536, size=int(rng.integers(1, 200))):
            w[p >> 5] |= np.uint32(1 << (int(p) & 31))
        return w
    if kind == "runs":                       # a few long runs: GAP with big interiors
        bits = np.zeros(65536, np.uint8)
        pos = np.sort(rng.integers(0, 65536, size=2 * int(rng.integers(1, 40))))
        for a, b in zip(pos[::2], pos[1::2]):
            bits[a:b + 1] = 1
        return np.packbits(bits, bitorder="little").view(np.uint32).copy()
    if kind == "antisparse":                 # almost all ones
        w = np.full(2048, 0xFFFFFFFF, np.uint32)
        for p in rng.integers(0, 65536, size=int(rng.integers(1, 100))):
            w[p >> 5] &= np.uint32(~(1 << (int(p) & 31)) & 0xFFFFFFFF)
        return w
    if kind == "edge":                       # bits at block / word / wave borders
        w = np.zeros(2048, np.uint32)
        for p in (0, 31, 32, 1023, 1024, 21824, 21825, 43648, 65535):
            if rng.integers(0, 2): w[p >> 5] |= np.uint32(1 << (p & 31))
        return w
    raise ValueError(kind)


KINDS = ["zero", "ones", "dense", "sparse", "runs", "antisparse", "edge"]


def _random_vector(rng, port, ctx, nblocks, optimize):
    words = np.concatenate([_random_block(rng, KINDS[int(rng.integers(0, len(KINDS)))]) for _ in range(nblocks)])
    pv = port.import_words(words, optimize, nblocks * 65536)
    if rng.integers(0, 2):                  # half of the vectors go through the host block-table upload
        k, o, b, g = pv.flatten()
        gv = bm.bvector.from_block_table(ctx, nblocks * 65536, k, o, b, g)
    else:
        gv = bm.bit_import_u32(ctx, words, optimize)
    return pv, gv


@pytest.mark.parametrize("seed", range(6))
def test_random_block_tables(ctx, port, seed, agg_path):
    rng = np.random.default_rng(1000 + seed)
    nvec = int(rng.integers(3, 14))
    vecs = [_random_vector(rng, port, ctx, int(rng.integers(1, 9)), bool(rng.integers(0, 2))) for _ in range(nvec)]
    pv = [v[0] for v in vecs]; gv = [v[1] for v in vecs]
    nwb = 9 * 2048
    agg = bm.aggregator(ctx)
    for v, g in zip(pv, gv):
        assert g.count() == v.count()
        assert (g.to_words(nwb) == v.to_words(nwb)).all()
    for _ in range(12):
        i, j = (int(x) for x in rng.integers(0, nvec, 2))
        op = int(rng.integers(0, 4)); opt = bm.opt_compress if rng.integers(0, 2) else bm.opt_none
        t = bm.bvector._op2(op, gv[i], gv[j], opt)
        e = port.op2(op, pv[i], pv[j], opt == bm.opt_compress)
        assert (t.to_words(nwb) == e.to_words(nwb)).all(), (seed, op, i, j)
        assert bm._count_op2(op, gv[i], gv[j]) == port.count_op2(op, pv[i], pv[j]) == e.count()
    groups = []
    for _ in range(10):
        na = int(rng.integers(1, nvec + 1)); ns = int(rng.integers(0, nvec))
        a = [int(x) for x in rng.integers(0, nvec, na)]; s = [int(x) for x in rng.integers(0, nvec, ns)]
        groups.append((a, s))
        t, any_ = agg.combine_and_sub([gv[k] for k in a], [gv[k] for k in s])
        e = port.agg_and_sub([pv[k] for k in a], [pv[k] for k in s])
        assert (t.to_words(nwb) == e.to_words(nwb)).all(), (seed, a, s)
        assert any_ == (e.count() > 0)
        f, idx = agg.find_first_and_sub([gv[k] for k in a], [gv[k] for k in s])
        pf, pidx = port.find_first_and_sub([pv[k] for k in a], [pv[k] for k in s])
        assert f == pf and (not f or idx == pidx), (seed, a, s, f, idx, pf, pidx)
        o = agg.combine_or([gv[k] for k in a + s])
        assert (o.to_words(nwb) == port.agg_or([pv[k] for k in a + s]).to_words(nwb)).all()
    exp = port.pipeline_counts([([pv[k] for k in a], [pv[k] for k in s]) for a, s in groups])
    for staged in (0, 1):
        ctx.set_tuning("pipe_staged", staged)
        try:
            pipe = bm.aggregator.pipeline(ctx)
            for a, s in groups:
                ag = pipe.add()
                for k in a: ag.add(gv[k], 0)
                for k in s: ag.add(gv[k], 1)
            pipe.complete()
            assert (agg.combine_and_sub(pipe) == exp).all(), (seed, staged)
        finally:
            ctx.set_tuning("pipe_staged", -1)
    # rank / select on a random member
    k = int(rng.integers(0, nvec))
    rs, prs = gv[k].build_rs_index(), port.rs_build(pv[k])
    nbits = pv[k].nbits
    q = rng.integers(0, nbits, size=400).astype(np.uint64)
    assert (gv[k].rank(q, rs) == prs.rank(q)).all()
    c = pv[k].count()
    r = rng.integers(0, c + 2, size=400).astype(np.uint64)
    found, pos = gv[k].select(r, rs)
    ppos, pfound = prs.select(r)
    assert (found == pfound).all() and (pos[found] == ppos[pfound]).all()
    bc, sub = rs.export(); pbc, psub = prs.export()
    assert (bc == pbc).all() and (sub == psub).all()


def test_error_behaviour(ctx):
    """status codes follow libbm's (lang-maps/libbm/include/libbm.h:28-35): BADARG = 2, RANGE = 3"""
    one = bm.bit_import_u32(ctx, np.array([5], np.uint32))
    with pytest.raises(bm.BmxError) as e:
        ctx.set_tuning("no_such_knob", 1)
    assert e.value.status == 2
    # malformed block tables are rejected, not trusted
    with pytest.raises(bm.BmxError) as e:
        bm.bvector.from_block_table(ctx, 65536, [bm.BIT], [3], np.zeros(2048, np.uint32), np.zeros(0, np.uint16))
    assert e.value.status == 3
    with pytest.raises(bm.BmxError) as e:                      # GAP block whose last run end is not 65535
        bm.bvector.from_block_table(ctx, 65536, [bm.GAP], [0], np.zeros(0, np.uint32), np.array([2 << 3, 100, 200], np.uint16))
    assert e.value.status == 3
    with pytest.raises(bm.BmxError) as e:
        bm.bvector.from_block_table(ctx, 65536, [7], [0], np.zeros(0, np.uint32), np.zeros(0, np.uint16))
    assert e.value.status == 2
    # run ends must be strictly ascending (checked on the device while the block is re-packed): an interior 65535, a
    # repeated end and a descending pair are all refused; so is a length beyond the top GAP level (1279 runs)
    good = np.array([(4 << 3), 10, 20, 30, 65535, 0, 0, 0], np.uint16)
    assert bm.bvector.from_block_table(ctx, 65536, [bm.GAP], [0], np.zeros(0, np.uint32), good).count() == 10 + 65505
    for bad in ([(4 << 3), 10, 65535, 30, 65535], [(4 << 3), 10, 10, 30, 65535], [(4 << 3), 30, 20, 40, 65535]):
        with pytest.raises(bm.BmxError) as e:
            bm.bvector.from_block_table(ctx, 65536, [bm.GAP], [0], np.zeros(0, np.uint32), np.array(bad + [0] * 3, np.uint16))
        assert e.value.status == 3, bad
    long_ = np.concatenate([[1280 << 3], np.arange(0, 2 * 1279, 2), [65535]]).astype(np.uint16)
    with pytest.raises(bm.BmxError) as e:
        bm.bvector.from_block_table(ctx, 65536, [bm.GAP], [0], np.zeros(0, np.uint32), long_)
    assert e.value.status == 3
    # two GAP blocks + a bit-block in one table: a bad SECOND block is found too, and nothing leaks (mem_used returns)
    used = ctx.mem_used()
    two = np.concatenate([good, np.array([(2 << 3), 500, 400, 0, 0, 0, 0, 0], np.uint16)])
    with pytest.raises(bm.BmxError):
        bm.bvector.from_block_table(ctx, 3 * 65536, [bm.GAP, bm.BIT, bm.GAP], [0, 0, 8], np.ones(2048, np.uint32), two)
    assert ctx.mem_used() == used
    pipe = bm.aggregator.pipeline(ctx)
    with pytest.raises(RuntimeError):
        bm.aggregator(ctx).combine_and_sub(pipe)               # not complete()
    ag = pipe.add(); ag.add(one, 0); pipe.complete()
    with pytest.raises(RuntimeError):
        pipe.add()                                             # no add() after complete()
    with pytest.raises(bm.BmxError) as e:                      # block range upside down
        bm.aggregator(ctx)._run_pipeline(pipe, 5, 2)
    assert e.value.status == 3
    # the allocator cache can be returned to the driver at any time
    before = ctx.mem_used(); ctx.trim(); assert ctx.mem_used() == before
    assert one.count() == 2


def test_degenerate_gap_blocks_and_unoptimized_tables(ctx, port):
    """vectors as an un-optimised reference container can hold them: GAP blocks that are
    all-zero / all-one (len 1), bit-blocks that are all-zero / all-one; treated by content, not by kind"""
    G0 = np.array([(1 << 3) | 0, 65535], np.uint16)              # all-zero GAP block
    G1 = np.array([(1 << 3) | 1, 65535], np.uint16)              # all-one GAP block
    G2 = np.array([(3 << 3) | 0, 99, 199, 65535], np.uint16)     # bits 100..199
    G3 = np.array([(2 << 3) | 1, 0, 65535], np.uint16)           # only bit 0
    gaps = np.concatenate([G0, G1, G2, G3])
    kinds = [bm.GAP, bm.GAP, bm.GAP, bm.GAP, bm.BIT, bm.BIT, bm.NULL, bm.FULL]
    offs = [0, 2, 4, 8, 0, 1, 0, 0]
    bits = np.concatenate([np.zeros(2048, np.uint32), np.full(2048, 0xFFFFFFFF, np.uint32)])
    nbits = 8 * 65536
    a = bm.bvector.from_block_table(ctx, nbits, kinds, offs, bits, gaps)
    pa = port.from_table(nbits, kinds, offs, bits, gaps)
    assert a.count() == pa.count() == 65536 + 100 + 1 + 65536 + 65536
    assert (a.to_words() == pa.to_words()).all()
    rng = np.random.default_rng(9)
    w = rng.integers(0, 1 << 32, size=8 * 2048, dtype=np.uint64).astype(np.uint32)
    b = bm.bit_import_u32(ctx, w, True); pb = port.import_words(w, True)
    agg = bm.aggregator(ctx)
    for op in range(4):
        for x, y, px, py in ((a, b, pa, pb), (b, a, pb, pa), (a, a, pa, pa)):
            for opt in (bm.opt_none, bm.opt_compress):
                t = bm.bvector._op2(op, x, y, opt)
                assert (t.to_words() == port.op2(op, px, py, opt == bm.opt_compress).to_words()).all(), (op, opt)
            assert bm._count_op2(op, x, y) == port.count_op2(op, px, py)
    t, _ = agg.combine_and_sub([a, b], [])
    assert (t.to_words() == port.agg_and_sub([pa, pb], []).to_words()).all()
    t, _ = agg.combine_and_sub([b], [a])
    assert (t.to_words() == port.agg_and_sub([pb], [pa]).to_words()).all()
    assert (agg.combine_or([a, b]).to_words() == port.agg_or([pa, pb]).to_words()).all()
    rs, prs = a.build_rs_index(), port.rs_build(pa)
    q = np.arange(0, nbits, 1237, dtype=np.uint64)
    assert (a.rank(q, rs) == prs.rank(q)).all()
    r = np.arange(1, pa.count() + 1, 997, dtype=np.uint64)
    f, p = a.select(r, rs); pp, pf = prs.select(r)
    assert f.all() and (p == pp).all()


def test_two_contexts_from_two_threads(port):
    """distinct contexts (one HIP stream each) may be driven concurrently from different host threads"""
    import threading
    nbits = 40 * 65536
    words = [port.gen_words(5150, v, 6554, nbits, with_common=True) for v in range(12)]
    exp = int(port.pipeline_counts([([port.import_words(w, True, nbits) for w in words], [])])[0])
    errors = []

    def work(tid):
        try:
            c = bm.context(0)
            vecs = [bm.bit_import_u32(c, w, True) for w in words]
            agg = bm.aggregator(c)
            for _ in range(20):
                pipe = bm.aggregator.pipeline(c)
                ag = pipe.add()
                for v in vecs: ag.add(v, 0)
                pipe.complete()
                got = int(agg.combine_and_sub(pipe)[0])
                t, _ = agg.combine_and_sub(vecs, [])
                if got != exp or t.count() != exp:
                    errors.append((tid, got, exp))
            del vecs
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append((tid, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors


@pytest.mark.parametrize("seed", range(40))
def test_block_kinds_follow_reference(ctx, port, seed):
    """REPRESENTATION parity on random block tables (tools/gpu_runs/soak.sh runs 400 of these): block kinds of the
    pairwise ops in both opt modes (copied bit / GAP blocks, GAP x GAP, computed blocks), of combine_and_sub, of
    combine_or with and without set_optimization, of pipeline results and the OR target, of shift-right-and.
    No whitelist: an all-ones GAP x GAP result is a 1-run GAP block here as in the reference (clone_gap_block)."""
    rng = np.random.default_rng(70000 + seed)
    nblk = int(rng.integers(1, 7)); nv = int(rng.integers(2, 7))
    vecs = [_random_vector(rng, port, ctx, nblk, bool(rng.integers(0, 2))) for _ in range(nv)]
    pv = [v[0] for v in vecs]; gv = [v[1] for v in vecs]
    for p, g in zip(pv, gv):
        assert g.block_table()[0].tolist() == p.flatten()[0].tolist()
    agg = bm.aggregator(ctx)
    for op in range(4):
        i, j = (int(x) for x in rng.integers(0, nv, 2))
        for oc in (True, False):
            kk = bm.bvector._op2(op, gv[i], gv[j], bm.opt_compress if oc else bm.opt_none).block_table()[0].tolist()
            ek = port.op2(op, pv[i], pv[j], oc).flatten()[0].tolist()
            assert kk == ek, (op, oc, i, j, kk, ek)
    na = int(rng.integers(1, nv + 1))
    t, _ = agg.combine_and_sub(gv[:na], gv[na:])
    assert t.block_table()[0].tolist() == port.agg_and_sub(pv[:na], pv[na:]).flatten()[0].tolist()
    for oc in (True, False):
        agg.set_optimization(oc)
        assert agg.combine_or(gv).block_table()[0].tolist() == port.agg_or(pv, oc).flatten()[0].tolist(), oc
        sel = [int(x) for x in rng.integers(0, nv, int(rng.integers(1, 12)))]
        t, f = agg.combine_shift_right_and([gv[k] for k in sel])
        e, ef = port.agg_shift_right_and([pv[k] for k in sel], oc, False)
        assert f == ef and t.block_table()[0].tolist()[:nblk] == e.flatten()[0].tolist()[:nblk]
    agg.set_optimization(False)


def test_result_memory_is_what_the_result_holds(ctx, port):
    """a materialised result keeps only what it holds: the full-size slab is transient.  AND of two vectors that meet in
    3 of 400 blocks: the result owns 3 bit-blocks (compacted), its download moves 3 blocks, content = oracle; a result
    that keeps 395 of 400 blocks stays in its slab (no second pass) and its download gathers the 395 live ones"""
    nblk = 400
    rng = np.random.default_rng(77)
    wa = np.zeros(nblk * 2048, np.uint32); wb = np.zeros(nblk * 2048, np.uint32)
    wa[:] = rng.integers(0, 1 << 32, wa.size, dtype=np.uint64).astype(np.uint32)
    for nb in (5, 200, 399):
        wb[nb * 2048:(nb + 1) * 2048] = rng.integers(0, 1 << 32, 2048, dtype=np.uint64).astype(np.uint32)
    a, b = bm.bit_import_u32(ctx, wa, True), bm.bit_import_u32(ctx, wb, True)
    pa, pb = port.import_words(wa, True, wa.size * 32), port.import_words(wb, True, wb.size * 32)
    used = ctx.mem_used()
    t = bm.bvector.bit_and(a, b)
    i = t.info()
    assert i["counts"][bm.BIT] == 3 and i["bit_slab_blocks"] == 3
    assert ctx.mem_used() - used < 3 * 8192 + 2 * (2 << 20)            # not 400 x 8 KiB: three blocks + table (pool granules)
    k, o, bits, gaps = t.block_table()
    assert bits.size == 3 * 2048 and sorted(o[k == bm.BIT].tolist()) == [0, 1, 2]
    e = port.op2(0, pa, pb, False)
    assert (t.to_words() == e.to_words()).all() and k.tolist() == e.flatten()[0].tolist()
    back = bm.bvector.from_block_table(ctx, nblk * 65536, k, o, bits, gaps)
    assert (back.to_words() == e.to_words()).all()
    # nearly full result: 395 of 400 blocks survive (b2 = all ones except five empty blocks)
    wb2 = np.full(nblk * 2048, 0xFFFFFFFF, np.uint32)
    for nb in (0, 7, 8, 123, 398):
        wb2[nb * 2048:(nb + 1) * 2048] = 0
    b2 = bm.bit_import_u32(ctx, wb2, True); pb2 = port.import_words(wb2, True, wb2.size * 32)
    t2 = bm.bvector.bit_and(a, b2)
    i2 = t2.info()
    assert i2["counts"][bm.BIT] == 395 and i2["bit_slab_blocks"] == 395
    k2, o2, bits2, gaps2 = t2.block_table()
    assert bits2.size == 395 * 2048 and sorted(o2[k2 == bm.BIT].tolist()) == list(range(395))
    e2 = port.op2(0, pa, pb2, False)
    back2 = bm.bvector.from_block_table(ctx, nblk * 65536, k2, o2, bits2, gaps2)
    assert (back2.to_words() == e2.to_words()).all() and (t2.to_words() == e2.to_words()).all()
    # a clone of a slab with unused slots keeps the ordinals; pipeline results over many groups stay small
    t3 = bm.bvector.bit_and(t2, t2)                                      # aliasing: block-for-block copy (src/bm.h:6191)
    assert (t3.to_words() == e2.to_words()).all() and t3.block_table()[1][k2 == bm.BIT].tolist() == o2[k2 == bm.BIT].tolist()
    pipe = bm.aggregator.pipeline(ctx, bm.agg_run_options(True, True))
    for _ in range(24):
        ag = pipe.add(); ag.add(a, 0); ag.add(b, 0)
    pipe.complete()
    used = ctx.mem_used()
    agg = bm.aggregator(ctx); agg.combine_and_sub(pipe)
    assert all(r.info()["bit_slab_blocks"] <= 3 for r in pipe.get_bv_res_vector())
    assert ctx.mem_used() - used < 24 * (3 * 8192 + (1 << 20)) + (8 << 20)   # 24 x 3 blocks, not 24 x 400


def test_randomized_soak_slices():
    """the randomized differential soaks (tools/soak_r04.py parts A-E: row kernel, collection members, long
    mixed pairwise operations, search limits, asynchronous chains; tools/soak_r05.py parts F-G: the AND rows kernel, per-group
    search limits; round 6: tools/soak_r06.py parts H-I: select lines, the search limit on the asynchronous counts entry) ran only under the builder's gpurun.  A bounded slice of each, fixed seeds, runs here: about a minute."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, rounds in (("soak_r04.py", "4"), ("soak_r05.py", "14"), ("soak_r06.py", "8")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", script), rounds], capture_output=True, text=True, timeout=900, cwd=root)
        tail = (r.stdout + r.stderr)[-2000:]
        assert r.returncode == 0 and "done, failures: 0" in r.stdout, (script, tail)


_REDZONE_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bitmagic_amd as bm
import __graft_entry__ as g
g.smoke()                                            # pipeline counts, combine_and_sub, count_and, rs index + rank on a red-zone context
ctx = bm.context(0)
out = {"enabled": ctx.redzone_check()["enabled"]}
nbits = 40 * 65536 - 77
vs = [bm.bvector.generate(ctx, 99, i, dq, nbits) for i, dq in enumerate((6554, 655, 66, 30000, 200, 66, 66, 200))]
agg = bm.aggregator(ctx)
for op in ("bit_and", "bit_or", "bit_xor", "bit_sub"):
    r = getattr(bm.bvector, op)(vs[0], vs[1]); r.count(); r2 = getattr(bm.bvector, op)(vs[2], vs[4], bm.opt_compress); r2.count()
agg.combine_or(vs[2:]); agg.combine_and_sub(vs[4:6], [vs[6]])
ctx.collection_prepare(vs[4:], 1); agg.combine_or(vs[4:])
for v in (vs[0], vs[1], vs[2]):
    rs = v.build_rs_index(); n = rs.count()
    if n: v.select(np.arange(1, min(n, 5000) + 1, dtype=np.uint64), rs); v.rank(np.arange(0, nbits, 997, dtype=np.uint64), rs)
ctx.synchronize()
out["hits_after_workload"] = ctx.redzone_check()["hits"]
ctx.inject_failure(5, 0)                             # one byte written right behind a 1000-byte allocation
rep = ctx.redzone_check()
out["hits_after_self_test"] = rep["hits"]; out["report"] = rep["report"]
ctx.inject_failure(5, 7)                             # ... and behind a 1007-byte one: the zone starts at the next multiple of 16
out["hits_after_second"] = ctx.redzone_check()["hits"]
try:
    ctx.synchronize(); out["sync_after"] = "ok"
except bm.BmxError as e:
    out["sync_after"] = "error %d" % e.status
print("REDZONE " + json.dumps(out))
'''


def test_red_zone_allocator_catches_overruns_and_the_suite_is_clean():
    """a context created under BMX_DEBUG_REDZONE=1 surrounds every device allocation with canaries (in front, and
    from the end of the REQUESTED bytes to the end of the block) and verifies them at free / synchronize / destroy.  Here: the
    smoke workload + pairwise ops, aggregations, a collection, rs indexes (rank lines, select lines) run clean; a deliberate
    one-byte overrun IS caught and names the allocation; a slice of both soaks runs clean under the checker.  (The whole -m gpu
    suite + both soaks under the checker: profiles/r06_redzone/.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMX_DEBUG_REDZONE="1")
    r = subprocess.run([sys.executable, "-c", _REDZONE_SCRIPT], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("REDZONE ")]
    assert r.returncode == 0 and line, (r.stdout + r.stderr)[-3000:]
    out = json.loads(line[0][8:])
    assert out["enabled"] and out["hits_after_workload"] == 0, out
    assert out["hits_after_self_test"] == 1 and out["hits_after_second"] == 2, out
    assert "BEHIND" in out["report"] and "1000 bytes" in out["report"] and "bmx.hip:" in out["report"], out
    assert out["sync_after"] == "ok"                   # (a damage is reported once: the zones were repainted)
    assert r.stderr.count("[bmx redzone] allocation") == 2, r.stderr[-2000:]
    for script, rounds in (("soak_r04.py", "2"), ("soak_r05.py", "6")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", script), rounds], capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert r.returncode == 0 and "done, failures: 0" in r.stdout and "[bmx redzone]" not in r.stderr, (script, (r.stdout + r.stderr)[-2000:])


def test_device_allocation_failures_come_back_as_status(port):
    """fault injection (bmx_debug_inject_failure kind 4): the k-th device allocation of a call fails.  Whatever k, the call
    returns BMX_ERR_BADALLOC (or succeeds, if it allocates fewer than k + 1 times) -- never a crash, never a wrong answer -- the
    context keeps working, and what the failed call had allocated is given back."""
    c = bm.context(0)
    nbits = 30 * 65536 - 5
    vs = [bm.bvector.generate(c, 5, i, dq, nbits) for i, dq in enumerate((6554, 655, 66, 66, 200, 66))]
    pv = [port.import_words(port.gen_words(5, i, dq, nbits), True, nbits) for i, dq in enumerate((6554, 655, 66, 66, 200, 66))]
    agg = bm.aggregator(c)
    exp_and = port.count_op2(0, pv[0], pv[1])
    exp_or = port.agg_or(pv[2:]).count()
    def pairwise(): return bm.bvector.bit_and(vs[0], vs[1]).count() == exp_and
    def combine_or(): return agg.combine_or(vs[2:]).count() == exp_or
    def rs_index():
        rs = vs[0].build_rs_index()
        return rs.count() == pv[0].count() and (vs[0].select(np.array([1, 77], np.uint64), rs)[1] == port.rs_build(pv[0]).select(np.array([1, 77], np.uint64))[0]).all()
    def prepare(): c.collection_prepare(vs[2:], 1); return agg.combine_or(vs[2:]).count() == exp_or
    def pipeline():
        pipe = bm.aggregator.pipeline(c)
        ag = pipe.add(); ag.add(vs[0], 0); ag.add(vs[1], 0); ag.add(vs[4], 1)
        pipe.complete()
        return int(agg.combine_and_sub(pipe)[0]) == int(port.pipeline_counts([([pv[0], pv[1]], [pv[4]])])[0])
    for name, fn in (("pairwise", pairwise), ("combine_or", combine_or), ("rs_index", rs_index), ("prepare", prepare), ("pipeline", pipeline)):
        assert fn(), name                                 # warm: pooled blocks, scratch sized
        c.synchronize(); c.trim()
        base = c.mem_used()
        failed = 0
        for k in range(0, 40):
            c.inject_failure(4, k)
            try:
                ok = fn()
                assert ok, (name, k)
            except bm.BmxError as e:
                assert e.status == 1, (name, k, str(e))
                failed += 1
            finally:
                c.inject_failure(0, 0)
            c.synchronize()
        assert fn(), name
        c.synchronize()
        assert failed >= 1, name                          # (every one of these calls allocates at least once)
        leaked = c.mem_used() - base
        assert leaked <= (2 << 20), (name, leaked)        # (pool rounding of a fresh result may differ; a lost slab would be MBs per failure)
    c.close()
