"""Round-6 randomized differential soak (GPU): what this round added, against the oracle on random inputs --
(H) select lines (k_select_sel / k_rs_sel_build / k_rs_sel_check): random vectors of every block kind (NULL, FULL, dense, sparse,
    runs, antisparse, edge bits), both offset widths and the policy form; every one of the vector in order + random batches + dead
    queries against the oracle's select; rank(select(r)) == r; the same answers with the lines switched off;
(I) pipeline::set_search_count_limit through the ASYNCHRONOUS counts entry (k_limit_null): random pipelines over bit-block, mixed
    and GAP-only operands, with / without a packed collection behind them -- the device totals equal the synchronous run's and lie
    in [min(limit, true), true].
Usage: python tools/soak_r06.py [rounds] [parts]   (one FAIL line per difference, then "soak_r06 done, failures: N")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import numpy as np
import oracle, bitmagic_amd as bm
import test_gpu_stress as S
import test_gpu_parity as P

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ONLY = sys.argv[2] if len(sys.argv) > 2 else "HI"
port = oracle.port()
bad = 0
ran = {"H": 0, "I": 0}


def fail(*a):
    global bad
    bad += 1
    print("FAIL", *a, flush=True)


# ---------------------------------------------------------------- (H) select lines
for seed in range(ROUNDS if "H" in ONLY else 0):
    rng = np.random.default_rng(610000 + seed)
    nblk = int(rng.integers(1, 40))
    mode = int(rng.choice([-1, 1, 2]))
    c = bm.context(0)
    c.set_tuning("rs_select_sel", mode)
    if rng.integers(0, 2): c.set_tuning("rs_lines", int(rng.choice([0, 2])))
    style = int(rng.integers(0, 3))
    if style == 0:                                     # every block kind mixed
        words = np.concatenate([S._random_block(rng, S.KINDS[int(rng.integers(0, len(S.KINDS)))]) for _ in range(nblk)])
    elif style == 1:                                   # even density (keeps the 16-bit form when it is dense enough)
        words = port.gen_words(int(rng.integers(1, 1 << 30)), seed, int(rng.choice([200, 655, 3000, 6554, 30000])), nblk * 65536)
    else:                                              # a few ones per block: lines span blocks (32-bit form)
        words = np.zeros(nblk * 2048, np.uint32)
        for p in rng.integers(0, nblk * 65536, size=int(rng.integers(1, 40 * nblk))): words[p >> 5] |= np.uint32(1 << (int(p) & 31))
    nbits = nblk * 65536 - int(rng.integers(0, 5000)) * int(rng.integers(0, 2))
    last = nbits - (nblk - 1) * 65536
    if last < 65536:
        tail = np.unpackbits(words[(nblk - 1) * 2048:].view(np.uint8), bitorder="little"); tail[last:] = 0
        words[(nblk - 1) * 2048:] = np.packbits(tail, bitorder="little").view(np.uint32)
    pv = port.import_words(words, bool(rng.integers(0, 2)), nbits)
    cnt = pv.count()
    if rng.integers(0, 2): gv = bm.bvector.from_block_table(c, nbits, *pv.flatten())
    else: gv = bm.bit_import_u32(c, words[: (nbits + 31) // 32], True)
    rs, prs = gv.build_rs_index(), port.rs_build(pv)
    info = rs.info()
    ran["H"] += 1
    if rs.count() != cnt: fail("H count", seed, rs.count(), cnt)
    if mode > 0 and cnt and info["select_offset_bits"] not in (16, 32): fail("H lines not built", seed, mode, info)
    if mode == 2 and cnt and info["select_offset_bits"] != 32: fail("H 32-bit form not taken", seed, info)
    if info["select_offset_bits"] and info["select_lines_bytes"] != (cnt + (59 if info["select_offset_bits"] == 16 else 29)) // (60 if info["select_offset_bits"] == 16 else 30) * 128:
        fail("H bytes", seed, info, cnt)
    if cnt:
        allr = np.arange(1, cnt + 1, dtype=np.uint64)
        if cnt > 300000: allr = allr[:: cnt // 300000 + 1]
        f, pos = gv.select(allr, rs)
        ppos, pf = prs.select(allr)
        if not (f.all() and (pos == ppos).all()): fail("H every one", seed, mode, info, int((pos != ppos).sum()))
        if not (np.asarray(gv.rank(pos, rs)) == allr).all(): fail("H rank(select)", seed)
    r = np.concatenate([rng.integers(1, max(cnt, 1) + 1, size=int(rng.integers(1, 5000))).astype(np.uint64), np.array([0, cnt, cnt + 1, 2 ** 40, 1], np.uint64)])
    f, pos = gv.select(r, rs)
    ppos, pf = prs.select(r)
    if not ((f == pf).all() and (pos[f] == ppos[pf]).all() and (pos[~f] == 0).all()): fail("H random batch", seed, mode, info)
    c.set_tuning("rs_select_sel", 0)                   # the same index without its lines
    f0, p0 = gv.select(r, rs)
    if not ((f0 == pf).all() and (p0[f0] == ppos[pf]).all()): fail("H lines off", seed)
    del rs, gv
    c.close()

# ---------------------------------------------------------------- (I) the search limit on the asynchronous counts entry
import torch
for seed in range(ROUNDS if "I" in ONLY else 0):
    rng = np.random.default_rng(620000 + seed)
    c = bm.context(0)
    nblk = int(rng.integers(20, 400))
    nbits = nblk * 65536
    flavour = int(rng.integers(0, 3))                  # 0 bit-blocks, 1 mixed, 2 GAP-only
    dqs = {0: [6554, 20000], 1: [655, 6554, 66], 2: [66, 120, 13]}[flavour]
    nvec = int(rng.integers(4, 20))
    vs = [bm.bvector.generate(c, 9000 + seed, i, int(rng.choice(dqs)), nbits, with_common=True) for i in range(nvec)]
    if flavour == 2 and rng.integers(0, 2):
        c.collection_prepare(vs, bm.ROLE_AND)
        if rng.integers(0, 2): c.set_tuning("coll_members", 1)
    if rng.integers(0, 3) == 0: c.set_tuning("pipe_staged", int(rng.choice([0, 1])))
    ng = int(rng.choice([1, 3, 40, 200]))
    def mk(limit):
        pipe = bm.aggregator.pipeline(c)
        g2 = np.random.default_rng(seed)
        for g in range(ng):
            ag = pipe.add()
            na = int(g2.integers(1, min(nvec, 6) + 1)); a = g2.choice(nvec, size=na, replace=False).tolist()
            for i in a: ag.add(vs[i], 0)
            if g2.integers(0, 3) == 0:
                rest = [i for i in range(nvec) if i not in a]
                if rest: ag.add(vs[int(g2.choice(rest))], 1)
        if limit: pipe.set_search_count_limit(limit)
        pipe.complete()
        return pipe
    agg = bm.aggregator(c)
    full = [int(x) for x in agg.combine_and_sub(mk(None))]
    limit = max(1, int(np.median(full)) // int(rng.choice([2, 8, 64])) + 1)
    p = mk(limit)
    sync = [int(x) for x in agg.combine_and_sub(p)]
    d = torch.full((ng,), -1, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    agg.run_counts_dev(p, d.data_ptr()); c.synchronize()
    dev = [int(x) for x in d.cpu().tolist()]
    ran["I"] += 1
    if dev != sync: fail("I dev != sync", seed, flavour, ng, [(k, a, b) for k, (a, b) in enumerate(zip(dev, sync)) if a != b][:4], p.describe())
    if not all(min(limit, t) <= g <= t for g, t in zip(dev, full)): fail("I bounds", seed, flavour, limit, dev[:6], full[:6])
    pn = mk(None)
    d.fill_(-1); torch.cuda.synchronize()
    agg.run_counts_dev(pn, d.data_ptr()); c.synchronize()
    if [int(x) for x in d.cpu().tolist()] != full: fail("I no limit", seed)
    del p, pn, vs
    c.close()

print("cases run:", ran)
print("soak_r06 done, failures:", bad)
