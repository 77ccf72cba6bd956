#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE ITSELF (BitMagic 9.2.1,
/root/reference/src compiled by oracle/Makefile into oracle/_ref/).  Runs only in
the build container; the outputs are committed so every other machine can check
the oracle (and through it the HIP path) without the reference present.

    python tests/golden/make_golden.py            # writes golden_avx2.json (+ checks scalar agrees)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import oracle  # noqa: E402
from cases import (AGG_GROUPS, BM64_NVEC, CASES, HINT_GROUPS, OR_SETS, PAIRS, SCANNER_EQ_BATCH, SCANNER_IN_LISTS, SCANNER_RANGES, SCANNER_ROWS,  # noqa: E402
                   SCANNER_S_RANGES, SCANNER_S_VALUES, SCANNER_VALUES, SEARCH_LIMITS, SEED, SHIFT_SETS, bm64_build, bm64_queries, make_inputs, range_hints,
                   rank_queries, scanner_values, scanner_values_signed, select_queries, sha)


def gap_slab_masked(kinds, offs, gaps):
    g = gaps.copy()
    for k, o in zip(kinds, offs):
        if k == oracle.GAP:
            g[o] &= 0xFFF9          # capacity level bits are an allocator detail
    return g


def run(R, P):
    out = {"reference": R.name, "simd_version": int(R.lib.ref_simd_version()), "seed": SEED, "cases": {}}
    # generator known answers (pins bmo_gen_word64; the HIP generator is checked against the same)
    out["generator_kat"] = [
        {"vec": v, "w64": w, "dq": d, "word": int(P.lib.bmo_gen_word64(SEED, v, w, d))}
        for v, w, d in [(0, 0, 6554), (1, 12345, 655), (0xFFFFFFFF, 7, 32768), (255, 15625000 - 1, 6554), (3, 1 << 33, 1)]]
    for case in CASES:
        words, nbits = make_inputs(P, case)
        vecs = [R.import_words(w, True, nbits) for w in words]
        c = {"nbits": nbits, "input_sha": [sha(w) for w in words]}
        flat = [v.flatten() for v in vecs]
        c["kinds"] = [f[0].tolist() for f in flat]
        c["gap_sha"] = [sha(gap_slab_masked(f[0], f[1], f[3])) for f in flat]
        c["gap_words"] = [int(f[3].size) for f in flat]
        c["count"] = [v.count() for v in vecs]
        # pairwise
        c["op2"] = {}
        for (i, j) in PAIRS:
            for op in range(4):
                t = R.op2(op, vecs[i], vecs[j], False)
                tc = R.op2(op, vecs[i], vecs[j], True)
                assert (t.to_words() == tc.to_words()).all()
                c["op2"][f"{op}:{i}:{j}"] = {"sha": sha(t.to_words()), "count": t.count(),
                                             "count_op": R.count_op2(op, vecs[i], vecs[j]),
                                             "kinds_opt": tc.flatten()[0].tolist()}
        # aggregator
        c["agg_and_sub"] = []
        for (a, s) in AGG_GROUPS:
            t = R.agg_and_sub([vecs[i] for i in a], [vecs[i] for i in s])
            c["agg_and_sub"].append({"and": a, "sub": s, "sha": sha(t.to_words()), "count": t.count(),
                                     "kinds": t.flatten()[0].tolist()})
        # find_first_and_sub: the logical first bit of the reference's own combine_and_sub result; the
        # reference call is recorded too (it may narrow the search range by the SUB group, :1526 TODO)
        c["find_first"] = []
        for (a, s_) in AGG_GROUPS:
            t = R.agg_and_sub([vecs[i] for i in a], [vecs[i] for i in s_])
            lf = R.find_first(t)
            rf = R.find_first_and_sub([vecs[i] for i in a], [vecs[i] for i in s_])
            c["find_first"].append({"and": a, "sub": s_, "found": bool(lf[0]), "idx": lf[1] if lf[0] else 0,
                                    "reference_call_agrees": bool(rf == lf or (not rf[0] and not lf[0]))})
        c["agg_or"] = []
        for o in OR_SETS:
            t = R.agg_or([vecs[i] for i in o])
            c["agg_or"].append({"src": o, "sha": sha(t.to_words()), "count": t.count(), "kinds": t.flatten()[0].tolist(),
                                "kinds_opt": R.agg_or([vecs[i] for i in o], True).flatten()[0].tolist()})
        # combine_shift_right_and (bmaggregator.h:2494): default opt_none target, opt_compress kinds,
        # the `any` form (first result block only) and the set_compute_count form
        c["shift_right_and"] = []
        for o in SHIFT_SETS:
            src = [vecs[i] for i in o]
            t, f = R.agg_shift_right_and(src, False, False)
            tc, fc = R.agg_shift_right_and(src, True, False)
            ta, fa = R.agg_shift_right_and(src, False, True)
            assert (t.to_words() == tc.to_words()).all() and f == fc == fa
            c["shift_right_and"].append({"src": o, "sha": sha(t.to_words()), "count": t.count(), "found": bool(f),
                                         "kinds": t.flatten()[0].tolist(), "kinds_opt": tc.flatten()[0].tolist(),
                                         "any_sha": sha(ta.to_words()), "any_count": ta.count(),
                                         "count_mode": R.agg_shift_right_and_count(src)})
        cnt = R.pipeline_counts([([vecs[i] for i in a], [vecs[i] for i in s]) for (a, s) in AGG_GROUPS])
        c["pipeline_counts"] = [int(x) for x in cnt]
        assert c["pipeline_counts"] == [g["count"] for g in c["agg_and_sub"]]
        # pipeline::set_search_count_limit (bmaggregator.h:255, honoured per block at :1361-1367): the reference's counts under a
        # limit ("can find more, cannot find less"), and the oracle's restatement of the rule checked against them on the spot
        groups_rp = lambda O, vv: [([vv[i] for i in a], [vv[i] for i in s]) for (a, s) in AGG_GROUPS]
        c["search_limit"] = {}
        pvecs = [P.import_words(w, True, nbits) for w in words]
        for lim in SEARCH_LIMITS:
            rc_ = [int(x) for x in R.pipeline_counts_limit(groups_rp(R, vecs), lim)]
            assert all(min(lim, t) <= x <= t for x, t in zip(rc_, c["pipeline_counts"])), (case, lim)
            assert rc_ == [int(x) for x in P.pipeline_counts_limit(groups_rp(P, pvecs), lim)], (case, lim)
            c["search_limit"][str(lim)] = rc_
        # full pipeline (agg_opt_bvect_and_counts + OR target), AggregatorTest t.cpp:10378-10580 shape
        res, rc, ort = R.pipeline_results([([vecs[i] for i in a], [vecs[i] for i in s_]) for (a, s_) in AGG_GROUPS])
        assert [int(x) for x in rc] == c["pipeline_counts"]
        c["pipeline_results"] = {
            "present": [r is not None for r in res],
            "sha": [sha(r.to_words()) if r is not None else None for r in res],
            "kinds": [r.flatten()[0].tolist() if r is not None else None for r in res],
            "or_target_sha": sha(ort.to_words()), "or_target_kinds": ort.flatten()[0].tolist(), "or_target_count": ort.count()}
        # rank / select
        c["rs"] = []
        for vi in (0, 1, 2):
            v = vecs[vi]
            rs = R.rs_build(v)
            bc, sub = rs.export(v.nblocks)
            rq = rank_queries(nbits)
            sq = select_queries(v.count())
            pos, found = rs.select(sq)
            rq_l = rq[::7][:60]; rq_r = rq[3::7][:60]
            fr_rank = sq[:40]; fr_from = rq[:40]
            fr = [rs.find_rank(int(a_), int(b_)) for a_, b_ in zip(fr_rank, fr_from)]
            c["rs"].append({"vec": vi, "count": rs.count(), "bcount": bc.tolist(), "sub_count": [int(x) for x in sub],
                            "count_range": [rs.count_range(int(a_), int(b_)) for a_, b_ in zip(rq_l, rq_r)],
                            "rank_corrected": [rs.rank_corrected(int(a_)) for a_ in rq[:120]],
                            "count_to_test": [rs.count_to_test(int(a_)) for a_ in rq[:120]],
                            "find_rank_found": [int(f) for f, _ in fr], "find_rank_pos": [int(p_) if f else 0 for f, p_ in fr],
                            "rank": [int(x) for x in rs.rank(rq)],
                            "select_found": found.astype(int).tolist(),
                            "select_pos": [int(p) if f else 0 for p, f in zip(pos, found)]})
        # set_range_hint (bmaggregator.h:481,974): find_first_and_sub under a hint, and a pipeline whose options enable
        # search masks (agg_run_options<true, true, true>, :65,78,1312-1346) -- results + counts per group
        c["range_hint"] = []
        for (frm, to) in range_hints(nbits):
            res, cnt = R.pipeline_masks([([vecs[i] for i in a], [vecs[i] for i in s_]) for (a, s_) in HINT_GROUPS], frm, to)
            # find_first_and_sub under the hint: the logical first bit of the reference's own hinted result; the reference
            # call itself is recorded too -- for a column whose AND operands are all FULL it reads a temp block it never
            # filled (is_res_full, bmaggregator.h:1483-1497,1536-1547), like the unhinted call noted above
            ff = []
            for gi, (a, s_) in enumerate(HINT_GROUPS):
                ok, f, idx = R.find_first_and_sub_range([vecs[i] for i in a], [vecs[i] for i in s_], frm, to)
                lf = R.find_first(res[gi]) if res[gi] is not None else (False, 0)
                ff.append({"hint_one_block": ok, "found": bool(lf[0]), "idx": lf[1] if lf[0] else 0,
                           "reference_call_agrees": bool((f, idx if f else 0) == (bool(lf[0]), lf[1] if lf[0] else 0))})
            c["range_hint"].append({"from": frm, "to": to, "find_first": ff, "counts": [int(x) for x in cnt],
                                    "present": [r is not None for r in res],
                                    "sha": [sha(r.to_words()) if r is not None else None for r in res],
                                    "kinds": [r.flatten()[0].tolist() if r is not None else None for r in res]})
        out["cases"][case] = c
        print("case", case, "done", file=sys.stderr)
    return out


def run_scanner(R):
    """bm::sparse_vector_scanner<bm::sparse_vector<unsigned, bvector<>>> over the SCANNER_* fixtures of cases.py, with and
    without NULL rows: find_gt / ge / lt / le / range / eq / zero / nonzero result vectors, find_eq first hits, and the
    counts of a batch of equality searches"""
    out = {}
    nw = (SCANNER_ROWS + 31) // 32
    for with_null in (False, True):
        vals, isn = scanner_values(with_null)
        sv = R.sparse_vector(vals, isn)
        assert sv.size() == SCANNER_ROWS
        c = {"rows": SCANNER_ROWS, "values_sha": sha(vals), "effective_slices": sv.effective_slices(),
             "null_sha": sha(isn) if isn is not None else None, "cmp": {}, "range": [], "eq_first": [], "eq_counts": []}
        planes = [sv.slice(i) for i in range(sv.effective_slices())]
        c["plane_counts"] = [p.count() if p is not None else None for p in planes]
        for pred, name in ((0, "gt"), (1, "ge"), (2, "lt"), (3, "le"), (5, "eq")):
            c["cmp"][name] = []
            for v in SCANNER_VALUES:
                r = sv.compare(pred, v)
                c["cmp"][name].append({"v": v, "count": r.count(), "sha": sha(r.to_words(nw))})
        for (a, b) in SCANNER_RANGES:
            r = sv.compare(4, a, b)
            c["range"].append({"from": a, "to": b, "count": r.count(), "sha": sha(r.to_words(nw))})
        for pred, name in ((6, "zero"), (7, "nonzero")):
            r = sv.compare(pred)
            c[name] = {"count": r.count(), "sha": sha(r.to_words(nw))}
        for v in SCANNER_VALUES:
            f, pos = sv.find_first_eq(v)
            c["eq_first"].append({"v": v, "found": f, "pos": pos if f else 0})
        c["eq_counts"] = [sv.compare(5, v).count() for v in SCANNER_EQ_BATCH]
        # IN-list find_eq(sv, start, end, bv_out) (:1399) and invert (:2321, applied to the find_gt(50) result)
        c["in_list"] = []
        for lst in SCANNER_IN_LISTS:
            r = sv.find_eq_in(lst)
            c["in_list"].append({"values": lst, "count": r.count(), "sha": sha(r.to_words(nw))})
        inv = sv.invert(sv.compare(0, 50))
        c["invert_gt50"] = {"count": inv.count(), "sha": sha(inv.to_words(nw))}
        out["with_null" if with_null else "no_null"] = c
        # the signed container over the same rows
        vals_s, isn_s = scanner_values_signed(with_null)
        svs = R.sparse_vector_signed(vals_s, isn_s)
        assert svs.size() == SCANNER_ROWS
        cs = {"rows": SCANNER_ROWS, "values_sha": sha(vals_s), "effective_slices": svs.effective_slices(), "cmp": {}, "range": []}
        cs["plane_counts"] = [(svs.slice(i).count() if svs.slice(i) is not None else None) for i in range(svs.effective_slices())]
        for pred, name in ((0, "gt"), (1, "ge"), (2, "lt"), (3, "le"), (5, "eq")):
            cs["cmp"][name] = []
            for v in SCANNER_S_VALUES:
                r = svs.compare(pred, v)
                cs["cmp"][name].append({"v": v, "count": r.count(), "sha": sha(r.to_words(nw))})
        for (a, b) in SCANNER_S_RANGES:
            r = svs.compare(4, a, b)
            cs["range"].append({"from": a, "to": b, "count": r.count(), "sha": sha(r.to_words(nw))})
        for pred, name in ((6, "zero"), (7, "nonzero")):
            r = svs.compare(pred)
            cs[name] = {"count": r.count(), "sha": sha(r.to_words(nw))}
        out["signed_with_null" if with_null else "signed_no_null"] = cs
    return out


def kinds_sha(v):
    return sha(np.asarray(v.flatten()[0], np.uint8))


def run_bm64(R):
    """vectors above 2^32 bits through the reference compiled with -DBM64ADDR"""
    vecs = [bm64_build(R, s_) for s_ in range(BM64_NVEC)]
    out = {"reference": R.name, "count": [v.count() for v in vecs], "kinds_sha": [kinds_sha(v) for v in vecs], "op2": {}}
    for op in range(4):
        t = R.op2(op, vecs[0], vecs[1], True)
        out["op2"][str(op)] = {"count": t.count(), "count_op": R.count_op2(op, vecs[0], vecs[1]), "kinds_sha": kinds_sha(t)}
    t = R.agg_and_sub(vecs[:3], vecs[3:])
    out["agg_and_sub"] = {"count": t.count(), "kinds_sha": kinds_sha(t), "find_first": list(R.find_first(t))}
    t = R.agg_or(vecs)
    out["agg_or"] = {"count": t.count()}
    t, f = R.agg_shift_right_and(vecs[:3], True, False)
    out["shift_right_and"] = {"count": t.count(), "found": bool(f), "kinds_sha": kinds_sha(t)}
    out["pipeline_counts"] = [int(x) for x in R.pipeline_counts([(vecs[:2], []), (vecs[:3], vecs[3:]), (vecs[1:], [])])]
    rs = R.rs_build(vecs[0])
    rq, sq = bm64_queries(vecs[0].count())
    pos, found = rs.select(sq)
    out["rs"] = {"count": rs.count(), "rank": [int(x) for x in rs.rank(rq)], "select_found": found.astype(int).tolist(),
                 "select_pos": [int(p_) if f_ else 0 for p_, f_ in zip(pos, found)]}
    return out


def main():
    oracle.build()
    P = oracle.port()
    g = {fl: run(oracle.reference(fl), P) for fl in ("avx2", "scalar")}
    a, s = g["avx2"], g["scalar"]
    assert a["cases"] == s["cases"], "AVX2 and scalar reference builds disagree"
    assert a["simd_version"] == 5 and s["simd_version"] == 0
    a["also_verified_with"] = s["reference"]
    a["bm64"] = run_bm64(oracle.reference("avx2_64"))
    sc_a, sc_s = run_scanner(oracle.reference("avx2")), run_scanner(oracle.reference("scalar"))
    assert sc_a == sc_s, "AVX2 and scalar scanner results disagree"
    a["scanner"] = sc_a
    with open(os.path.join(HERE, "golden_ref.json"), "w") as f:
        json.dump(a, f, separators=(",", ":"))
    print("wrote golden_ref.json", os.path.getsize(os.path.join(HERE, "golden_ref.json")), "bytes")


if __name__ == "__main__":
    main()
