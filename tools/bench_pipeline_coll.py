"""many arg-groups over the vectors of ONE prepared collection (k_coll_members) against the descriptor-table kernels:
counts-only pipelines of G groups, each AND a few + SUB many GAP-only vectors (the sparse_vector_scanner shape over sparse planes)"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import bitmagic_amd as bm

def run(nvec, dq, nbits, ngroups, n_and, n_sub, prepared):
    ctx = bm.context(0)
    if not prepared: ctx.set_tuning("gap_pack", 0)
    vecs = [bm.bvector.generate(ctx, 0xB17A61C, 500 + i, dq, nbits) for i in range(nvec)]
    ctx.synchronize()
    if prepared:
        ctx.collection_prepare(vecs, bm.ROLE_OR); ctx.collection_prepare(vecs, bm.ROLE_AND)
    rng = np.random.default_rng(3)
    pipe = bm.aggregator.pipeline(ctx)
    for g in range(ngroups):
        ag = pipe.add()
        sel = rng.permutation(nvec)[: n_and + n_sub]
        for i in sel[:n_and]: ag.add(vecs[int(i)], 0)
        for i in sel[n_and:]: ag.add(vecs[int(i)], 1)
    pipe.complete()
    agg = bm.aggregator(ctx)
    c = agg.combine_and_sub(pipe).copy()
    ctx.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): agg.combine_and_sub(pipe)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    d = pipe.describe()
    ob = pipe.operand_bytes()
    del pipe, vecs
    ctx.close()
    return ms, c, d, ob

if __name__ == "__main__":
    for (nvec, dq, nbits, G, na, ns) in [(512, 66, 1_000_000_000, 64, 2, 30), (512, 66, 1_000_000_000, 64, 20, 100), (4096, 13, 4_000_000_000, 16, 1, 256)]:
        a_ms, a_c, a_d, ob = run(nvec, dq, nbits, G, na, ns, True)
        b_ms, b_c, b_d, _ = run(nvec, dq, nbits, G, na, ns, False)
        print(json.dumps({"nvec": nvec, "dq": dq, "nbits": nbits, "groups": G, "and": na, "sub": ns, "operand_GB": round(ob / 1e9, 2),
                          "prepared_ms": round(a_ms, 3), "prepared_kernel": a_d, "tables_ms": round(b_ms, 3), "tables_kernel": b_d,
                          "same_counts": bool((a_c == b_c).all())}))
