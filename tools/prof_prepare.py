"""rocprofv3 target: bmx_collection_prepare over the configs[4] operand set (4096 x 4e9 bits at 0.02 %), twice (OR role, AND role)"""
import sys, time
sys.path.insert(0, ".")
import bitmagic_amd as bm
ctx = bm.context(0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vecs = [bm.bvector.generate(ctx, 0xB17A61C, 10000 + i, 13, 4_000_000_000) for i in range(nv)]
ctx.synchronize()
for role in (bm.ROLE_OR, bm.ROLE_AND):
    t0 = time.perf_counter(); ctx.collection_prepare(vecs, role); ctx.synchronize()
    print("role", role, "wall ms", round((time.perf_counter() - t0) * 1e3, 2), ctx.pack_stats())
