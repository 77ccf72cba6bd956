"""bmx_collection_prepare(ROLE_OR) over the configs[4] operand set (4096 x 4e9 bits at 0.02 %): build_ms of the collection"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitmagic_amd as bm
ctx = bm.context(0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vecs = [bm.bvector.generate(ctx, 0xB17A61C, 10000 + i, 13, 4_000_000_000) for i in range(nv)]
ctx.synchronize()
ms = []
for rep in range(3):
    ctx.collection_prepare(vecs[:nv - rep], bm.ROLE_OR); ctx.synchronize()       # (a different operand list every time: a new collection)
    ms.append(round(ctx.pack_stats()["last_build_ms"], 3))
print("BMX_DIAG_C2", os.environ.get("BMX_DIAG_C2"), "BMX_COLL_BUILD", os.environ.get("BMX_COLL_BUILD"), "build_ms", ms)
