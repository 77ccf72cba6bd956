"""materialised pairwise operations on two 1e9-bit vectors at 1 %: host-call and kernel-side time per op (A/B of library builds via BMX_LIB)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
ctx = bm.context(0)
dq = int(sys.argv[1]) if len(sys.argv) > 1 else 655
pairs = [(bm.bvector.generate(ctx, 1234, 2 * i, dq, 1_000_000_000), bm.bvector.generate(ctx, 1234, 2 * i + 1, dq, 1_000_000_000)) for i in range(6)]
out = {}
for op, name in ((0, "and"), (1, "or"), (2, "xor"), (3, "sub")):
    for _ in range(12):
        for a, b in pairs: r = bm.bvector._op2(op, a, b, bm.opt_none)
    ctx.synchronize(); t0 = time.perf_counter()
    n = 0
    for _ in range(40):
        for a, b in pairs: r = bm.bvector._op2(op, a, b, bm.opt_none); n += 1
    ctx.synchronize()
    out[name] = round((time.perf_counter() - t0) / n * 1e3, 4)
print(json.dumps({"lib": os.environ.get("BMX_LIB", "default"), "dq": dq, "ms_per_call": out}))
