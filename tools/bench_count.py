import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bitmagic_amd as bm
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
for nbits in (1_000_000_000, 4_000_000_000):
    vs = [bm.bvector.generate(ctx, 0xB17A61C, 700 + i, 6554, nbits) for i in range(6)]     # rotation: HBM-cold
    out = {"nbits": nbits}
    for ps in (0, -1):
        ctx.set_tuning("pair_stream", ps)
        ref = [v.count() for v in vs]
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            for v in vs: v.count()
        out["count_ms_stream%d" % ps] = round((time.perf_counter() - t0) / 120 * 1e3, 4)
        out["c%d" % ps] = ref[0]
    print(json.dumps(out))
