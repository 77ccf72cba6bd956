#!/usr/bin/env python3
"""The headline workload through the in-library device group (bmx_group: one process, one host thread, n members):
256 x 1e9-bit vectors sharded by block range over the members, counts-only pipeline, counts summed on the host
(or with --rccl by the in-library RCCL all-reduce when the members are distinct devices).
On a 1-GPU box the members share the GPU (--members 8 = eight streams on device 0): what this measures there is the
host-side cost of driving n members from one thread -- n launches, n 8-byte read-backs, n synchronises -- next to the
single-context step.   python tools/bench_group.py --members 1,2,4,8"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bitmagic_amd as bm

ap = argparse.ArgumentParser()
ap.add_argument("--members", default="1,2,4,8")
ap.add_argument("--devices", default="", help="explicit device list (e.g. 0,1,2,3,4,5,6,7 on an 8-GPU node); default: device 0 repeated")
ap.add_argument("--nvec", type=int, default=256)
ap.add_argument("--nbits", type=int, default=1_000_000_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rccl", action="store_true")
a = ap.parse_args()
for m in [int(x) for x in a.members.split(",")]:
    devs = [int(x) for x in a.devices.split(",")][:m] if a.devices else [0] * m
    grp = bm.group(devs, bm.GROUP_RCCL if a.rccl else bm.GROUP_HOST_SUM)
    vecs = [bm.gbvector.generate(grp, 0xB17A61C, v, 6554, a.nbits, with_common=True) for v in range(a.nvec)]
    agg = bm.gaggregator(grp)
    pipe = bm.gaggregator.pipeline(grp)
    g = pipe.add()
    for v in vecs: g.add(v, 0)
    pipe.complete()
    for _ in range(3): cnt = agg.combine_and_sub(pipe)
    t0 = time.perf_counter()
    for _ in range(a.steps): cnt = agg.combine_and_sub(pipe)
    dt = (time.perf_counter() - t0) / a.steps * 1e3
    ms = pipe.last_ms()
    print(json.dumps({"members": m, "devices": devs, "rccl": a.rccl, "ms_per_step_host": round(dt, 4), "count": int(cnt[0]),
                      "member_device_ms": [round(x, 4) for x in ms], "Tbit_s": round(a.nvec * a.nbits / dt / 1e9, 2)}), flush=True)
    del pipe, vecs, agg
    grp.close()
