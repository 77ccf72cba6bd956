#!/usr/bin/env python3
"""Round 5: select over the configs[3] vector (4e9 bits, 10 % / 1 %) by batch size, random and ascending ranks, the two kernels
(directory summary in LDS: k_select_top; directory in global memory: k_select_sdir) and the lanes per query -- one JSON line each.
The table of profiles/r05_select/README.md."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
from bitmagic_amd import _ffi
L = _ffi.lib()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
for dq in (6554, 655):
    v = bm.bvector.generate(ctx, 0xB17A61C, 7, dq, 4_000_000_000)
    rs = v.build_rs_index(); cnt = rs.count()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for nq in (100_000, 1_000_000, 10_000_000, 100_000_000):
        qr = torch.randint(1, cnt + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        qs, _ = torch.sort(qr)
        pos = torch.zeros(nq, dtype=torch.int64, device="cuda"); found = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        out = {"density_q16": dq, "queries": nq}
        for name, top, lanes in (("top_2", 1, 2), ("top_4", 1, 4), ("sdir_2", 0, 2), ("sdir_4", 0, 4), ("default", -1, 0)):
            ctx.set_tuning("rs_select_top", top); ctx.set_tuning("rs_lanes", lanes)
            for label, qq in (("random", qr), ("sorted", qs)):
                fn = lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qq.data_ptr(), nq, pos.data_ptr(), found.data_ptr()))
                fn(); ctx.synchronize(); ctx.timer_start()
                for _ in range(5): fn()
                out["%s_%s_ms" % (name, label)] = round(ctx.timer_stop_ms() / 5, 4)
        chk = torch.zeros(nq, dtype=torch.int64, device="cuda")
        _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, pos.data_ptr(), nq, chk.data_ptr())); torch.cuda.synchronize()
        out["rank_of_select_ok"] = bool((chk == qs).all().item()) and bool(found.all().item())
        print(json.dumps(out), flush=True)
        del qr, qs, pos, found, chk
    del rs, v
