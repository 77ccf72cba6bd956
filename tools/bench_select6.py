#!/usr/bin/env python3
"""Round 6: select through the select lines (k_select_sel, bmx_kernels11.h: one 128-byte line per query, no search) next to the
round-5 kernel (k_select_top: directory summary in LDS, interpolated guess verified by the line header) and rank, on the configs[3]
vector (4e9 bits) at 10 % / 1 % / 0.1 % and at 50 % (too dense for select lines under the memory policy: the `select_lines` column is then
k_select_top with its round-6 position-exact summary) -- one JSON line per (density, batch).  Also what build_rs_index costs with and without the
select lines and what the index holds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
from bitmagic_amd import _ffi
L = _ffi.lib()
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
dens = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("6554", "655", "66", "32768"))]
batches = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("100000", "1000000", "10000000", "100000000"))]
def avg(fn, n=5):
    fn(); ctx.synchronize(); ctx.timer_start()
    for _ in range(n): fn()
    return round(ctx.timer_stop_ms() / n, 4)
for dq in dens:
    v = bm.bvector.generate(ctx, 0xB17A61C, 7, dq, 4_000_000_000)
    build = {}
    for name, sel in (("without_select_lines", 0), ("policy", -1)):
        ctx.set_tuning("rs_select_sel", sel)
        build[name + "_ms"] = avg(lambda: v.build_rs_index(), 3)
    rs = v.build_rs_index(); cnt = rs.count()
    print(json.dumps({"density_q16": dq, "count": cnt, "vector": v.info(), "index": rs.info(), "build_rs_index": build}), flush=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for nq in batches:
        qr = torch.randint(1, cnt + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
        qn = torch.randint(0, 4_000_000_000, (nq,), device="cuda", dtype=torch.int64, generator=g)
        qs, _ = torch.sort(qr)
        pos = torch.zeros(nq, dtype=torch.int64, device="cuda"); found = torch.zeros(nq, dtype=torch.uint8, device="cuda")
        out = {"density_q16": dq, "queries": nq}
        out["rank_ms"] = avg(lambda: _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, qn.data_ptr(), nq, pos.data_ptr())))
        for name, sel, top in (("select_lines", -1, -1), ("top", 0, 1), ("r05_default", 0, -1)):
            ctx.set_tuning("rs_select_sel", sel); ctx.set_tuning("rs_select_top", top)
            for label, qq in (("random", qr), ("sorted", qs)):
                ctx.set_tuning("rs_sorted_hint", 1 if (label == "sorted" and name == "r05_default") else 0)
                out["%s_%s_ms" % (name, label)] = avg(lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qq.data_ptr(), nq, pos.data_ptr(), found.data_ptr())))
        ctx.set_tuning("rs_select_sel", -1); ctx.set_tuning("rs_select_top", -1); ctx.set_tuning("rs_sorted_hint", 0)
        _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qr.data_ptr(), nq, pos.data_ptr(), found.data_ptr()))
        chk = torch.zeros(nq, dtype=torch.int64, device="cuda")
        _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, pos.data_ptr(), nq, chk.data_ptr())); torch.cuda.synchronize()
        out["rank_of_select_ok"] = bool((chk == qr).all().item()) and bool(found.all().item())
        print(json.dumps(out), flush=True)
        del qr, qs, qn, pos, found, chk
    del rs, v
