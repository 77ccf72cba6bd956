"""combine_or over 4096 sparse vectors at lengths around a whole number of machine rounds of tiles (14 block columns per
workgroup, 256 CUs): does the kernel pay a whole extra round for a few tail tiles?  (round 4)"""
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
from bitmagic_amd import _ffi

def main():
    ctx = bm.context(0)
    L = _ffi.lib()
    nvec = 4096
    for ntiles in ([int(a) for a in sys.argv[1:]] or (4352, 4360, 4480, 4608)):
        nbits = min(ntiles * 14 * 65536, 4_000_000_000) if ntiles == 4360 else ntiles * 14 * 65536
        vecs = [bm.bvector.generate(ctx, 1234, 10000 + i, 13, nbits) for i in range(nvec)]
        arr = (C.c_void_p * nvec)(*[v._h for v in vecs])
        def call():
            h = C.c_void_p()
            _ffi.check(L.bmx_agg_or_opt(ctx._h, arr, nvec, 0, C.byref(h)))
            return bm.bvector(ctx, h)
        for _ in range(3): call()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        for _ in range(10): r = call()
        ctx.synchronize(); ms = (time.perf_counter() - t0) * 100
        gb = sum(v.operand_bytes() for v in vecs) / 1e9
        print(json.dumps({"ntiles": (nbits + 65535) // 65536 // 14 + ((nbits + 65535) // 65536 % 14 != 0), "nbits": nbits, "ms": round(ms, 4), "rounds": round(ntiles / 256, 3),
                          "GB": round(gb, 3), "ms_per_round": round(ms / (ntiles / 256), 4), "count": r.count(), "diag": os.environ.get("BMX_DIAG_ROWS", "")}), flush=True)
        del vecs, arr, r
main()
