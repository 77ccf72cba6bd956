#!/usr/bin/env python3
"""PCIe-inclusive side of the boundary: host block table -> device (bmx_vec_upload) and raw bits -> device
(bmx_vec_import_bits), 1e9-bit vector at 10 % density (125 MB of bit-blocks)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bitmagic_amd as bm
ctx = bm.context(0)
nbits = 1_000_000_000
v = bm.bvector.generate(ctx, 0xB17A61C, 1, 6554, nbits)
kinds, offs, bits, gaps = v.block_table()
words = v.to_words()
for name, fn in [("bmx_vec_upload (block table)", lambda: bm.bvector.from_block_table(ctx, nbits, kinds, offs, bits, gaps)),
                 ("bmx_vec_import_bits (raw words, optimize on device)", lambda: bm.bit_import_u32(ctx, words, True))]:
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); u = fn(); ctx.synchronize(); ts.append(time.perf_counter() - t0); del u
    best = min(ts[1:])
    print(json.dumps({"path": name, "MB": round(bits.nbytes / 1e6, 1), "ms_best": round(best * 1e3, 2), "GBps": round(bits.nbytes / best / 1e9, 2)}))
t0 = time.perf_counter(); w2 = v.to_words(); dt = time.perf_counter() - t0
print(json.dumps({"path": "bmx_vec_to_words (D2H)", "ms": round(dt * 1e3, 2), "GBps": round(w2.nbytes / dt / 1e9, 2)}))
