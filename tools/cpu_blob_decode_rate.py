#!/usr/bin/env python3
"""What SURVEY section 8(f)-3's second leg (decode serialised BLOBs on the device) would be about, MEASURED with the
reference itself on the host (CPU only; needs oracle/_ref, i.e. /root/reference at build time):

  * how big a bm::serializer BLOB is next to the blocks the vector holds in memory (= what the frozen-arena upload of
    DESIGN.md section 2.5 moves over PCIe), per density;
  * how fast bm::deserialize turns it back into blocks on ONE host core.

A device-side decoder can only save the PCIe difference (arena bytes - BLOB bytes) and would have to beat the host
decoder's rate on a format that is one sequential, variable-length token stream per vector (block boundaries are only
known after decoding the blocks before them; bookmarks are optional and off by default).  Prints one JSON line per case."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle

if not oracle.have_reference():
    sys.exit("oracle/_ref/libbmref_avx2.so is absent (built from /root/reference by `make -C oracle`)")
P, R = oracle.port(), oracle.reference()
L = R.lib
L.ref_serialize.restype = C.c_uint64
L.ref_deserialize_timed.restype = C.c_void_p
nbits = int(os.environ.get("NBITS", 250_000_000))
PCIE_GBPS = 54.0          # frozen-arena upload rate measured on the GPU box (2.3 ms per 1e9 bits)
for dq, label in ((32768, "50 %"), (6554, "10 %"), (655, "1 %"), (66, "0.1 %"), (13, "0.02 %")):
    w = P.gen_words(2024, 7, dq, nbits)
    v = R.import_words(w, True, nbits)
    counts, gap_words = v.stat()                               # [NULL, FULL, BIT, GAP] blocks, u16 words of all GAP blocks
    nbit, ngap = counts[2], counts[3]
    arena = nbit * 8192 + 2 * gap_words                        # bit-blocks + GAP blocks, as the frozen-arena upload moves them
    for level in (5, 4):
        size = int(L.ref_serialize(C.c_void_p(v.h), C.c_uint(level), None, C.c_uint64(0)))
        buf = (C.c_ubyte * size)()
        L.ref_serialize(C.c_void_p(v.h), C.c_uint(level), buf, C.c_uint64(size))
        best = C.c_double()
        h = L.ref_deserialize_timed(buf, C.c_uint(3), C.byref(best))
        back = oracle.Vec(R, h, nbits)
        ok = back.count() == v.count() and bool(R.lib.ref_vec_equal(C.c_void_p(back.h), C.c_void_p(v.h)))
        t_host = best.value
        t_pcie_arena = arena / (PCIE_GBPS * 1e9)
        t_pcie_blob = size / (PCIE_GBPS * 1e9)
        print(json.dumps({"nbits": nbits, "density": label, "serializer_level": level, "bit_blocks": nbit, "gap_blocks": ngap,
                          "arena_MB": round(arena / 1e6, 2), "blob_MB": round(size / 1e6, 2), "blob_over_arena": round(size / max(arena, 1), 3),
                          "host_deserialize_ms_1core": round(t_host * 1e3, 2),
                          "host_deserialize_GBps_of_blob": round(size / t_host / 1e9, 3),
                          "host_deserialize_GBps_of_blocks": round(arena / t_host / 1e9, 3),
                          "pcie_ms_arena": round(t_pcie_arena * 1e3, 3), "pcie_ms_blob": round(t_pcie_blob * 1e3, 3),
                          "pcie_ms_a_device_decoder_could_save": round((t_pcie_arena - t_pcie_blob) * 1e3, 3),
                          "roundtrip_equal": ok}))
        del back
    del v
