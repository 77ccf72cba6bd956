#!/usr/bin/env python3
"""Streaming-read rate of the 256-stream access pattern under every cache policy of a raw buffer load
(sc0 / sc1 / nt bits) next to the global_load paths of the product kernels (plain, nt)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitmagic_amd as bm
from bitmagic_amd import _ffi
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(0, s.cuda_stream)
names = {0: "global plain", 1: "global nt", 100: "buffer -", 101: "buffer sc0", 102: "buffer nt", 103: "buffer sc0 nt",
         116: "buffer sc1", 117: "buffer sc0 sc1", 118: "buffer sc1 nt", 119: "buffer sc0 sc1 nt"}
for rnd in range(2):
    for nt, name in names.items():
        ms = C.c_float()
        _ffi.check(_ffi.lib().bmx_diag_stream_read(ctx._h, 16 << 30, nt, 256, 1, 5, C.byref(ms)))
        print(f"round {rnd} {name:20s} {ms.value:.4f} ms  {(16 << 30) / ms.value / 1e6:.0f} GB/s")
