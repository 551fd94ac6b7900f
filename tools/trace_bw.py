#!/usr/bin/env python3
"""Phase timeline of the backward-weight MFMA kernel (25 -> 16, 256x128x128); needs a -DDPI_TRACE build."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_prior_interpolation_amd import ops, _lib
L = _lib.load()
shp = (256, 128, 128)
x = torch.randn((1, 25) + shp, device="cuda")
w = torch.randn((16, 25, 3, 3, 3), device="cuda") * 0.05
dy = torch.randn((1, 16) + shp, device="cuda")
dw = torch.empty_like(w)
d = ops.make_desc(x, w, 1)
for _ in range(10):
    ops.raw_conv_bwd_weight(d, x, None, dy, dw)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L.dpi_debug_read_trace.restype = ctypes.c_int
print("rc", L.dpi_debug_read_trace(buf))
for blk in range(2):
    t = [buf[blk * 64 + i] for i in range(64)]
    print("block", blk, "prologue", t[1] - t[0])
    i = 1
    while i + 4 < 64 and t[i + 4] > 0:
        print("   tile: barrier-wait %6d  store+sync %6d  dy loads+prefetch issue %6d  mfma rows %6d" % (
            t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2], t[i + 4] - t[i + 3]))
        i += 4
