#!/usr/bin/env python3
"""Launches (a) the dominant conv (fwd 25->16, 3x3x3, 256x128x128) and (b) two calibration kernels with KNOWN HBM byte
counts and the same access widths (dpi_chain_apply: float4 loads/stores; dpi_crop_copy: dword loads/stores), so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be corrected as MI355X_MICROARCH.md prescribes.  Run under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_prior_interpolation_amd import _lib, ops
L = _lib.load()
shp = (256, 128, 128)
V = shp[0] * shp[1] * shp[2]
x = torch.randn((1, 25) + shp, device="cuda")
w = torch.randn((16, 25, 3, 3, 3), device="cuda") * 0.05
b = torch.randn(16, device="cuda")
y = torch.empty((1, 16) + shp, device="cuda")
d = ops.make_desc(x, w, 1)
big = torch.randn((1, 25) + shp, device="cuda")     # > 256 MiB total working set between launches: defeats the Infinity Cache
out = torch.empty_like(big)
chain = ops.slope_chain(25, 0.2, "cuda")
for _ in range(3):
    ops.raw_conv_fwd(d, x, None, w, b, y)                                   # conv: algorithmic 25V*4 read + 16V*4 written
    ops.raw_chain_apply(big, chain, 25, V, out)                             # known: 25V*4 read, 25V*4 written (float4)
    _lib.check(L.dpi_crop_copy(_lib.ptr(big), 25, *shp, 0, 0, 0, *shp, _lib.ptr(out), _lib.stream()))   # known, dword
torch.cuda.synchronize()
print("V", V, "conv_alg_read", 25 * V * 4, "conv_alg_write", 16 * V * 4, "calib_bytes", 25 * V * 4)
