#!/usr/bin/env python3
"""Runs the dominant conv back to back for N seconds (clock / power sampling with rocm-smi from another process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_prior_interpolation_amd import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
shp = (256, 128, 128)
x = torch.randn((1, 25) + shp, device="cuda")
w = torch.randn((16, 25, 3, 3, 3), device="cuda") * 0.05
b = torch.randn(16, device="cuda")
y = torch.empty((1, 16) + shp, device="cuda")
d = ops.make_desc(x, w, 1)
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        ops.raw_conv_fwd(d, x, None, w, b, y)
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
print("launches", n, "avg ms", 1e3 * dt / n, "TF", 2 * 25 * 27 * 16 * shp[0] * shp[1] * shp[2] * n / dt / 1e12)
