#!/usr/bin/env python3
"""Phase timeline (shader clocks) of a few workgroups of the dominant conv; needs a build with -DDPI_TRACE."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_prior_interpolation_amd import ops, _lib
L = _lib.load()
CIN, COUT = int(os.environ.get("TR_CIN", 25)), int(os.environ.get("TR_COUT", 16))
shp = tuple(int(v) for v in os.environ.get("TR_SHAPE", "256,128,128").split(","))
x = torch.randn((1, CIN) + shp, device="cuda")
w = torch.randn((COUT, CIN, 3, 3, 3), device="cuda") * 0.05
b = torch.randn(COUT, device="cuda")
y = torch.empty((1, COUT) + shp, device="cuda")
STRIDE = int(os.environ.get("TR_STRIDE", 1))
d = ops.make_desc(x, w, STRIDE)
y = torch.empty((1, COUT) + ops.desc_out_dims(d), device="cuda")
for _ in range(20):
    ops.raw_conv_fwd(d, x, None, w, b, y)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L.dpi_debug_read_trace.restype = ctypes.c_int
print("rc", L.dpi_debug_read_trace(buf))
for blk in range(4):
    t = [buf[blk * 64 + i] for i in range(64)]
    t0 = t[0]
    print("block slot", blk, "prologue", t[1] - t0, "total", t[41] - t0, "epilogue", t[41] - t[40])
    for c in range(min(9, (CIN + 3) // 4)):
        o = 2 + c * 4
        print("   chunk %d: wait-barrier1 %6d  stage_store %6d  barrier2 %6d  mfma-phase %6d" % (
            c, t[o] - (t[o - 1] if c else t[1]), t[o + 1] - t[o], t[o + 2] - t[o + 1], t[o + 3] - t[o + 2]))

import collections
big = (ctypes.c_longlong * (8192 * 4))()
L.dpi_debug_read_blocks.restype = ctypes.c_int
print("rc", L.dpi_debug_read_blocks(big))
NB = int(os.environ.get('TR_BLOCKS', 4096))
rows = [(big[i * 4], big[i * 4 + 1], big[i * 4 + 2], big[i * 4 + 3]) for i in range(NB)]
t0 = min(r[0] for r in rows)
print("kernel span us", (max(r[1] for r in rows) - t0) / 100.0, "mean block us", sum(r[1] - r[0] for r in rows) / len(rows) / 100.0)
percu = collections.defaultdict(list)
for i, (a, b, hw, xcc) in enumerate(rows):
    percu[(xcc & 0xf, hw >> 8)].append((a - t0, b - t0, i, hw & 0xff))     # hw bits: wave/simd in low byte, cu/sh/se above
print("distinct (xcc, hw>>8) units", len(percu))
k = sorted(percu)[0]
print("unit", k, "blocks", len(percu[k]))
for a, b, i, lo in sorted(percu[k])[:12]:
    print("   blk %5d  start %8.1f us  end %8.1f us  hw_lo %#x" % (i, a / 100.0, b / 100.0, lo))
# max concurrency per unit
mx = collections.Counter()
for k, v in percu.items():
    ev = sorted([(a, 1) for a, b, _, _ in v] + [(b, -1) for a, b, _, _ in v])
    c = m = 0
    for _, dlt in ev:
        c += dlt; m = max(m, c)
    mx[m] += 1
print("max concurrent blocks per unit histogram", dict(mx))
