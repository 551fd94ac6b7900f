#!/usr/bin/env python3
"""Which rounding do the bf16-mode kernels apply to their WEIGHTS?  Forward of a layer with bf16 input / output tensors and generic fp32 weights against the
fp64 oracle with the weights (a) rounded to nearest-even, (b) truncated, (c) left fp32.   python tools/diag_bf16_weights.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from deep_prior_interpolation_amd import ops  # noqa: E402

BF = torch.bfloat16
nrm = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())


def trunc(w):
    return (w.view(torch.int32) & -65536).view(torch.float32)


def main():
    g = torch.Generator().manual_seed(5)
    for (cin, cout, k, s, shape) in [(16, 16, 3, 2, (32, 32, 64)), (25, 25, 3, 2, (32, 32, 64)), (25, 16, 3, 1, (16, 32, 64)), (16, 25, 1, 1, (16, 32, 64))]:
        x = torch.randn((1, cin) + shape, generator=g).to(BF)
        w = torch.randn((cout, cin, k, k, k), generator=g) * 0.1
        b = torch.randn(cout, generator=g) * 0.1
        with ops.mode_scope("bf16", "bf16"):
            d = ops.make_desc(x.cuda(), w.cuda(), s, BF)
            Do, Ho, Wo = ops.desc_out_dims(d)
            y = torch.empty((1, cout, Do, Ho, Wo), dtype=BF, device="cuda")
            ops.raw_conv_fwd(d, x.cuda(), None, w.cuda(), b.cuda(), y)
        torch.cuda.synchronize()
        y = y.float().cpu()
        ref = lambda ww: F.conv3d(x.double(), ww.double(), b.double(), stride=s, padding=(k - 1) // 2).to(BF).float()
        print("%d->%d k%d s%d: vs RNE weights %.3e | truncated %.3e | fp32 weights %.3e" % (cin, cout, k, s, nrm(y, ref(w.to(BF).float())), nrm(y, ref(trunc(w))), nrm(y, ref(w))))


def chained():
    """... and to the CHAINED input T(x) = qs * act(ps * x + pb) + qb of a 3x3x3 layer: re-rounded to bf16 (the MFMA operand) or not?  Two sizes: the big-tile
    variant (>= 512 workgroups) and the row-band variant of the coarse levels."""
    g = torch.Generator().manual_seed(6)
    for (cin, cout, shape) in [(8, 17, (32, 32, 64)), (8, 17, (16, 16, 32)), (17, 26, (16, 16, 32)), (25, 8, (16, 16, 32)), (8, 17, (8, 8, 16))]:
        x = torch.randn((1, cin) + shape, generator=g).to(BF)
        w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1).to(BF).float()
        ch = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3, torch.full((cin,), 0.2), torch.rand(cin, generator=g) + 0.5,
                          torch.randn(cin, generator=g) * 0.3], dim=1).contiguous()
        with ops.mode_scope("bf16", "bf16"):
            d = ops.make_desc(x.cuda(), w.cuda(), 1, BF)
            y = torch.empty((1, cout) + shape, dtype=BF, device="cuda")
            ops.raw_conv_fwd(d, x.cuda(), ch.cuda().flatten(), w.cuda(), None, y)
        torch.cuda.synchronize()
        y = y.float().cpu()
        bc = lambda v: v.view(1, -1, 1, 1, 1).double()
        t = bc(ch[:, 0]) * x.double() + bc(ch[:, 1])
        t = bc(ch[:, 3]) * torch.where(t >= 0, t, t * bc(ch[:, 2])) + bc(ch[:, 4])
        ref = lambda tt: F.conv3d(tt, w.double(), None, padding=1).to(BF).float()
        t32 = t.float()
        print("%d->%d @%s with input chain: vs T(x) rounded to bf16 %.3e | T(x) in fp32 %.3e" % (cin, cout, shape, nrm(y, ref(t32.to(BF).double())), nrm(y, ref(t))))


if __name__ == "__main__":
    chained()
    main()
