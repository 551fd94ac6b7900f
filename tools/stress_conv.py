"""Run-to-run determinism stress of one conv configuration (fwd / bwd_data / bwd_weight), to flush out races."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_prior_interpolation_amd import ops
cases = [(64, 4, (16, 48, 64), 3, 1), (13, 4, (32, 32, 40), 3, 1), (25, 16, (32, 32, 64), 3, 1), (25, 25, (16, 32, 64), 3, 2), (64, 25, (16, 32, 64), 1, 1)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for cin, cout, shp, k, s in cases:
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((1, cin) + shp, device="cuda", generator=gen)
    w = torch.randn((cout, cin, k, k, k), device="cuda", generator=gen) * 0.05
    b = torch.randn(cout, device="cuda", generator=gen)
    d = ops.make_desc(x, w, s)
    osh = ops.desc_out_dims(d)
    dy = torch.randn((1, cout) + osh, device="cuda", generator=gen)
    ref = {}
    bad = {"fwd": 0, "bwd_data": 0, "bwd_weight": 0}
    for it in range(reps):
        junk = torch.full((1 << 22,), float("nan"), device="cuda")     # poison freed memory
        del junk
        y = torch.empty((1, cout) + osh, device="cuda"); ops.raw_conv_fwd(d, x, None, w, b, y)
        dx = torch.empty_like(x); ops.raw_conv_bwd_data(d, dy, w, dx)
        dw = torch.empty_like(w); ops.raw_conv_bwd_weight(d, x, None, dy, dw)
        for name, t in (("fwd", y), ("bwd_data", dx), ("bwd_weight", dw)):
            if it == 0:
                ref[name] = t.clone()
            elif not torch.equal(ref[name], t):
                bad[name] += 1
                if bad[name] == 1:
                    diff = (ref[name] - t).abs()
                    print("  MISMATCH", name, "max", float(diff.max()), "count", int((diff > 0).sum()), "nan", int(torch.isnan(t).sum()))
    print((cin, cout, shp, k, s), "mismatching runs:", bad)
