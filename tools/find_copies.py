#!/usr/bin/env python3
"""Which python frames issue device-to-device copies / fills in one optimisation step (torch.profiler with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
T, args = bench.make_interpolator((64, 64, 64), "linear", "cuda", 0)
from deep_prior_interpolation_amd.optim import FusedAdam
T.optimizer = FusedAdam(T.net.parameters(), lr=args.lr)
def step():
    T.optimizer.zero_grad()
    T.optimization_loop()
    T.optimizer.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::add_", "aten::add"):
        st = [s for s in (e.stack or []) if "deep_prior" in s or "bench" in s or "torch/autograd" in s]
        cnt[(e.name, (st[0] if st else "?") + "  shapes=" + str(e.input_shapes)[:80])] += 1
for (n, s), c in cnt.most_common(25):
    print(c, n, s[:150])
