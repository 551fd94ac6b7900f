#!/usr/bin/env python3
"""Diagnostics of the bf16-storage mode against fp32 on one patch: forward output, loss, per-tensor weight-gradient error for
(arithmetic, storage) = (fp32, bf16), (bf16, fp32), (bf16, bf16); then short optimisation runs.  python tools/diag_storage.py [D H W]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_prior_interpolation_amd import ops, utils as u  # noqa: E402
from deep_prior_interpolation_amd.main import Interpolator  # noqa: E402
from deep_prior_interpolation_amd.parameter import parse_arguments  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 32, 64)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 60


def make(prec, epochs):
    vol = u.hyperbolic_volume(shape, seed=3)
    mask = u.random_trace_mask(shape, 0.5, seed=4)
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "16", "--upsample", "linear",
                            "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--epochs", str(epochs), "--gpu", "0", "--precision", prec])
    u.set_seed(7)
    T = Interpolator(args, "/tmp", seed=7)
    T.load_data({"image": (vol.astype(np.float64) * 40)[..., None], "mask": mask.astype(np.float64)[..., None], "name": "0"})
    T.build_model()
    T.build_input()
    return T


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


T = make("fp32", 1)
z = T.input_.clone()
res = {}
for name, arith, store in (("fp32", "fp32", "fp32"), ("fp32 arithmetic, bf16 storage", "fp32", "bf16"), ("bf16 operands, fp32 storage", "bf16mm", "fp32"),
                           ("bf16 operands, bf16 storage", "bf16mm", "bf16")):
    ops.set_precision(arith)
    ops.set_storage(store)
    T.net.zero_grad()
    o = T.net(z.to(torch.bfloat16) if store == "bf16" else z)
    loss, _ = ops.masked_loss(o, T.img_, T.mask_, "mae")
    loss.backward()
    res[name] = (o.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in T.net.named_parameters() if p.grad is not None})
o0, l0, g0 = res["fp32"]
for name, (o, l, g) in res.items():
    rels = {k: rel(g[k], g0[k]) for k in g0 if float(g0[k].norm()) > 0}
    cos = {k: float((g[k].double().flatten() @ g0[k].double().flatten()) / (g[k].double().norm() * g0[k].double().norm() + 1e-300)) for k in rels}
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
    print("%-34s output rel %.3e  loss %.6f (%.2e)  grad rel median %.3e max %.3e  min cos %.4f  worst %s" % (
        name, rel(o, o0), l, abs(l - l0) / l0, float(np.median(list(rels.values()))), max(rels.values()), min(cos.values()),
        [(k, round(v, 3)) for k, v in worst]))
ops.set_precision("fp32")
ops.set_storage("fp32")
for prec in ("fp32", "bf16mm", "bf16"):
    T = make(prec, iters)
    T.optimize(verbose=False, mode="eager")
    h = T.history
    print("%-7s loss %s  snr %s" % (prec, np.round([h.loss[i] for i in (0, 9, 19, 39, iters - 1)], 4), np.round([h.snr[i] for i in (0, 9, 19, 39, iters - 1)], 2)))


# ---- where does the difference come from?  (a) error of every block output along the net, storage mode vs fp32; (b) the same net in fp32
# storage with its block outputs rounded to bf16 by hand (forward hooks): a LOWER bound of what rounding alone does to output and gradients
def block_outputs(T, z, store):
    outs = []
    hooks = [m.register_forward_hook(lambda mod, i, o: outs.append(o.detach().float().clone())) for m in T.net.modules()
             if type(m).__name__ in ("MultiResBlock", "SkipConcat")]
    ops.set_precision("fp32")
    ops.set_storage(store)
    T.net.zero_grad()
    o = T.net(z.to(torch.bfloat16) if store == "bf16" else z)
    loss, _ = ops.masked_loss(o, T.img_, T.mask_, "mae")
    loss.backward()
    for h in hooks:
        h.remove()
    return outs, o.detach().clone(), {k: p.grad.detach().clone() for k, p in T.net.named_parameters() if p.grad is not None}


T = make("fp32", 1)
z = T.input_.clone()
o32, out32, g32 = block_outputs(T, z, "fp32")
o16, out16, g16 = block_outputs(T, z, "bf16")
print("block / join outputs in forward order, rel error of the storage mode:", [round(rel(a, b), 4) for a, b in zip(o16, o32)])
ops.set_storage("fp32")
hooks = [m.register_forward_hook(lambda mod, i, o: o.to(torch.bfloat16).float()) for m in T.net.modules() if type(m).__name__ == "MultiResBlock"]
T.net.zero_grad()
o = T.net(z)
loss, _ = ops.masked_loss(o, T.img_, T.mask_, "mae")
loss.backward()
for h in hooks:
    h.remove()
gh = {k: p.grad.detach().clone() for k, p in T.net.named_parameters() if p.grad is not None}
convw = [k for k in g32 if g32[k].ndim == 5]
print("fp32 kernels, block outputs rounded by hand (forward only): output rel %.3e, conv-weight grad rel median %.3e max %.3e" % (
    rel(o, out32), float(np.median([rel(gh[k], g32[k]) for k in convw])), max(rel(gh[k], g32[k]) for k in convw)))
print("storage mode: output rel %.3e, conv-weight grad rel median %.3e max %.3e; per conv weight: %s" % (
    rel(out16, out32), float(np.median([rel(g16[k], g32[k]) for k in convw])), max(rel(g16[k], g32[k]) for k in convw),
    [(k, round(rel(g16[k], g32[k]), 3)) for k in convw]))
