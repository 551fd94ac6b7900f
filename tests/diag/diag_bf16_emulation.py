#!/usr/bin/env python3
"""Diagnostic behind tests/test_gpu_bf16_storage.py::test_whole_net_in_storage_mode_against_the_fp64_oracle_with_the_same_rounding_points:
block-by-block distance between the HIP net in bf16 storage mode and the fp64 oracle with emulated rounding points (and a few variants of
the emulation), so that a wrong rounding point shows up at the first block it affects.   python tests/diag/diag_bf16_emulation.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import dpi_oracle as O  # noqa: E402
from test_gpu_bf16_storage import _net_run  # noqa: E402

BF = torch.bfloat16
rb = lambda t: t.to(BF).to(t.dtype)
nrm = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())


def main():
    shape = (32, 32, 64)
    T = _net_run(shape, "bf16", 1, extra=["--filters", "16", "32", "64", "--skip", "16", "32"], inputdepth=16)
    init = {k: v.detach().cpu().clone() for k, v in T.net.state_dict().items()}
    gen = torch.Generator().manual_seed(11)
    x = (0.1 * torch.randn((1, 16) + shape, generator=gen)).to(BF).float()
    cfg = {"ndim": 3, "filters": T.args.filters, "skip": T.args.skip, "upsample": "trilinear"}
    hip = []
    hooks = [m.register_forward_hook(lambda mod, i, o: hip.append(o.detach().float().cpu())) for m in T.net.modules() if type(m).__name__ == "MultiResBlock"]
    from deep_prior_interpolation_amd import ops
    cba_out, orig_cba = [], ops.conv_bn_act

    def rec_cba(*a, **k):
        y = orig_cba(*a, **k)
        cba_out.append(y.detach().float().cpu())
        return y
    ops.conv_bn_act = rec_cba
    name_of = {p.data_ptr(): n for n, p in T.net.named_parameters()}
    raw_hip, orig_raw = {}, ops._cba_raw

    def rec_raw(d, xx, in_chain, w, b, bn, slope, r_out, mi_out, chain_out):
        orig_raw(d, xx, in_chain, w, b, bn, slope, r_out, mi_out, chain_out)
        raw_hip[name_of[w.data_ptr()][:-len(".weight")]] = (r_out.detach().float().cpu(), mi_out.detach().float().cpu())
    ops._cba_raw = rec_raw
    ops.BRANCH_SKIP = False
    with T.precision_scope():
        out = T.net(x.cuda().to(BF)).float().cpu()
    ops.conv_bn_act = orig_cba
    ops._cba_raw = orig_raw
    for h in hooks:
        h.remove()
    orig = {n: getattr(O, n) for n in ("block3d", "respath3d", "upsample2x")}

    def run(round_w3=True, round_w1=False, round_x=True, round_store=True, round_blocks=True):
        class St(O.NetState):
            def conv(self, key, xin, stride=1):
                w, b = self.P[key + ".weight"], self.P.get(key + ".bias")
                if round_x:
                    xin = rb(xin)
                if (w.shape[-1] == 3 and round_w3) or (w.shape[-1] == 1 and round_w1):
                    w = rb(w)
                y = O.conv_nd(xin, w, b, stride)
                y = y if (key == "4.0" or not round_store) else rb(y)
                run.raw[key] = y
                return y
        for n, f in orig.items():
            setattr(O, n, (lambda f: (lambda *a, **k: rb(f(*a, **k))))(f) if round_blocks else f)
        taps = {}
        run.raw = {}
        acts = []
        orig_act = O.activation

        def rec_act(name, t):
            r = orig_act(name, t)
            acts.append(r)
            return r
        O.activation = rec_act
        o = O.net_forward(St(init, dtype=torch.float64, requires_grad=False), x.double(), cfg, taps=taps)
        O.activation = orig_act
        run.acts = acts
        for n, f in orig.items():
            setattr(O, n, f)
        order = ["enc0", "enc1", "enc2", "dec2", "dec1"]
        return [taps[k] for k in order if k in taps], o
    exact_t, exact_o = run(False, False, False, False, False)
    t, o = run()
    print("raw conv outputs (stored tensor) and BatchNorm statistics, HIP vs emulated, in the oracle's call order:")
    for key, ye in run.raw.items():
        if key in raw_hip:
            yh, mi = raw_hip[key]
            C_ = ye.shape[1]
            mean_e = ye.mean(dim=(0, 2, 3, 4)); var_e = ye.var(dim=(0, 2, 3, 4), unbiased=False)
            inv_e = 1.0 / torch.sqrt(var_e + 1e-5)
            print("  %-28s %-22s r: %.2e   mean: %.2e   invstd: %.2e   (min channel std / |mean|: %.3f)"
                  % (key, tuple(ye.shape[1:]), nrm(yh, ye), nrm(mi[:C_], mean_e), nrm(mi[C_:], inv_e), float((var_e.sqrt() / mean_e.abs().clamp_min(1e-30)).min())))
    # the stride-2 layers' activations: in the oracle the activation call right after each stride-2 conv + BN — find them by shape
    for y in cba_out:
        cands = [a for a in run.acts if tuple(a.shape) == tuple(y.shape)]
        print("stride-2 layer output %s: HIP vs emulated (best match over %d same-shape activations) %.3e" % (tuple(y.shape), len(cands), min(nrm(y, rb(a)) for a in cands)))
    print("HIP block outputs: %d; oracle taps: %d" % (len(hip), len(exact_t)))
    variants = {"emulated (3x3x3 weights + operands, stored tensors)": dict(),
                "... + 1x1x1 weights rounded": dict(round_w1=True),
                "... 3x3x3 weights NOT rounded": dict(round_w3=False),
                "... operands not re-rounded (stored roundings only)": dict(round_x=False),
                "... block / ResPath / up-sampling outputs not rounded": dict(round_blocks=False)}
    for name, kw in variants.items():
        t, o = run(**kw)
        print("%-62s output: HIP vs it %.3e (it vs exact %.3e) | per block HIP vs it: %s" % (name, nrm(out, o), nrm(o, exact_o), " ".join("%.2e" % nrm(h, e) for h, e in zip(hip, t))))
    print("HIP vs exact: output %.3e | per block %s" % (nrm(out, exact_o), " ".join("%.2e" % nrm(h, e) for h, e in zip(hip, exact_t))))


if __name__ == "__main__":
    main()
