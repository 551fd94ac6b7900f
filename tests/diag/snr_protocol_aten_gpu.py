#!/usr/bin/env python3
"""A THIRD implementation at the bench geometry (test infrastructure, round 6): the oracle's restatement of the reference loop (oracle/dpi_oracle.py: plain
torch ops, torch.optim.Adam on every parameter incl. the dead conv biases — i.e. the reference's algorithm as written) run in fp32 ON THE GPU through
aten / MIOpen kernels, same volume / mask / hyper-parameters / seeds as the reference recordings of oracle/make_snr_spread.py --mid and as
tools/snr_protocol_gpu.py.  Neither the HIP path's kernels nor torch's CPU kernels are involved: if this implementation follows the HIP curve, the
+0.5 ... +1.3 dB by which the HIP runs lead the reference's CPU runs at 256x128x128 is not a property of the HIP path; if it follows the reference's CPU
curve, it is.

    python tests/diag/snr_protocol_aten_gpu.py --shape 256 128 128 --epochs 600 --seeds 0 1 2 3 4 5 --out gpurun_out/r06/snr_head_aten_gpu6.json

Initial weights: bit-identical to the reference's for equal seeds (u.set_seed -> get_net -> init_weights); z: torch's CPU generator right after init, i.e. the
reference's own z; per-iteration perturbation: torch's device generator."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import dpi_oracle as O  # noqa: E402


def run_seed(seed, vol, mask, epochs, dtype):
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.architectures import get_net
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear",
                            "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--reg_noise_std", "0.03", "--noise_std", "0.1", "--epochs", str(epochs)])
    dev = torch.device("cuda", 0)
    u.set_seed(seed)
    net = get_net(args, 1)
    u.init_weights(net, args.inittype, args.initgain)
    z = u.get_noise((1, args.inputdepth) + vol.shape, "n").float()
    z *= args.noise_std
    S = O.NetState({k: v.detach().to(dev) for k, v in net.state_dict().items()}, dtype=dtype)
    S.track_running = False
    cfg = {"ndim": 3, "filters": args.filters, "skip": args.skip, "upsample": "trilinear"}
    img = torch.from_numpy((vol.astype(np.float64) * args.gain)[None, None]).to(dev).to(dtype)
    msk = torch.from_numpy(mask.astype(np.float64)[None, None]).to(dev).to(dtype)
    z = z.to(dev).to(dtype)
    opt = torch.optim.Adam(S.params(), lr=args.lr)
    gen = torch.Generator(device=dev).manual_seed(1000 + seed)
    snr, loss_h = [], []
    best, loss_min = None, None
    t0 = time.time()
    for it in range(epochs):
        inp = z + args.reg_noise_std * torch.randn(z.shape, generator=gen, device=dev, dtype=dtype)
        opt.zero_grad(set_to_none=True)
        out = O.net_forward(S, inp, cfg)
        loss = O.masked_loss(out, img, msk, "mae")
        loss.backward()
        l = loss.item()
        loss_h.append(l)
        snr.append(O.snr(out.detach(), img).item())
        if it == 0 or l <= loss_min:
            loss_min, best = l, out.detach().clone()
        opt.step()
        if it in (1, 5, 11):
            torch.cuda.synchronize()
            print("  iteration %d at %.1f s" % (it, time.time() - t0), flush=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    tgt = img.double()
    return {"seed": seed, "snr_out_best": float(10.0 * torch.log10((tgt ** 2).sum() / ((tgt - best.double()) ** 2).sum())), "loss_min": float(loss_min),
            "seconds": dt, "finite": bool(np.isfinite(loss_h).all()), "snr": [round(float(s), 4) for s in snr]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[256, 128, 128])
    ap.add_argument("--seeds", type=int, nargs="*", default=list(range(6)))
    ap.add_argument("--epochs", type=int, default=600)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--no-miopen", action="store_true", help="torch.backends.cudnn.enabled = False: aten's own vol2col + rocBLAS GEMM convolutions instead of MIOpen's "
                                                             "(MIOpen's fp32 3-D kernels: 13 s per iteration at this size incl. their just-in-time builds)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.no_miopen:
        torch.backends.cudnn.enabled = False
    from deep_prior_interpolation_amd import utils as u
    shape = tuple(a.shape)
    vol, mask = u.hyperbolic_volume(shape, seed=0), u.random_trace_mask(shape, 0.66, seed=1)
    runs = []
    for s in a.seeds:
        r = run_seed(s, vol, mask, a.epochs, torch.float32 if a.dtype == "fp32" else torch.float64)
        runs.append(r)
        w = lambda it: float(np.mean(r["snr"][it - 10:it + 1])) if it < len(r["snr"]) else float("nan")
        print("seed %d: %.1f s (%.2f it/s), SNR(out_best) %.2f dB; SNR at 100 / 220 / 300 / 400 / 500 / 599: %.2f %.2f %.2f %.2f %.2f %.2f"
              % (s, r["seconds"], a.epochs / r["seconds"], r["snr_out_best"], w(100), w(220), w(300), w(400), w(500), w(599)), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fp:
            json.dump({"shape": list(shape), "epochs": a.epochs, "precision": a.dtype, "implementation": "oracle/dpi_oracle.py on aten / MIOpen GPU kernels, torch.optim.Adam",
                       "z": "torch_cpu", "noise": "torch device generator", "dead_bias": "stepped (reference's algorithm)", "runs": runs}, fp)


if __name__ == "__main__":
    main()
