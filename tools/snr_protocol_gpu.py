#!/usr/bin/env python3
"""HIP side of the SNR protocol on the notebook-like stand-in at any size, next to the reference recordings of
oracle/make_snr_spread.py --mid (tests/golden/snr_mid_128x64x64.npz, snr_bench_head_256x128x128.npz).

    python tools/snr_protocol_gpu.py --shape 128 64 64 --seeds 0 1 2 3 4 5 6 7 8 9 10 11 --out gpurun_out/r05/snr_mid_hip12.json
    python tools/snr_protocol_gpu.py --shape 256 128 128 --epochs 600 --seeds 0 1 2 3 4 5 --out gpurun_out/r05/snr_head_hip6.json
    ... --precision bf16                 (bf16 storage; at these sizes the bf16 kernels are dispatched by default)
    ... --noise torch                    (bisect: the per-iteration perturbation drawn by torch's device generator instead of dpi_noise_add)
    ... --no-overlap                     (bisect: weight gradients on the main stream)
    ... --z torch                        (bisect, round 6: z itself drawn by torch's device generator instead of dpi_fill_normal)
    ... --z torch_cpu [--noise-offset K] (round 6, PAIRED runs: z drawn by torch's CPU generator right after build_model — bit for bit the z the
                                          reference's run of that seed optimises (tests/test_host.py::test_torch_cpu_noise_source_...), with the same
                                          initial weights; only the per-iteration perturbation (Philox, stream seed + 1000 K) differs.  Splits the
                                          seed-to-seed variance into its (weights, z) part and its chaotic part)
    ... --dead-bias sum|noise            (bisect, round 6: the conv biases that feed a BatchNorm Adam-stepped as the reference steps them — on the
                                          rounding residue of the per-channel sum of the pre-BatchNorm gradient (sum), or on N(0, 1e-6^2), far above
                                          Adam's eps: full-size random steps, the upper bound of what such residues can do (noise); ops.DEAD_BIAS)

Writes per-seed SNR / loss histories and SNR(out_best); prints the comparison with the reference recording of that shape
(mean +- s.e. at the checkpoints the tests use).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def run_seed(seed, vol, mask, epochs, precision, noise, overlap, zsrc="philox", dead_bias="off", noise_offset=0):
    from deep_prior_interpolation_amd import ops, utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.optim import FusedAdam
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear",
                            "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--reg_noise_std", "0.03", "--noise_std", "0.1",
                            "--epochs", str(epochs), "--gpu", "0", "--precision", precision])
    u.set_seed(seed)
    T = Interpolator(args, "/tmp", seed=seed)
    T.load_data({"image": (vol.astype(np.float64) * args.gain)[..., None], "mask": mask.astype(np.float64)[..., None], "name": "0"})
    T.build_model()
    T.build_input()
    if zsrc == "torch_cpu":
        zc = u.get_noise(tuple(T.input_.shape), "n").float()          # reference main.py:61-64, generator state = after init_weights
        zc *= args.noise_std
        T.input_ = zc.to(T.device)
        T._z_philox = None
    T.noise_seed = seed + 1000 * noise_offset
    if zsrc == "torch":
        T.input_ = torch.randn(T.input_.shape, generator=torch.Generator(device=T.device).manual_seed(5000 + seed), device=T.device) * args.noise_std
    ops.DEAD_BIAS = dead_bias
    t0 = time.time()
    if noise == "philox" and overlap:
        T.optimize(verbose=False)
    else:
        # the reference's loop (main.py:195-220) spelled out, so that the perturbed input can come from torch's generator
        T.optimizer = FusedAdam(T.net.parameters(), lr=args.lr)
        ops.set_weight_grad_overlap(overlap and T.wants_weight_grad_overlap())
        gen = torch.Generator(device=T.device).manual_seed(1000 + seed)
        for _ in range(epochs):
            T.optimizer.zero_grad()
            inp = None
            if noise == "torch":
                inp = T.input_ + args.reg_noise_std * torch.randn(T.input_.shape, generator=gen, device=T.device)
                if precision == "bf16" and T.storage_bf16_ok():
                    inp = inp.to(torch.bfloat16)
            T.optimization_loop(inp)
            T.optimizer.step()
        torch.cuda.synchronize()
        T.out_best = T._to_numpy_out(T._out_best_dev)
    dt = time.time() - t0
    ops.DEAD_BIAS = "off"
    bias_rms = float(torch.sqrt(torch.mean(torch.cat([p.detach().flatten() for n, p in T.net.named_parameters() if n.endswith(".bias") and p.ndim == 1
                                                        and ".0.bias" in n]) ** 2)).item()) if dead_bias != "off" else None
    target = vol.astype(np.float64) * args.gain
    ob = np.asarray(T.out_best, dtype=np.float64)
    return {"seed": seed, "snr_out_best": float(10.0 * np.log10(np.sum(target ** 2) / np.sum((target - ob) ** 2))),
            "loss_min": float(np.min(T.history.loss)), "seconds": dt, "finite": bool(np.isfinite(T.history.loss).all()),
            "snr": [round(float(s), 4) for s in T.history.snr], "loss": [float(l) for l in T.history.loss], "bias_rms": bias_rms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[128, 64, 64])
    ap.add_argument("--seeds", type=int, nargs="*", default=list(range(6)))
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--noise", default="philox", choices=["philox", "torch"])
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--z", default="philox", choices=["philox", "torch", "torch_cpu"])
    ap.add_argument("--noise-offset", type=int, default=0)
    ap.add_argument("--dead-bias", default="off", choices=["off", "sum", "noise"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from deep_prior_interpolation_amd import utils as u
    shape = tuple(a.shape)
    tag = "x".join(str(n) for n in shape)
    ref = None
    for name in ("snr_mid_%s.npz" % tag, "snr_bench_head_%s.npz" % tag):
        p = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(p):
            ref = np.load(p)
    epochs = a.epochs or (int(ref["snr"].shape[1]) if ref is not None else 1200)
    vol, mask = u.hyperbolic_volume(shape, seed=0), u.random_trace_mask(shape, 0.66, seed=1)
    runs = []
    for s in a.seeds:
        r = run_seed(s, vol, mask, epochs, a.precision, a.noise, not a.no_overlap, a.z, a.dead_bias, a.noise_offset)
        runs.append(r)
        print("seed %d: SNR(out_best) %.2f dB, min loss %.4f, %.1f s%s" % (s, r["snr_out_best"], r["loss_min"], r["seconds"],
                                                                            "" if r["bias_rms"] is None else ", rms of the conv biases %.3e" % r["bias_rms"]), flush=True)
    mine = np.array([r["snr"] for r in runs])
    lines = []
    if ref is not None:
        rs = ref["snr"].astype(np.float64)
        its = np.asarray(ref["iterations"]).astype(int) if "iterations" in ref else np.full(rs.shape[0], rs.shape[1])
        n_it = min(int(its.max()), mine.shape[1])
        for it in [i for i in (100, 220, 300, 400, 500, 599, 800, 1199) if i < n_it]:
            cover = [k for k in range(rs.shape[0]) if its[k] > it]
            x, y = mine[:, it - 10:it + 1].mean(axis=1), rs[cover, it - 10:it + 1].mean(axis=1)
            sx = x.std(ddof=1) if len(x) > 1 else float("nan")
            sy = y.std(ddof=1) if len(y) > 1 else float("nan")
            se = np.sqrt(sx ** 2 / len(x) + (sy if len(y) > 1 else sx) ** 2 / len(y))
            lines.append("iteration %4d: HIP %.2f +- %.2f dB (n=%d)   reference %.2f +- %.2f dB (n=%d)   difference %+.2f dB, s.e. %.2f%s"
                         % (it, x.mean(), sx, len(x), y.mean(), sy, len(y), x.mean() - y.mean(), se,
                            "" if len(y) > 1 else " (reference spread taken as HIP's)"))
        if bool(np.all(ref["done"] == 1)) and mine.shape[1] >= rs.shape[1]:
            hb, rb = np.array([r["snr_out_best"] for r in runs]), ref["snr_out_best"].astype(np.float64)
            se = np.sqrt(hb.var(ddof=1) / len(hb) + rb.var(ddof=1) / len(rb))
            lines.append("SNR(out_best): HIP %.2f +- %.2f dB (n=%d)   reference %.2f +- %.2f dB (n=%d)   difference %+.2f dB, 2 s.e. %.2f"
                         % (hb.mean(), hb.std(ddof=1), len(hb), rb.mean(), rb.std(ddof=1), len(rb), hb.mean() - rb.mean(), 2 * se))
    for ln in lines:
        print(ln)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fp:
        json.dump({"shape": list(shape), "epochs": epochs, "precision": a.precision, "noise": a.noise, "overlap": not a.no_overlap, "z": a.z, "dead_bias": a.dead_bias, "noise_offset": a.noise_offset,
                   "summary": lines, "runs": [{k: v for k, v in r.items() if k != "loss"} for r in runs]}, fp)


if __name__ == "__main__":
    main()
