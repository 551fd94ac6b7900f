#!/usr/bin/env python3
"""One complete deep-prior optimisation on the GPU at the bench geometry (BASELINE configs[1]: patch 256x128x128, default
MulResUnet3D, 3000 Adam iterations, MAE, trilinear, gain 40; reference main.py:195-220, proof_of_concept_3D.ipynb:354-358)
on the synthetic hyperbolic stand-in.  Writes the loss / SNR / PCORR trajectory and SNR(out_best) as JSON.

    python tools/full_run.py --out gpurun_out/full_run.json [--patch 256 128 128] [--epochs 3000] [--seed 0]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch", type=int, nargs=3, default=[256, 128, 128])
    ap.add_argument("--epochs", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--missing", type=float, default=0.66)
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--sparse-events", type=int, default=0, help="> 0: the round-1/2 stand-in (`sparse_hyperbolic_volume`) with this many "
                    "narrow events instead of the notebook-like cube (`hyperbolic_volume`, utils/synthetic.py)")
    ap.add_argument("--background", type=float, default=0.02, help="weak band-limited background of the stand-in (0 = exact zeros above the first event)")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_run.json"))
    a = ap.parse_args()
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear",
                            "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--reg_noise_std", "0.03", "--noise_std", "0.1",
                            "--epochs", str(a.epochs), "--gpu", "0", "--precision", a.precision])
    shape = tuple(a.patch)
    if a.sparse_events > 0:
        vol = u.sparse_hyperbolic_volume(shape, seed=0, nev=a.sparse_events)
    else:
        vol = u.hyperbolic_volume(shape, seed=0, background=a.background)
    mask = u.random_trace_mask(shape, a.missing, seed=1)
    u.set_seed(a.seed)
    T = Interpolator(args, "/tmp", seed=a.seed)
    std = T.load_data({"image": (vol * args.gain)[..., None].astype(np.float64), "mask": mask[..., None].astype(np.float64), "name": "0"})
    T.build_model()
    T.build_input()
    torch.cuda.synchronize()
    t0 = time.time()
    T.optimize(verbose=False, mode=a.mode)
    torch.cuda.synchronize()
    dt = time.time() - t0
    target = vol.astype(np.float64) * args.gain
    ob = np.asarray(T.out_best, dtype=np.float64)
    snr_best = 10.0 * np.log10(np.sum(target ** 2) / np.sum((target - ob) ** 2))
    h = T.history
    every = max(1, a.epochs // 300)
    res = {"patch": list(shape), "stand_in": ("sparse, %d events" % a.sparse_events) if a.sparse_events > 0 else "notebook-like, background %g" % a.background,
           "precision": a.precision, "epochs": len(h.loss), "seed": a.seed, "missing_traces": a.missing, "std_masked": std,
           "seconds": round(dt, 2), "it_per_s": round(len(h.loss) / dt, 3), "num_params": T.num_params,
           "finite": bool(np.isfinite(h.loss).all() and np.isfinite(ob).all()),
           "snr_out_best_db": float(snr_best), "loss_min": float(np.min(h.loss)), "argmin": int(np.argmin(h.loss)),
           "final": {"loss": h.loss[-1], "snr_db": h.snr[-1], "pcorr": h.pcorr[-1]},
           "snr_last50_mean": float(np.mean(h.snr[-50:])), "snr_last50_std": float(np.std(h.snr[-50:])),
           "trajectory_every": every, "loss": [float(x) for x in h.loss[::every]], "snr_db": [float(x) for x in h.snr[::every]],
           "pcorr": [float(x) for x in h.pcorr[::every]]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fp:
        json.dump(res, fp)
    print(json.dumps({k: v for k, v in res.items() if k not in ("loss", "snr_db", "pcorr")}))


if __name__ == "__main__":
    main()
