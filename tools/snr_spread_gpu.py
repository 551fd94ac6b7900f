#!/usr/bin/env python3
"""HIP-path side of the SURVEY 8(d)(iv) SNR-parity protocol: the (48,32,32) stand-in of tests/golden/snr_spread.npz (or, before
that fixture exists, the same generator), default MulResUnet3D, 1000 Adam iterations, one run per seed.

    python tools/snr_spread_gpu.py --seeds 0 1 2 3 4 5 6 7 --out gpurun_out/snr_spread_gpu.json
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


def run_seed(seed, vol, mask, epochs=1000, mode="auto"):
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear",
                            "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--reg_noise_std", "0.03", "--noise_std", "0.1",
                            "--epochs", str(epochs), "--gpu", "0"])
    u.set_seed(seed)
    T = Interpolator(args, "/tmp", seed=seed)
    T.load_data({"image": (vol.astype(np.float64) * args.gain)[..., None], "mask": mask.astype(np.float64)[..., None], "name": "0"})
    T.build_model()
    T.build_input()
    t0 = time.time()
    T.optimize(verbose=False, mode=mode)
    dt = time.time() - t0
    target = vol.astype(np.float64) * args.gain
    ob = np.asarray(T.out_best, dtype=np.float64)
    return {"seed": seed, "snr_out_best": float(10.0 * np.log10(np.sum(target ** 2) / np.sum((target - ob) ** 2))),
            "loss_min": float(np.min(T.history.loss)), "loss_final": float(T.history.loss[-1]), "seconds": dt,
            "snr": [float(s) for s in T.history.snr[::50]], "loss0": float(T.history.loss[0])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="*", default=list(range(8)))
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "snr_spread_gpu.json"))
    a = ap.parse_args()
    from deep_prior_interpolation_amd import utils as u
    gp = os.path.join(ROOT, "tests", "golden", "snr_spread.npz")
    if os.path.exists(gp):
        z = np.load(gp)
        vol, mask = z["volume"].astype(np.float32), z["mask"].astype(np.float32)
    else:
        vol, mask = u.hyperbolic_volume((48, 32, 32), seed=0), u.random_trace_mask((48, 32, 32), 0.66, seed=1)
    runs = []
    for s in a.seeds:
        r = run_seed(s, vol, mask, a.epochs, a.mode)
        runs.append(r)
        print("seed %d: SNR(out_best) %.2f dB, min loss %.3e, final %.3e, %.1f s" % (s, r["snr_out_best"], r["loss_min"], r["loss_final"], r["seconds"]),
              flush=True)
    v = np.array([r["snr_out_best"] for r in runs])
    summary = {"n": len(v), "mean": float(v.mean()), "std": float(v.std(ddof=1)) if len(v) > 1 else None,
               "se": float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else None, "runs": runs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fp:
        json.dump(summary, fp)
    print("HIP path: SNR(out_best) mean %.2f dB, std %s, s.e. %s over %d seeds" % (summary["mean"], summary["std"], summary["se"], len(v)))


if __name__ == "__main__":
    main()
