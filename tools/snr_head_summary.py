#!/usr/bin/env python3
"""The bench-geometry (256x128x128) SNR comparison from the committed recordings: HIP runs (tools/snr_protocol_gpu.py: profiles/r05/snr_head_hip6*.json —
seeds 0..11, round 5 — and profiles/r06/snr_head_*.json — the round-6 bisect runs: dead-bias variants and runs on the reference's own z, all
statistically the same curve) against the reference seeds in tests/golden/snr_bench_head_256x128x128.npz (oracle/make_snr_spread.py --mid 256 128 128;
`iterations` = how far each seed was recorded).  No GPU, no reference needed.

    python tools/snr_head_summary.py [--bf16] [--r05-only]
"""
import argparse
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true", help="the six bf16-storage seeds of profiles/r05/snr_head_hip6_bf16.json instead")
    ap.add_argument("--r05-only", action="store_true", help="only the twelve default-path seeds of round 5")
    ap.add_argument("--aten", action="store_true", help="the THIRD implementation instead of the HIP path: the oracle's restatement of the reference loop on aten GPU kernels "
                                                        "(tests/diag/snr_protocol_aten_gpu.py, profiles/r06/snr_head_aten_gpu*.json)")
    a = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "snr_bench_head_256x128x128.npz"))
    ref, its = z["snr"].astype(np.float64), np.asarray(z["iterations"]).astype(int)
    files = [os.path.join(ROOT, "profiles", "r05", f) for f in (["snr_head_hip6_bf16.json"] if a.bf16 else ["snr_head_hip6.json", "snr_head_hip6_seeds6to11.json"])]
    if not a.bf16 and not a.r05_only:
        files += [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "snr_head_*.json"))) if "aten" not in os.path.basename(f)]
    if a.aten:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "snr_head_aten_gpu*.json")))
    runs = []
    for f in files:
        with open(f) as fp:
            d = json.load(fp)
        runs += d["runs"]
        print("%-48s %2d runs  (z %s, dead biases %s, noise offset %s)" % (os.path.basename(f), len(d["runs"]), d.get("z", "philox"), d.get("dead_bias", "off"), d.get("noise_offset", 0)))
    n = min(len(r["snr"]) for r in runs)
    mine = np.array([r["snr"][:n] for r in runs])
    print("reference seeds %s recorded to iteration %s; %s runs: %d (%s)" % ([int(s) for s in z["seed"]], [int(k) for k in its], "aten-GPU (third implementation; printed as HIP below)" if a.aten else "HIP", len(mine), "bf16 storage" if a.bf16 else d.get("precision", "fp32")))
    for it in (100, 150, 220, 250, 300, 350, 400, 450, 500, 550, 599):
        cover = [k for k in range(ref.shape[0]) if its[k] > it]
        if len(cover) < 2 or it >= n:
            continue
        x, y = mine[:, it - 10:it + 1].mean(axis=1), ref[cover, it - 10:it + 1].mean(axis=1)
        sx, sy = x.std(ddof=1), y.std(ddof=1)
        se = np.sqrt(sx ** 2 / len(x) + sy ** 2 / len(y))
        # Welch-Satterthwaite degrees of freedom: with 3-5 reference draws the "s.e." count is a t statistic with few degrees of freedom
        dof = se ** 4 / ((sx ** 2 / len(x)) ** 2 / (len(x) - 1) + (sy ** 2 / len(y)) ** 2 / (len(y) - 1))
        print("iteration %3d: HIP %.2f +- %.2f dB (n=%d)  reference %.2f +- %.2f dB (n=%d: %s)  difference %+.2f dB, s.e. %.2f = %.1f s.e. (Welch dof %.1f)"
              % (it, x.mean(), sx, len(x), y.mean(), sy, len(y), " / ".join("%.2f" % v for v in y), x.mean() - y.mean(), se, abs(x.mean() - y.mean()) / se, dof))
    # one number per run: the mean SNR over a 100-iteration span (single checkpoints jitter by +-0.5 dB per run)
    for lo, hi in ((200, 300), (300, 400), (400, 500), (500, 600)):
        cover = [k for k in range(ref.shape[0]) if its[k] >= hi]
        if len(cover) < 2 or hi > n:
            continue
        x, y = mine[:, lo:hi].mean(axis=1), ref[cover, lo:hi].mean(axis=1)
        se = np.sqrt(x.var(ddof=1) / len(x) + y.var(ddof=1) / len(y))
        print("iterations %d-%d (mean over the span): HIP %.2f +- %.2f  reference %.2f +- %.2f (n=%d: %s)  difference %+.2f dB = %.1f s.e."
              % (lo, hi - 1, x.mean(), x.std(ddof=1), y.mean(), y.std(ddof=1), len(y), " / ".join("%.2f" % v for v in y), x.mean() - y.mean(), abs(x.mean() - y.mean()) / se))


if __name__ == "__main__":
    main()
