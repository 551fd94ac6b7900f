"""Record the reference's own run-to-run SNR spread — build container only.

    cd /tmp && python /root/repo/oracle/make_snr_spread.py --seeds 0 1 2 --threads 2      (one process per seed group)
    cd /tmp && python /root/repo/oracle/make_snr_spread.py --merge                        (-> tests/golden/snr_spread.npz)

Drives the reference `Interpolator` (imported from /root/reference through oracle/ref_shim.py, exactly as
proof_of_concept_3D.ipynb cell 15 does) on the (48,32,32) hyperbolic stand-in: default MulResUnet3D (5 923 614
parameters), 66 % random missing traces, gain 40, MAE, trilinear, param_noise=False, 1000 Adam iterations, one run per
seed (`u.set_seed(seed)` before build_model: weights, z and the per-iteration noise all follow from it).
Recorded per seed: loss / SNR / PCORR history, SNR(out_best), min and final loss.  The committed .npz holds data only
(the volume, the mask and those numbers); initial weights are NOT stored — `init_weights` under the same seed is
bit-identical between the reference and the build (tests/test_host.py::test_same_seed_init_is_bit_identical).
"""
import argparse
import glob
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
PART = os.path.join(OUT, "_snr_spread_parts")
SHAPE = (48, 32, 32)
ARGV = ["--imgdir", "/nonexistent", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear",
        "--loss", "mae", "--lr", "1e-3", "--gain", "40", "--reg_noise_std", "0.03", "--noise_std", "0.1"]


def stand_in():
    """The volume/mask every seed shares (ours: deep_prior_interpolation_amd.utils.synthetic)."""
    from deep_prior_interpolation_amd.utils.synthetic import hyperbolic_volume, random_trace_mask
    vol = hyperbolic_volume(SHAPE, seed=0).astype(np.float64)
    mask = random_trace_mask(SHAPE, 0.66, seed=1).astype(np.float64)
    return vol, mask


def snr_db(out, target):
    return 10.0 * np.log10(np.sum(target ** 2) / np.sum((target - out) ** 2))


def run_seed(seed, epochs, threads):
    torch.set_num_threads(threads)
    main = ref_shim.load_main()
    import utils as u  # reference module
    args = ref_shim.parse_args(ARGV + ["--epochs", str(epochs)])
    args.param_noise = False
    vol, mask = stand_in()
    image = (vol * args.gain)[..., None]
    u.set_seed(seed)
    T = main.Interpolator(args, tempfile.mkdtemp())
    t0 = time.time()
    with redirect_stdout(io.StringIO()):
        std = T.load_data({"image": image, "mask": mask[..., None], "name": "0"})
        T.build_model()
        T.build_input()
        T.optimize()
    dt = time.time() - t0
    out_best = np.asarray(T.out_best, dtype=np.float64)
    d = {"seed": np.int64(seed), "threads": np.int64(threads), "epochs": np.int64(epochs), "std": np.float64(std),
         "loss": np.array(T.history.loss), "snr": np.array(T.history.snr), "pcorr": np.array(T.history.pcorr),
         "snr_out_best": np.float64(snr_db(out_best, image[..., 0])), "loss_min": np.float64(T.loss_min),
         "argmin": np.int64(int(np.argmin(T.history.loss))), "seconds": np.float64(dt)}
    os.makedirs(PART, exist_ok=True)
    np.savez_compressed(os.path.join(PART, "seed%03d.npz" % seed), **d)
    print("seed %d: %.0f s, SNR(out_best) %.2f dB, min loss %.3e, final loss %.3e" %
          (seed, dt, d["snr_out_best"], d["loss_min"], d["loss"][-1]), flush=True)


def merge():
    files = sorted(glob.glob(os.path.join(PART, "seed*.npz")))
    parts = [dict(np.load(f)) for f in files]
    vol, mask = stand_in()
    out = {"volume": vol.astype(np.float32), "mask": mask.astype(np.uint8), "shape": np.array(SHAPE),
           "argv": np.array(" ".join(ARGV)), "torch": np.array(torch.__version__)}
    for k in ("seed", "threads", "epochs", "std", "snr_out_best", "loss_min", "argmin", "seconds"):
        out[k] = np.array([p[k] for p in parts])
    for k in ("loss", "snr", "pcorr"):
        out[k] = np.stack([p[k] for p in parts]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "snr_spread.npz"), **out)
    s = out["snr_out_best"]
    print("merged %d seeds: SNR(out_best) mean %.2f dB, std %.2f, s.e. %.2f" % (len(s), s.mean(), s.std(ddof=1),
                                                                               s.std(ddof=1) / np.sqrt(len(s))))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="*", default=[])
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--merge", action="store_true")
    a = ap.parse_args()
    ref_shim.install()
    for s in a.seeds:
        run_seed(s, a.epochs, a.threads)
    if a.merge:
        merge()
