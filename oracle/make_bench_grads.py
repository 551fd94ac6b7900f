"""Record the reference's OWN gradients of iteration 0 at the bench geometry (256x128x128) — build container only.

    cd /tmp && python /root/repo/oracle/make_bench_grads.py [--seed 0] [--threads 2]        (-> tests/golden/bench_grads_256x128x128_seed<k>.npz)

Drives the reference's Interpolator (imported from /root/reference through oracle/ref_shim.py) exactly as oracle/make_snr_spread.py --mid 256 128 128
does — same stand-in, mask, hyper-parameters, `u.set_seed(seed)` before build_model — and stops after the backward of iteration 0
(`optimizer.zero_grad(); optimization_loop()` as main.py:209-212 runs them; no optimiser step).  5.9 M gradient values are 24 MB, so the fixture keeps, per
parameter tensor: its norm, its sum, its first 64 values, and its dot product with a fixed +-1 vector (numpy RandomState(1234 + index of the tensor)) —
a checksum no systematic error of a kernel survives.  Data only; `tests/test_gpu_bench_size.py` compares the HIP path's gradients with it, fed with the
reference's own input stream (--noise_source torch_cpu)."""
import argparse
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
from oracle.make_snr_spread import ARGV, stand_in  # noqa: E402

SHAPE = (256, 128, 128)


def sign_vector(k, n):
    return np.random.RandomState(1234 + k).randint(0, 2, size=n).astype(np.float64) * 2.0 - 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--fp64-oracle", action="store_true", help="instead of the reference's own fp32 backward: OUR oracle (oracle/dpi_oracle.py) in float64 on the same "
                                                              "weights and the same perturbed input -> ..._fp64.npz: the yardstick that says whose fp32 gradients are off")
    ap.add_argument("--tag", default="", help="suffix of the output file (a second recording with another thread count = the reference's own reproducibility)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ref_shim.install()
    main_mod = ref_shim.load_main()
    import utils as u  # reference module
    args = ref_shim.parse_args(ARGV + ["--epochs", "3000"])
    args.param_noise = False
    vol, mask = stand_in(SHAPE, dense=True)
    image = (vol * args.gain)[..., None]
    u.set_seed(a.seed)
    T = main_mod.Interpolator(args, tempfile.mkdtemp())
    t0 = time.time()
    with redirect_stdout(io.StringIO()):
        T.load_data({"image": image, "mask": mask[..., None], "name": "0"})
        T.build_model()
        T.build_input()
        if not a.fp64_oracle:
            T.optimizer = torch.optim.Adam(T.net.parameters(), lr=args.lr)
            T.optimizer.zero_grad()
            T.optimization_loop()
    if a.fp64_oracle:
        from oracle import dpi_oracle as O
        inp = T.input_.detach().clone()                       # main.py:148-150, the draw optimization_loop() would make next
        inp += args.reg_noise_std * inp.clone().normal_()
        S = O.NetState({k: v.detach().clone() for k, v in T.net.state_dict().items()}, dtype=torch.float64)
        S.track_running = False
        cfg = {"ndim": 3, "filters": args.filters, "skip": args.skip, "upsample": "trilinear"}
        out_ = O.net_forward(S, inp.double(), cfg)
        loss = O.masked_loss(out_, T.img_.double(), T.mask_.double(), "mae")
        loss.backward()
        T.history.loss.append(loss.item())
        T.history.snr.append(O.snr(out_.detach(), T.img_.double()).item())

        class _P:                                             # the same (name, gradient) walk as below
            def __init__(self, t):
                self.grad, self.ndim = t.grad, t.ndim
        named = [(n, _P(S.P[n])) for n, _ in T.net.named_parameters()]
        a.tag = a.tag or "_fp64"
    else:
        named = list(T.net.named_parameters())
    out = {"shape": np.array(SHAPE), "seed": np.int64(a.seed), "threads": np.int64(a.threads), "torch": np.array(torch.__version__),
           "loss0": np.float64(T.history.loss[0]), "snr0": np.float64(T.history.snr[0]), "seconds": np.float64(time.time() - t0)}
    names, norms, sums, dots, heads, numels, ndims = [], [], [], [], [], [], []
    for k, (n, p) in enumerate(named):
        g = (p.grad if p.grad is not None else torch.zeros(1)).detach().double().flatten().numpy()
        names.append(n)
        norms.append(np.linalg.norm(g))
        sums.append(g.sum())
        dots.append(float(np.dot(g, sign_vector(k, g.size))))
        h = np.zeros(64)
        h[:min(64, g.size)] = g[:64]
        heads.append(h)
        numels.append(g.size)
        ndims.append(p.ndim)
    out.update(names=np.array(names), norm=np.array(norms), sum=np.array(sums), dot=np.array(dots), head=np.stack(heads), numel=np.array(numels), ndim=np.array(ndims))
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "bench_grads_256x128x128_seed%d%s.npz" % (a.seed, a.tag))
    np.savez_compressed(path, **out)
    print("seed %d: loss[0] %.7f, %d tensors, %.0f s -> %s" % (a.seed, out["loss0"], len(names), out["seconds"], path))


if __name__ == "__main__":
    main()
