#!/usr/bin/env python3
"""Headline benchmark: Adam iterations/sec of the deep-prior loop on the 3-D MultiRes-UNet (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one full iteration of reference main.py:141-213 on one synthetic patch already resident in HBM:
input perturbation -> MulResUnet3D forward -> masked MAE + SNR/PCORR -> backward -> Adam.
Workload (N=1): BASELINE configs[1] geometry — patch (256,128,128), 64-channel noise input, default
MulResUnet3D (5 923 614 parameters), trilinear up-sampling, fp32.  N>1: every rank optimises its own patch
(patches are independent: weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work per voxel and iteration of the default MulResUnet3D (SURVEY §8d / BASELINE.md §3)
FLOP_PER_VOXEL_ITER = 1712.6e9 / (256 * 128 * 128)
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak (nominal, 2.4 GHz)
# calibration on the box: the dominant kernel with staging and LDS reads compiled out (pure
# v_mfma_f32_16x16x4_f32 stream) sustains 107.8 TFLOP/s on this shape = 120.8 TFLOP/s of MFMA issue (clock ~1.85 GHz under load)
FP32_SUSTAINED_TFLOPS = 154.0   # tools/ubench/mfma_rate: pure v_mfma_f32_16x16x4_f32 stream, 32.25 clk/MFMA at 2.39 GHz
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, nargs=3, default=[256, 128, 128])
    ap.add_argument("--upsample", default="linear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-patch", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--mode", default="auto", choices=["auto", "eager", "graph"])
    return ap.parse_args()


def make_interpolator(patch, upsample, device, seed):
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64",
                            "--upsample", upsample, "--loss", "mae", "--lr", "1e-3", "--gain", "40",
                            "--reg_noise_std", "0.03", "--noise_std", "0.1", "--epochs", "3000", "--gpu", "0"])
    vol = u.hyperbolic_volume(tuple(patch), seed=seed)
    mask = u.random_trace_mask(tuple(patch), 0.66, seed=seed + 1)
    u.set_seed(seed)
    T = Interpolator(args, "/tmp", device=device, seed=seed)
    T.load_data({"image": (vol * args.gain)[..., None], "mask": mask[..., None], "name": "0"})
    T.build_model()
    T.build_input()
    return T, args


def cpu_baseline(patch_full, patch_cpu, upsample):
    """The CPU oracle (oracle/dpi_oracle.py: our restatement of the reference, verified against its golden vectors)
    timed on this box's host cores on a bounded sample: 1 warm-up + 2 timed iterations on a `patch_cpu` sub-patch of the
    workload; it/s is scaled to the full patch by the voxel ratio (work per iteration is proportional to V)."""
    from oracle import dpi_oracle as O
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.architectures import get_net
    from deep_prior_interpolation_amd.parameter import parse_arguments
    # more threads than ~32 only add synchronisation overhead to torch's CPU conv at this size (256 threads: 435 s/it)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    a = parse_arguments(["--imgdir", "x", "--datadim", "3d", "--upsample", upsample])
    u.set_seed(0)
    net = get_net(a, 1)
    u.init_weights(net, a.inittype, a.initgain)
    S = O.NetState({k: v.detach().clone() for k, v in net.state_dict().items()})
    cfg = {"ndim": 3, "filters": a.filters, "skip": a.skip, "upsample": a.upsample}
    shp = tuple(patch_cpu)
    vol = torch.from_numpy(u.hyperbolic_volume(shp, seed=0) * 40.0)[None, None]
    mask = torch.from_numpy(u.random_trace_mask(shp, 0.66, seed=1))[None, None]
    z = 0.1 * torch.randn((1, 64) + shp, generator=torch.Generator().manual_seed(0))
    gen = torch.Generator().manual_seed(1)
    t0 = time.time()
    O.optimize(S, cfg, z, vol, mask, 1, generator=gen)                  # warm-up
    n_timed = 2 if time.time() - t0 < 8.0 else 1                        # keep the sample bounded (~10-30 s of CPU work)
    t0 = time.time()
    O.optimize(S, cfg, z, vol, mask, n_timed, generator=gen)
    dt = (time.time() - t0) / n_timed
    scale = float(np.prod(patch_cpu)) / float(np.prod(patch_full))
    return {"value": round(scale / dt, 5), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": "oracle/dpi_oracle.py (torch-CPU fp32 restatement), 1 warm-up + 1-2 timed Adam iterations on a %dx%dx%d "
                      "sub-patch (%.2f s/it), scaled by voxel ratio %.4f to the %dx%dx%d workload patch"
                      % (tuple(patch_cpu) + (dt, scale) + tuple(patch_full))}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    from deep_prior_interpolation_amd import ops
    from deep_prior_interpolation_amd.optim import FusedAdam

    T, args = make_interpolator(a.patch, a.upsample, device, seed=rank)
    V = int(np.prod(a.patch))
    T.optimizer = FusedAdam(T.net.parameters(), lr=args.lr)

    # dominant kernel for the roofline line: the heaviest single launch of the iteration, the full-resolution
    # ResPath 25->16 3x3x3 forward convolution (5.66 GF at 64^3, scales with V)
    def is_dominant(kind, d):
        return kind == "conv_fwd" and d.k == 3 and d.stride == 1 and d.Cin == 25 and d.Cout == 16 and d.D == a.patch[0]
    timer = ops.KernelTimer(is_dominant)

    mode = a.mode
    if mode == "auto":                      # big patches are GPU-bound either way; small ones are launch-bound without a graph
        mode = "eager" if V >= (1 << 20) else "graph"

    ops.set_weight_grad_overlap(mode == "eager" and V >= (1 << 20))

    def eager_step():
        T.optimizer.zero_grad()
        T.optimization_loop()
        T.optimizer.step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    if mode == "graph":
        graph = T.graph_prepare()           # iteration 0 eager + capture of one full iteration
        step = graph.replay
        for _ in range(max(a.warmup - 1, 1)):
            step()
    else:
        step = eager_step
        for _ in range(a.warmup):
            step()
    barrier()
    if mode == "eager":
        ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    timing_src = "HIP events around every launch of this kernel inside the timed region"
    if mode == "graph":
        # launches inside a replayed graph cannot be bracketed from the host: time the same kernel on the same tensors in a
        # short eager tail right after the timed region
        T.graph_finish()
        ops.set_timer(timer)
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize(device)
        ops.set_timer(None)
        timing_src = "HIP events around this kernel in 3 eager iterations run right after the graph-replayed timed region"
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / a.steps * 1e3
        durs = timer.durations()
        dom_ms = float(np.mean(durs)) if durs else None
        dom_flop = 2.0 * 25 * 27 * 16 * V
        iter_flop = FLOP_PER_VOXEL_ITER * V
        roof = None
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_dominant_conv.json")
        if tuple(a.patch) == (256, 128, 128) and os.path.exists(tpath):      # PMC passes cannot run inside this process
            with open(tpath) as fp:
                tj = json.load(fp)
            traffic = tj["traffic_bytes_per_launch"]["total"]
            traffic_src = "profiles/r01_traffic_dominant_conv.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated)"
        if dom_ms:
            ach = dom_flop / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv_mfma_kernel<3,4,2,tail-packed> fwd 25->16 k3 @%dx%dx%d" % tuple(a.patch),
                    "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "algorithmic_bytes": 4.0 * (25 + 16) * V, "launch_ms": round(dom_ms, 4), "launches_timed": len(durs), "launch_timing": timing_src,
                    "note": "fp32 FMA-bound stencil (AI 41-44 FLOP/B > ridge 19.7); peak = nominal fp32 vector = fp32 MFMA rate",
                    "frac_of_sustained_mfma": round(ach / FP32_SUSTAINED_TFLOPS, 4), "sustained_mfma_tflops": FP32_SUSTAINED_TFLOPS,
                    "whole_iteration": {"achieved_tflops": round(iter_flop / (ms * 1e-3) / 1e12, 3),
                                        "frac_fp32": round(iter_flop / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}}
        out = {"metric": "Adam iters/sec on 3D MultiRes-UNet per GPU", "value": round(world * a.steps / dt, 4), "unit": "it/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: MulResUnet3D defaults (5923614 params), patch %dx%dx%d, inputdepth 64, %s, MAE, "
                                      "one independent patch per GPU, loop mode %s" % (tuple(a.patch) + (args.upsample, mode)),
                          "last_loss": T.history.loss[-1], "last_snr_db": T.history.snr[-1]},
               "roofline": roof,
               "cpu_baseline": None if (a.no_cpu_baseline or world > 1) else cpu_baseline(a.patch, a.cpu_patch, a.upsample)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
