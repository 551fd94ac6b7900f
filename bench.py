#!/usr/bin/env python3
"""Headline benchmark: Adam iterations/sec of the deep-prior loop on the 3-D MultiRes-UNet (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload c2 (default; BASELINE configs[1] geometry): a step = one full iteration of reference main.py:141-213 on one synthetic
patch already resident in HBM: input perturbation -> MulResUnet3D forward -> masked MAE + SNR/PCORR -> backward -> Adam.
Patch (256,128,128), 64-channel noise input, default MulResUnet3D (5 923 614 parameters), trilinear up-sampling, fp32.
N>1: every rank optimises its own patch (patches are independent: weak scaling, no data-path collective).

Workload c3 (BASELINE configs[2]): the patch-parallel job itself — a queue of 64^3 patches (stride 32) of a synthetic 256^3
volume with 50 % missing traces, pulled by all ranks from the shared counter (parallel.PatchQueue), `--concurrent` patches at a
time per GPU as replayed hipGraphs, K Adam iterations per patch, timed END TO END: per-patch set-up (weights, z, graph capture),
the K iterations, dpi_overlap_add, the single all-reduce of the accumulator volume and the normalisation.  `--patches P` bounds
the queue to the first P*N patches (the full 343 x 3000 iterations take hours); value = patch-iterations/s over all ranks.

Workload c4 (BASELINE configs[3] data): the shipped 2-D section datasets/lines (170 x 100; the copy recorded in
tests/golden/host.npz — /root/reference does not exist on the GPU box) with its random66 mask, default 2-D MulResUnet
(2 186 704 parameters), the geometry of proof_of_concept_2D.ipynb (3000 iterations in 142 s = 21 it/s on a V100); `--aa-weight W`
adds the anti-aliasing regulariser (dips + Hale2D adjoint every iteration).

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
BMIN_PER_VOXEL_ITER = (39.18e9 - 0.166e9) / (256 * 128 * 128)      # compulsory HBM bytes: 9.3 kB / voxel / iteration ...
BMIN_CONST = 0.166e9                                               # ... + 28 B x 5 923 614 parameters (Adam)
FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak (nominal, 2.4 GHz)
FP32_SUSTAINED_TFLOPS = 154.0     # tools/ubench/mfma_rate: pure v_mfma_f32_16x16x4_f32 stream, 32.25 clk/MFMA at 2.39 GHz
HBM_PEAK_GBS = 8000.0
PROFILE_JSON = os.path.join(ROOT, "profiles", "r02_traffic.json")   # rocprofv3 --pmc results (cannot be collected in-process)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"])
    ap.add_argument("--aa-weight", type=float, default=0.0, help="c4: weight of the anti-aliasing (directional Laplacian) regulariser")
    ap.add_argument("--patch", type=int, nargs=3, default=None)
    ap.add_argument("--upsample", default="linear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="c2: skip the short bf16 / split passes reported under other_modes")
    ap.add_argument("--cpu-patch", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--mode", default="auto", choices=["auto", "eager", "graph"])
    ap.add_argument("--patches", type=int, default=12, help="c3: patches per rank taken from the queue (a multiple of --concurrent avoids a part-filled last round)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "split"],
                    help="bf16: BASELINE configs[4] mixed precision (bf16 MFMA operands in the 3x3x3 convs, fp32 accumulate / storage / Adam); "
                         "a SECOND bench line, the headline stays fp32")
    ap.add_argument("--concurrent", type=int, default=6, help="c3: patches optimised side by side on one GPU")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = {"c2": 10, "c3": 100, "c4": 300}[a.workload]
    if a.patch is None:
        a.patch = [256, 128, 128] if a.workload == "c2" else [64, 64, 64]
    return a


PRECISION = "fp32"


def default_args(upsample, epochs=3000):
    from deep_prior_interpolation_amd.parameter import parse_arguments
    return parse_arguments(["--precision", PRECISION, "--imgdir", "synthetic", "--datadim", "3d", "--net", "multiunet", "--inputdepth", "64",
                            "--upsample", upsample, "--loss", "mae", "--lr", "1e-3", "--gain", "40",
                            "--reg_noise_std", "0.03", "--noise_std", "0.1", "--epochs", str(epochs), "--gpu", "0"])


def make_interpolator(patch, upsample, device, seed):
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    args = default_args(upsample)
    vol = u.hyperbolic_volume(tuple(patch), seed=seed)
    mask = u.random_trace_mask(tuple(patch), 0.66, seed=seed + 1)
    u.set_seed(seed)
    T = Interpolator(args, "/tmp", device=device, seed=seed)
    T.load_data({"image": (vol * args.gain)[..., None], "mask": mask[..., None], "name": "0"})
    T.build_model()
    T.build_input()
    return T, args


def gpu_small_patch_rate(patch, upsample, device, iters=60):
    """it/s of ONE patch of the CPU-baseline sample size on the GPU (hipGraph replay), for the side-by-side CPU / GPU figure."""
    T, _ = make_interpolator(patch, upsample, device, seed=0)
    g = T.graph_prepare()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    T.graph_finish()
    return iters / dt


def cpu_baseline(patch_full, patch_cpu, upsample, iters, gpu_rate_same_patch):
    """The CPU oracle (oracle/dpi_oracle.py: our restatement of the reference, verified against its golden vectors) timed on
    this box's host cores on a bounded sample of the workload: ONE `patch_cpu` patch (the configs[2] patch size), 1 warm-up +
    `iters` timed Adam iterations.  `value` is the MEASURED rate on that patch; the GPU rate on the same patch
    is reported beside it (no extrapolation to the larger bench patch)."""
    from oracle import dpi_oracle as O
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.architectures import get_net
    from deep_prior_interpolation_amd.parameter import parse_arguments
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)     # torch's CPU conv does not scale past ~32 threads at this size (measured: 256 threads 435 s/it vs 1.2 s/it)
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
    warm = time.time() - t0
    n_timed = iters if warm < 6.0 else max(1, int(30.0 / warm))         # keep the sample bounded (~10-30 s of CPU work)
    t0 = time.time()
    O.optimize(S, cfg, z, vol, mask, n_timed, generator=gen)
    dt = (time.time() - t0) / n_timed
    scale = float(np.prod(patch_cpu)) / float(np.prod(patch_full))
    return {"value": round(1.0 / dt, 5), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": "oracle/dpi_oracle.py (torch-CPU fp32 restatement of the reference loop), ONE %dx%dx%d patch (default MulResUnet3D, "
                      "64-ch input): 1 warm-up + %d timed Adam iterations, %.2f s/it measured, %d of %d host cores (torch CPU conv does not "
                      "scale past ~32 threads here)" % (tuple(patch_cpu) + (n_timed, dt, cores, ncpu)),
            "gpu_same_sample": {"value": round(gpu_rate_same_patch, 2), "unit": "it/s",
                                "note": "the HIP path on the same single %dx%dx%d patch (one hipGraph replay stream)" % tuple(patch_cpu)},
            "scaled_to_workload_patch": {"value": round(scale / dt, 5), "voxel_ratio": scale,
                                         "note": "work per iteration is proportional to voxels; informational only"}}


def load_profile_json():
    if os.path.exists(PROFILE_JSON):
        with open(PROFILE_JSON) as fp:
            return json.load(fp)
    return None


def run_c2(a, rank, world, device):
    from deep_prior_interpolation_amd import ops
    from deep_prior_interpolation_amd.optim import FusedAdam
    T, args = make_interpolator(a.patch, a.upsample, device, seed=rank)
    V = int(np.prod(a.patch))
    T.optimizer = FusedAdam(T.net.parameters(), lr=args.lr)
    ops.set_precision(a.precision)

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
    timing_src = "HIP events (torch.cuda.Event on the launch stream) around every launch of this kernel inside the timed region"
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
    if rank != 0:
        return None
    ms = dt / a.steps * 1e3
    durs = timer.durations()
    dom_ms = float(np.mean(durs)) if durs else None
    dom_flop = 2.0 * 25 * 27 * 16 * V
    iter_flop = FLOP_PER_VOXEL_ITER * V
    bmin = BMIN_PER_VOXEL_ITER * V + BMIN_CONST
    prof = load_profile_json() if tuple(a.patch) == (256, 128, 128) else None
    roof = None
    if dom_ms:
        ach = dom_flop / (dom_ms * 1e-3) / 1e12
        whole = {"achieved_tflops": round(iter_flop / (ms * 1e-3) / 1e12, 3),
                 "frac_fp32": round(iter_flop / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                 "algorithmic_bytes_per_iteration": bmin,
                 "frac_hbm": round(bmin / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "binding_roofline_ms": round(iter_flop / (FP32_PEAK_TFLOPS * 1e12) * 1e3, 3)}
        traffic = traffic_src = None
        if prof:
            dk = prof.get("dominant_conv", {})
            traffic = dk.get("traffic_bytes_per_launch", {}).get("total")
            traffic_src = prof.get("source")
            wi = prof.get("whole_iteration", {})
            if wi.get("hbm_bytes_per_iteration"):
                whole["measured_hbm_bytes_per_iteration"] = wi["hbm_bytes_per_iteration"]
                whole["measured_over_algorithmic"] = round(wi["hbm_bytes_per_iteration"] / bmin, 3)
                whole["measured_hbm_gbs"] = round(wi["hbm_bytes_per_iteration"] / (ms * 1e-3) / 1e9, 1)
        if a.precision == "fp32":
            roof = {"bound": "mfma", "kernel": "conv_mfma_kernel<3,4,2,tail-packed> fwd 25->16 k3 @%dx%dx%d" % tuple(a.patch),
                    "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "algorithmic_bytes": 4.0 * (25 + 16) * V, "launch_ms": round(dom_ms, 4), "launches_timed": len(durs), "launch_timing": timing_src,
                    "note": "fp32 FMA-bound stencil (AI 41-44 FLOP/B > ridge 19.7); peak = nominal fp32 vector = fp32 MFMA rate",
                    "frac_of_sustained_mfma": round(ach / FP32_SUSTAINED_TFLOPS, 4), "sustained_mfma_tflops": FP32_SUSTAINED_TFLOPS,
                    "whole_iteration": whole}
        else:
            # bf16 MFMA runs at 16x the fp32 matrix rate (2.5 PFLOP/s dense): the same launch is bound by HBM (AI 132 FLOP/B < ridge 312)
            alg_bytes = 4.0 * (25 + 16) * V
            gbs = alg_bytes / (dom_ms * 1e-3) / 1e9
            whole.pop("measured_hbm_bytes_per_iteration", None); whole.pop("measured_over_algorithmic", None); whole.pop("measured_hbm_gbs", None)
            roof = {"bound": "hbm", "kernel": "conv_bf16_kernel fwd 25->16 k3 @%dx%dx%d (%s, fp32 accumulate)"
                              % (tuple(a.patch) + ("bf16 operands" if a.precision == "bf16" else "three-term bf16 split, 6 products",)),
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes": alg_bytes, "launch_ms": round(dom_ms, 4), "launches_timed": len(durs), "launch_timing": timing_src,
                    "achieved_tflops": round(ach, 1),
                    "note": "fp32 tensors in HBM (the mode rounds operands on the way into LDS); traffic not measured for this mode",
                    "whole_iteration": whole}
    # the other arithmetic modes on the same patch, right after the timed region (information only: `value` above is the mode asked for)
    other = None
    if a.precision == "fp32" and world == 1 and mode == "eager" and not a.no_other_modes:
        other = {}
        for prec, label in (("bf16", "bf16"), ("split", "f32 (3 x bf16 split)")):
            ops.set_precision(prec)
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                eager_step()
            torch.cuda.synchronize(device)
            dtp = time.perf_counter() - t1
            other[prec] = {"dtype": label, "ms_per_step": round(dtp / a.steps * 1e3, 3), "value": round(a.steps / dtp, 4), "unit": "it/s",
                           "steps": a.steps, "note": "python bench.py --precision %s reports this mode as its own line" % prec}
        ops.set_precision("fp32")
    cpu = None
    if not (a.no_cpu_baseline or world > 1):
        cpu = cpu_baseline(a.patch, a.cpu_patch, a.upsample, a.cpu_iters, gpu_small_patch_rate(a.cpu_patch, a.upsample, device))
    return {"metric": "Adam iters/sec on 3D MultiRes-UNet per GPU", "value": round(world * a.steps / dt, 4), "unit": "it/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "split": "f32 (3 x bf16 split)"}[a.precision], "data": "synthetic",
            "config": {"workload": {"fp32": "", "bf16": "MIXED PRECISION (bf16 MFMA operands in the 3x3x3 convolutions, fp32 accumulate, fp32 tensors / master "
                                                          "weights / BatchNorm / Adam) — ",
                                   "split": "SPLIT MODE (forward / backward-data operands split exactly into three bf16 terms, six partial products "
                                            "accumulated in fp32: fp32-class accuracy on the bf16 matrix cores) — "}[a.precision] + "configs[1]: MulResUnet3D defaults (5923614 params), patch %dx%dx%d, inputdepth 64, %s, MAE, "
                                   "one independent patch per GPU, loop mode %s" % (tuple(a.patch) + (args.upsample, mode)),
                       "last_loss": T.history.loss[-1], "last_snr_db": T.history.snr[-1]},
            "roofline": roof, "cpu_baseline": cpu, "other_modes": other}


def run_c3(a, rank, world, device):
    """configs[2]: the sharded patch queue, end to end."""
    from deep_prior_interpolation_amd import parallel, utils as u
    from deep_prior_interpolation_amd.data import patch_extractor_for
    args = default_args(a.upsample, epochs=a.steps)
    vshape = (256, 256, 256)
    args.patch_shape, args.patch_stride = list(a.patch), [p // 2 for p in a.patch]
    vol = u.hyperbolic_volume(vshape, seed=0)
    mask = u.random_trace_mask(vshape, 0.5, seed=1)
    pe = patch_extractor_for(vshape, args.patch_shape, args.patch_stride, "3d")
    origins = u.window_origins(vshape, pe.dim, pe.stride)
    n_total = len(origins)
    n_run = min(n_total, a.patches * world)
    # the queue's first n_run patches, spread over the volume (every 343 // n_run-th window) so masks / content differ
    pick = [int(i) for i in np.linspace(0, n_total - 1, n_run).round()]
    patches = []
    for i in pick:
        sl = tuple(slice(int(o), int(o) + d) for o, d in zip(origins[i], pe.dim))
        patches.append({"image": (vol[sl] * args.gain)[..., None].astype(np.float64), "mask": mask[sl][..., None].astype(np.float64),
                        "name": str(i).zfill(3)})
    sel_origins = [origins[i] for i in pick]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # warm-up: one patch for a few iterations (lazy caches, allocator), outside the timed region
    wargs = default_args(a.upsample, epochs=max(a.warmup, 3))
    wargs.patch_shape, wargs.patch_stride = args.patch_shape, args.patch_stride
    parallel.optimise_volume(wargs, patches[:1], sel_origins[:1], vshape, pe, device, "/tmp", 1, parallel.PatchQueue(1), save=False)
    barrier()
    timings = {}
    queue = parallel.PatchQueue.for_process_group(n_run, key="dpi/bench_c3")
    t0 = time.perf_counter()
    rec, mine = parallel.optimise_volume(args, patches, sel_origins, vshape, pe, device, "/tmp", a.concurrent, queue, save=False,
                                         timings=timings)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    V = int(np.prod(a.patch))
    its = n_run * a.steps
    rate = its / dt
    roof_rate = FP32_PEAK_TFLOPS * 1e12 / (FLOP_PER_VOXEL_ITER * V)            # 1470 it/s per GPU at 64^3
    loop_rate = len(mine) * a.steps / max(timings.get("loop_s", dt), 1e-9)
    return {"metric": "Adam iters/sec on 3D MultiRes-UNet per GPU", "value": round(rate, 3), "unit": "it/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "split": "f32 (3 x bf16 split)"}[a.precision], "data": "synthetic",
            "config": {"workload": "configs[2]: 256^3 synthetic volume, 50 %% missing traces, %dx%dx%d patches stride %d (%d windows); queue of %d "
                                   "patches (%d per rank) pulled from the shared counter, %d concurrent hipGraph patches per GPU, %d Adam iterations "
                                   "each; timed end to end incl. per-patch set-up, dpi_overlap_add, the all-reduce and normalisation; a step = one "
                                   "iteration of every patch in the queue" % (tuple(a.patch) + (a.patch[0] // 2, n_total, n_run, a.patches,
                                                                                 a.concurrent, a.steps)),
                       "patches_rank0": len(mine), "reconstructed_finite": bool(np.isfinite(rec).all())},
            "roofline": {"bound": "mfma", "unit": "it/s", "achieved": round(rate / world, 2), "peak": round(roof_rate, 1),
                         "frac": round(rate / world / roof_rate, 4), "traffic": None,
                         "note": "whole-job patch-iterations/s per GPU against the fp32-FMA roofline of one 64^3 iteration (107.0 GFLOP / 157.3 TFLOP/s); "
                                 "the per-kernel roofline of the dominant conv is on the c2 line",
                         "loop_only_it_per_s_rank0": round(loop_rate, 2), "setup_s_rank0": round(timings.get("setup_s", 0.0), 3),
                         "loop_s_rank0": round(timings.get("loop_s", 0.0), 3)},
            "cpu_baseline": None}


def run_c4(a, rank, world, device):
    """configs[3] data: 2-D MulResUnet on the datasets/lines section; hipGraph loop without the regulariser, eager with it."""
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    z = np.load(os.path.join(ROOT, "tests", "golden", "host.npz"))
    img, mask = z["lines/original"].astype(np.float64), z["lines/mask"].astype(np.float64)
    argv = ["--imgdir", "lines", "--datadim", "2d", "--net", "multiunet", "--inputdepth", "64", "--upsample", "linear", "--loss", "mae",
            "--gain", "1", "--epochs", str(a.steps + a.warmup + 2), "--gpu", "0", "--precision", a.precision]
    if a.aa_weight > 0:
        argv += ["--aa_weight", str(a.aa_weight)]
    args = parse_arguments(argv)
    u.set_seed(rank)
    T = Interpolator(args, "/tmp", device=device, seed=rank)
    T.load_data({"image": img, "mask": mask, "name": "0"})
    T.build_model()
    T.build_input()
    T.build_regularizer()
    from deep_prior_interpolation_amd.optim import FusedAdam
    T.optimizer = FusedAdam(T.net.parameters(), lr=args.lr)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)
    if a.aa_weight > 0:
        def step():
            T.optimizer.zero_grad()
            T.optimization_loop()
            T.optimizer.step()
        mode = "eager (regulariser in the loop)"
    else:
        graph = T.graph_prepare()
        step = graph.replay
        mode = "graph"
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if a.aa_weight <= 0:
        T.graph_finish()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    flop_iter, bmin = 6.2e9, 0.278e9                         # 2-D 170x100 row
    ms = dt / a.steps * 1e3
    return {"metric": "Adam iters/sec on 2D MultiRes-UNet per GPU (datasets/lines)", "value": round(world * a.steps / dt, 2), "unit": "it/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.precision == "fp32" else a.precision, "data": "datasets/lines section (recorded fixture)",
            "config": {"workload": "configs[3] data: 2-D MulResUnet defaults (%d params) on datasets/lines 170x100, random66 mask, gain 1, MAE, bilinear, "
                                   "loop mode %s, aa_weight %g; reference notebook: 21 it/s on a V100 (different hardware, no regulariser)"
                                   % (T.num_params, mode, a.aa_weight), "last_loss": T.history.loss[-1] if T.history.loss else None},
            "roofline": {"bound": "launch", "unit": "it/s", "achieved": round(a.steps / dt, 2), "peak": round(1.0 / max(flop_iter / (FP32_PEAK_TFLOPS * 1e12), bmin / (HBM_PEAK_GBS * 1e9)), 1),
                         "frac": round((a.steps / dt) * max(flop_iter / (FP32_PEAK_TFLOPS * 1e12), bmin / (HBM_PEAK_GBS * 1e9)), 5), "traffic": None,
                         "note": "17 000 pixels: 6.2 GFLOP and 0.28 GB per iteration (roofline 0.04 ms); the iteration is ~400 dependent launches of a "
                                 "few microseconds each, i.e. bound by launch / dependency latency, not by a throughput roofline"},
            "cpu_baseline": None}


def main():
    global PRECISION
    a = parse()
    PRECISION = a.precision
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
    out = {"c2": run_c2, "c3": run_c3, "c4": run_c4}[a.workload](a, rank, world, device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
