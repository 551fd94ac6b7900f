"""Patch-parallel execution across the GPUs of one node.

Patches are independent optimisations (reference main.py:274-295 iterates them sequentially), so rank r of W
takes patch indices {p : p mod W == r}; there is NO collective inside the hot path.  The only exchange is the
final reassembly (reference data.reconstruct_patches, data.py:87-130): every rank overlap-adds its best outputs
into a local full-volume accumulator and ONE all-reduce(sum) of that fp32 volume (RCCL over xGMI; gloo in the
CPU tests) followed by the analytic hit-count normalisation gives every rank the reconstructed volume.

One process per GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m deep_prior_interpolation_amd.parallel ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import utils as u

__all__ = ["shard_indices", "DeviceOverlapAccumulator", "HostOverlapAccumulator", "gather_volume", "run_patches", "main"]


def shard_indices(num_patches, rank, world):
    """Round-robin ownership: equal-cost patches => at most one patch of imbalance (343 patches / 8 = 43,43,...,42)."""
    return list(range(rank, num_patches, world))


class HostOverlapAccumulator:
    """numpy overlap-add with the reference's arithmetic (float64 accumulate, utils/patch_extractor.py:395-426).
    Used for 2-D / 2.5-D data, by data.reconstruct_patches-style host flows and by the gloo (CPU) tests."""

    def __init__(self, shape, dim, stride, device="cpu"):
        self.shape, self.dim, self.stride = tuple(shape), tuple(dim), tuple(stride)
        self.acc = torch.zeros(self.shape, dtype=torch.float64)

    def add(self, patch, origin):
        sl = tuple(slice(int(o), int(o) + d) for o, d in zip(origin, self.dim))
        self.acc[sl] += torch.as_tensor(np.asarray(patch), dtype=torch.float64)

    def tensor(self):
        return self.acc

    def finalize(self, gain):
        hits = torch.zeros(self.shape, dtype=torch.float64)
        for org in u.window_origins(self.shape, self.dim, self.stride):
            hits[tuple(slice(int(o), int(o) + d) for o, d in zip(org, self.dim))] += 1
        return (self.acc / hits / gain).numpy()


class DeviceOverlapAccumulator:
    """fp32 accumulator volume in HBM; dpi_overlap_add / dpi_overlap_normalize kernels (3-D volumes)."""

    def __init__(self, shape, dim, stride, device):
        if len(shape) != 3:
            raise NotImplementedError("device overlap-add handles 3-D volumes; use HostOverlapAccumulator otherwise")
        self.shape, self.dim, self.stride = tuple(int(s) for s in shape), tuple(int(d) for d in dim), tuple(int(s) for s in stride)
        self.acc = torch.zeros(self.shape, dtype=torch.float32, device=device)

    def add(self, patch, origin):
        p = patch if torch.is_tensor(patch) else torch.from_numpy(np.ascontiguousarray(patch, dtype=np.float32))
        p = p.to(self.acc.device, torch.float32).contiguous()
        if tuple(p.shape) != self.dim:
            raise _lib.DpiError("overlap_add: patch shape %s != %s" % (tuple(p.shape), self.dim))
        _lib.check(_lib.load().dpi_overlap_add(_lib.ptr(p), *self.dim, *[int(o) for o in origin], _lib.ptr(self.acc),
                                               *self.shape, _lib.stream()), "dpi_overlap_add")

    def tensor(self):
        return self.acc

    def finalize(self, gain):
        _lib.check(_lib.load().dpi_overlap_normalize(_lib.ptr(self.acc), *self.shape, *self.dim, *self.stride, float(gain),
                                                     _lib.stream()), "dpi_overlap_normalize")
        return self.acc.cpu().numpy()


def gather_volume(acc):
    """The single data-path collective: all-reduce(sum) of the accumulator volume (no-op for world size 1)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc.tensor(), op=dist.ReduceOp.SUM)
    return acc


def run_patches(patches, origins, vol_shape, dim, stride, gain, optimise_fn, rank=0, world=1, accumulator_cls=None,
                device="cpu"):
    """Optimise this rank's shard with `optimise_fn(index, patch) -> best output (patch-shaped)`, overlap-add locally,
    all-reduce once, normalise.  Returns (reconstructed volume of the cropped shape, indices this rank processed)."""
    accumulator_cls = accumulator_cls or HostOverlapAccumulator
    cropped = u.in_content_cropped_shape(vol_shape, dim, stride)
    acc = accumulator_cls(cropped, dim, stride, device)
    mine = shard_indices(len(patches), rank, world)
    for i in mine:
        acc.add(optimise_fn(i, patches[i]), origins[i])
    gather_volume(acc)
    return acc.finalize(gain), mine


def _run_patches_concurrently(args, patches, origins, vol_shape, pe, T0, conc, rank, world, device, outpath):
    """This rank's shard, `conc` patches at a time on one GPU (DPI_CONCURRENT_PATCHES=K): every patch of a group gets its own
    Interpolator, stream and captured iteration graph (main.optimize_concurrently).  Patches whose loop cannot run as a
    graph (--save_every, data forgetting, >= 2^20 voxels) and flat patches are handled one by one as usual."""
    from .main import Interpolator, optimize_concurrently
    cropped = u.in_content_cropped_shape(vol_shape, pe.dim, pe.stride)
    acc = DeviceOverlapAccumulator(cropped, pe.dim, pe.stride, device)
    mine = shard_indices(len(patches), rank, world)
    Ts = [T0] + [Interpolator(args, outpath, device=device, seed=rank * 1000 + k) for k in range(1, conc)]
    for g0 in range(0, len(mine), conc):
        group = mine[g0:g0 + conc]
        live = []
        for T, i in zip(Ts, group):
            std = T.load_data(patches[i])
            if np.isclose(std, 0.0, atol=1e-12):
                T.out_best, T.elapsed = T.img * T.mask, 0.0
                T._best_for_acc = torch.from_numpy(np.ascontiguousarray(T.out_best[..., 0], dtype=np.float32))
                continue
            T.build_model()
            T.build_input()
            if T.graph_capable():
                live.append(T)
            else:
                T.optimize(verbose=False)
                T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
        optimize_concurrently(live)
        for T in live:
            T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
        for T, i in zip(Ts, group):
            acc.add(T._best_for_acc, origins[i])
            T.save_result()
            T.clean()
    gather_volume(acc)
    return acc.finalize(args.gain), mine


def main(argv=None):
    """Multi-GPU counterpart of main.main(): same flags; each rank optimises its shard, result files are written per
    patch exactly as in the single-process run, rank 0 additionally saves `reconstructed.npy`."""
    from .data import extract_patches, patch_extractor_for
    from .main import Interpolator
    from .parameter import parse_arguments
    args = parse_arguments(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise _lib.DpiError("no HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)       # "nccl" is RCCL on ROCm
    u.set_seed(0)
    outpath = os.path.join("./results/", args.outdir if args.outdir is not None else "run")
    if rank == 0:
        os.makedirs(outpath, exist_ok=True)
        u.write_args(os.path.join(outpath, "args.txt"), args)
    if world > 1:
        dist.barrier()
    patches = extract_patches(args)
    vol = np.load(os.path.join(args.imgdir, args.imgname), allow_pickle=True)
    pe = patch_extractor_for(vol.shape, args.patch_shape, args.patch_stride, args.datadim, args.imgchannel)
    origins = u.window_origins(vol.shape, pe.dim, pe.stride)
    T = Interpolator(args, outpath, device=device, seed=rank)

    def optimise(i, patch):
        std = T.load_data(patch)
        if np.isclose(std, 0.0, atol=1e-12):
            T.out_best, T.elapsed = T.img * T.mask, 0.0
            best = torch.from_numpy(np.ascontiguousarray(T.out_best[..., 0], dtype=np.float32))
        else:
            if T.net is None or not args.start_from_prev:
                T.build_model()
            T.build_input()
            T.optimize(verbose=False)
            best = T._out_best_dev.reshape(T._out_best_dev.shape[2:]) if args.datadim == "3d" else torch.from_numpy(T.out_best)
        T.save_result()
        T.clean()
        return best

    conc = int(os.environ.get("DPI_CONCURRENT_PATCHES", "1"))
    if args.datadim == "3d" and vol.ndim == 3 and conc > 1:
        rec, mine = _run_patches_concurrently(args, patches, origins, vol.shape, pe, T, conc, rank, world, device, outpath)
    elif args.datadim == "3d" and vol.ndim == 3:
        rec, mine = run_patches(patches, origins, vol.shape, pe.dim, pe.stride, args.gain, optimise, rank, world,
                                DeviceOverlapAccumulator, device)
    else:
        # 2-D / 2.5-D slabs: result files only; rank 0 re-assembles them on the host like the reference does
        from .data import reconstruct_patches
        mine = shard_indices(len(patches), rank, world)
        for i in mine:
            optimise(i, patches[i])
        if world > 1:
            dist.barrier()
        rec = reconstruct_patches(args) if rank == 0 else None
    if rank == 0:
        np.save(os.path.join(outpath, "reconstructed.npy"), rec)
        print("rank 0: %d patches total, %d local; reconstructed volume %s saved" % (len(patches), len(mine), rec.shape))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
