"""Patch-parallel execution across the GPUs of one node.

Patches are independent optimisations (reference main.py:274-295 iterates them sequentially), so ranks pull patch
indices from a shared counter (PatchQueue: an atomic fetch-add on torch.distributed's store — no data-path collective;
skipped or early-stopped patches therefore never idle a rank; DPI_STATIC_SHARD=1 restores the static p mod W split).
Every patch seeds its own weights / z / noise stream from its index (Interpolator.begin_patch), so a patch's result does
not depend on the rank, the world size or the schedule — except with --start_from_prev, where a patch starts from the net of
the patch its Interpolator optimised before (reference main.py:286): that chain is only defined per process, so the flag forces
the static `p mod W` shard and one patch at a time per GPU, which makes the chains reproducible for a given W.
The only exchange is the
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

__all__ = ["shard_indices", "PatchQueue", "DeviceOverlapAccumulator", "HostOverlapAccumulator", "gather_volume", "run_patches",
           "optimise_volume", "main"]


def shard_indices(num_patches, rank, world):
    """Round-robin ownership: equal-cost patches => at most one patch of imbalance (343 patches / 8 = 43,43,...,42)."""
    return list(range(rank, num_patches, world))


class PatchQueue:
    """Work queue over patch indices 0..num-1 shared by all ranks.

    claim(n) hands out the next n indices (fewer at the end, [] when exhausted).  With a process group the counter lives
    in the c10d store (`store.add` is an atomic fetch-add served by rank 0's TCPStore — control plane only, a few bytes
    per patch); without one it is a local counter.  static=(rank, world) gives the round-robin split instead."""

    def __init__(self, num, store=None, key="dpi/next_patch", static=None):
        self.num, self.store, self.key = int(num), store, key
        self._local = 0
        self._static = None if static is None else shard_indices(self.num, *static)

    @classmethod
    def for_process_group(cls, num, key="dpi/next_patch", static=False):
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if static or os.environ.get("DPI_STATIC_SHARD", "0") == "1":
            return cls(num, static=(rank, world))
        if world > 1:
            from torch.distributed import distributed_c10d
            return cls(num, store=distributed_c10d._get_default_store(), key=key)
        return cls(num)

    def claim(self, n=1):
        n = int(n)
        if self._static is not None:
            out, self._static = self._static[:n], self._static[n:]
            return out
        if self.store is not None:
            end = int(self.store.add(self.key, n))
        else:
            self._local += n
            end = self._local
        return [i for i in range(end - n, end) if i < self.num]


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
                device="cpu", queue=None):
    """Optimise this rank's share with `optimise_fn(index, patch) -> best output (patch-shaped)`, overlap-add locally,
    all-reduce once, normalise.  `queue` (PatchQueue) hands out the indices; None = static round-robin shard.
    Returns (reconstructed volume of the cropped shape, indices this rank processed)."""
    accumulator_cls = accumulator_cls or HostOverlapAccumulator
    cropped = u.in_content_cropped_shape(vol_shape, dim, stride)
    acc = accumulator_cls(cropped, dim, stride, device)
    queue = queue or PatchQueue(len(patches), static=(rank, world))
    mine = []
    while True:
        got = queue.claim(1)
        if not got:
            break
        i = got[0]
        acc.add(optimise_fn(i, patches[i]), origins[i])
        mine.append(i)
    gather_volume(acc)
    return acc.finalize(gain), mine


class _Replayer:
    """Host thread that keeps the captured iteration graphs of the active concurrency slots replaying (round-robin, each slot at most
    `depth` replays ahead of the device) while the main thread sets up, captures and finishes patches.  hipGraphLaunch releases the GIL; the
    thread touches nothing but its entries' graphs, streams and events."""

    def __init__(self, device, depth):
        import threading
        self.device, self.depth = device, depth
        self.lock = threading.Lock()
        self.entries = []              # dicts: graph, stream, left (replays still to launch), events
        self.stop = False
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def add(self, graph, stream, replays):
        e = {"graph": graph, "stream": stream, "left": int(replays), "events": []}
        with self.lock:
            self.entries.append(e)
        return e

    def remove(self, e):
        with self.lock:
            if e in self.entries:
                self.entries.remove(e)

    def _run(self):
        from time import sleep
        try:
            torch.cuda.set_device(self.device)
            while not self.stop:
                with self.lock:
                    todo = [e for e in self.entries if e["left"] > 0]
                if not todo:
                    sleep(0.0005)
                    continue
                for e in todo:
                    if len(e["events"]) >= self.depth:
                        e["events"].pop(0).synchronize()     # the replay `depth` launches back has run
                    with self.lock:
                        if e not in self.entries or e["left"] <= 0:
                            continue                         # taken out (early stop) while this thread was waiting
                        with torch.cuda.stream(e["stream"]):
                            e["graph"].replay()
                            ev = torch.cuda.Event()
                            ev.record()
                        e["events"].append(ev)
                        e["left"] -= 1
        except Exception as exc:      # surfaced by the main thread (close())
            self.error = exc

    def close(self, primary=None):
        """Stop the thread.  A replay error is raised here — chained to `primary` (the exception already in flight in the main thread, if any)
        instead of replacing it."""
        self.stop = True
        self.thread.join()
        with self.lock:
            del self.entries[:]
        if self.error is not None:
            if primary is not None:
                raise self.error from primary
            raise self.error


def _optimise_rolling(args, patches, origins, acc, Ts, queue, save, device, check_every=64):
    """K concurrency slots, each with its own Interpolator, stream and captured iteration graph, kept full from the shared queue: a slot
    that has finished its patch (overlap-add, result file) claims the next index and prepares it — weights, z, iteration 0, the graph
    capture: ~50 ms of host time alone, ~250 ms next to five running patches (tools/c3_setup_probe.py) — while a replay thread keeps the
    other slots' graphs running.  Rounds 2-4 worked in groups of K: set up K patches (device idle), replay them to the end together,
    repeat; with 100-iteration patches that idled the device for 13 % of the job.
    Same arithmetic per patch as a standalone graph-mode optimize() (every patch is seeded from its index).  Patches that cannot run as a
    captured graph (>= 2^20 voxels, --save_every, data forgetting) and flat patches are handled on the spot, one by one.
    Returns (indices processed, host seconds spent preparing while other slots were running)."""
    from time import perf_counter, sleep
    streams = [torch.cuda.Stream(device=device) for _ in Ts]
    slot = [None] * len(Ts)            # per slot: {"i": patch index, "t0": start time, "entry": the replay thread's entry, "polled": replays at the last poll}
    mine, t_prep = [], [0.0]
    main_stream = torch.cuda.current_stream(device)
    rep = _Replayer(device, int(os.environ.get("DPI_SLOT_DEPTH", "6")))

    def done(T, i):
        acc.add(T._best_for_acc, origins[i])
        if save:
            T.save_result()
        T.clean()
        mine.append(i)

    def start(k):
        """Claim and prepare the next graph-capable patch for slot k; False when the queue is empty."""
        T, st = Ts[k], streams[k]
        while True:
            got = queue.claim(1)
            if not got:
                slot[k] = None
                return False
            i = got[0]
            t0 = perf_counter()
            std = T.load_data(patches[i])
            if np.isclose(std, 0.0, atol=1e-12):
                T.out_best, T.elapsed = T.img * T.mask, 0.0
                T._best_for_acc = torch.from_numpy(np.ascontiguousarray(T.out_best[..., 0], dtype=np.float32))
                done(T, i)
                continue
            T.begin_patch(i)
            T.build_model(netpath=args.netdir[i]) if len(args.netdir) != 0 else T.build_model()
            T.build_input()
            T.build_regularizer()
            if not T.graph_capable():
                T.optimize(verbose=False)
                T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
                done(T, i)
                continue
            st.wait_stream(main_stream)                      # data / weights / z were produced on the caller's stream
            with torch.cuda.stream(st):
                T.optimizer = None
                g = T.graph_prepare(quiet_device=False)
            slot[k] = {"i": i, "t0": t0, "entry": rep.add(g, st, args.epochs - 1), "polled": 0}
            t_prep[0] += perf_counter() - t0
            return True

    def finish(k):
        T, st = Ts[k], streams[k]
        rep.remove(slot[k]["entry"])
        with torch.cuda.stream(st):
            T.graph_finish(quiet_device=False)               # waits for THIS slot's queued replays only
        main_stream.wait_stream(st)
        T.elapsed = perf_counter() - slot[k]["t0"]
        T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
        done(T, slot[k]["i"])
    primary = None
    try:
        for k in range(len(Ts)):
            if not start(k):
                break
        while any(s_ is not None for s_ in slot):
            if rep.error is not None:                        # checked at the TOP of every poll round: no finish() / set-up of a next patch behind a failed replay
                break
            progressed = False
            for k, s_ in enumerate(slot):
                if s_ is None:
                    continue
                e = s_["entry"]
                launched = (args.epochs - 1) - e["left"]
                stopped = False
                if e["left"] > 0 and launched - s_["polled"] >= check_every:       # early stop / NaN decided on the device: stop replaying this patch
                    s_["polled"] = launched
                    with torch.cuda.stream(streams[k]):
                        stopped = int(Ts[k].optimizer.active.item()) == 0
                if e["left"] == 0 or stopped:
                    finish(k)
                    start(k)
                    progressed = True
            if not progressed:
                sleep(0.001)
    except BaseException as exc:
        primary = exc
        raise
    finally:
        # take every slot out of the replay thread's hands and let its stream drain BEFORE the Interpolators (graphs, private pools) are dropped
        for s_ in slot:
            if s_ is not None:
                rep.remove(s_["entry"])
        rep.stop = True
        rep.thread.join()
        for st in streams:
            try:
                st.synchronize()
            except Exception:                                # noqa: BLE001 — a failed stream must not mask the error that brought us here
                pass
        rep.close(primary)
    return mine, t_prep[0]


def optimise_volume(args, patches, origins, vol_shape, pe, device, outpath=None, conc=1, queue=None, save=True, timings=None):
    """Deep-prior optimisation of every patch this rank pulls from `queue`, `conc` patches at a time on one GPU, overlap-added
    into a device accumulator; ONE all-reduce at the end; returns (reconstructed volume, indices processed here).

    conc > 1 (DPI_CONCURRENT_PATCHES): every patch of a group gets its own Interpolator, stream and captured iteration graph
    (main.optimize_concurrently).  Patches whose loop cannot run as a graph (--save_every, data forgetting, >= 2^20 voxels)
    and flat patches are handled one by one.  save=False skips the per-patch result files (bench.py)."""
    from time import perf_counter
    from .main import Interpolator, optimize_concurrently
    cropped = u.in_content_cropped_shape(vol_shape, pe.dim, pe.stride)
    acc = DeviceOverlapAccumulator(cropped, pe.dim, pe.stride, device)
    if args.start_from_prev:
        if conc > 1:
            raise _lib.DpiError("--start_from_prev chains the patches of a process one after the other (reference main.py:286): "
                                "it cannot be combined with DPI_CONCURRENT_PATCHES > 1")
        if queue is not None and queue._static is None and (queue.store is not None):
            raise _lib.DpiError("--start_from_prev needs the static patch shard (a shared queue would make the warm-start chain depend on timing)")
    queue = queue or PatchQueue.for_process_group(len(patches), static=bool(args.start_from_prev))
    Ts = [Interpolator(args, outpath, device=device) for _ in range(max(conc, 1))]
    mine = []
    t_setup = t_loop = t_prep_live = 0.0
    if len(Ts) > 1 and os.environ.get("DPI_ROLLING_SLOTS", "1") == "1":
        t0 = perf_counter()
        mine, t_prep_live = _optimise_rolling(args, patches, origins, acc, Ts, queue, save, device)
        t_loop = perf_counter() - t0
        queue = PatchQueue(0)          # drained: the group loop below has nothing left to claim
    while True:
        group = queue.claim(len(Ts))
        if not group:
            break
        t0 = perf_counter()
        t_solo = 0.0
        live = []
        index_of = {}

        def prepare(T):
            i = index_of[id(T)]
            T.begin_patch(i)
            if T.net is None or not args.start_from_prev:
                T.build_model(netpath=args.netdir[i]) if len(args.netdir) != 0 else T.build_model()
            T.build_input()
            T.build_regularizer()
        for T, i in zip(Ts, group):
            std = T.load_data(patches[i])
            if np.isclose(std, 0.0, atol=1e-12):
                T.out_best, T.elapsed = T.img * T.mask, 0.0
                T._best_for_acc = torch.from_numpy(np.ascontiguousarray(T.out_best[..., 0], dtype=np.float32))
                continue
            index_of[id(T)] = i
            if len(Ts) > 1 and T.graph_capable():
                live.append(T)             # prepared inside optimize_concurrently, patch by patch, while the earlier ones already run
            else:
                prepare(T)
                torch.cuda.synchronize(device)
                to = perf_counter()
                T.optimize(verbose=False)
                T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
                t_solo += perf_counter() - to          # a patch optimised on its own (>= 2^20 voxels, --save_every, ...): loop time, not set-up
        t1 = perf_counter()
        ct = {}
        optimize_concurrently(live, prepare=prepare, timings=ct)
        t_prep_live += ct.get("prepare_s", 0.0)
        for T in live:
            T._best_for_acc = T._out_best_dev.reshape(T._out_best_dev.shape[2:])
        for T, i in zip(Ts, group):
            acc.add(T._best_for_acc, origins[i])
            if save:
                T.save_result()
            T.clean()
            mine.append(i)
        t_setup += t1 - t0 - t_solo
        t_loop += perf_counter() - t1 + t_solo
    if timings is not None:          # (bench.py) the collective ALONE, bracketed by device synchronisations, beside the rank's own seconds
        torch.cuda.synchronize(device)
        t2 = perf_counter()
        gather_volume(acc)
        torch.cuda.synchronize(device)
        # (prepare_overlapped_s: host seconds of per-patch set-up + graph capture spent INSIDE loop_s, while earlier patches of the group ran)
        timings.update(setup_s=t_setup, loop_s=t_loop, prepare_overlapped_s=t_prep_live, collective_s=perf_counter() - t2, patches=len(mine))
    else:
        gather_volume(acc)
    rec = acc.finalize(args.gain)
    return rec, mine


def main(argv=None):
    """Multi-GPU counterpart of main.main(): same flags; each rank optimises the patches it pulls from the shared queue, result
    files are written per patch exactly as in the single-process run, rank 0 additionally saves `reconstructed.npy`."""
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
    args.outdir = args.outdir or "run"       # reconstruct_patches(args) joins it too
    outpath = os.path.join("./results/", args.outdir)
    if rank == 0:
        os.makedirs(outpath, exist_ok=True)
        u.write_args(os.path.join(outpath, "args.txt"), args)
    if world > 1:
        dist.barrier()
    patches = extract_patches(args)
    vol = np.load(os.path.join(args.imgdir, args.imgname), allow_pickle=True)
    pe = patch_extractor_for(vol.shape, args.patch_shape, args.patch_stride, args.datadim, args.imgchannel)
    origins = u.window_origins(vol.shape, pe.dim, pe.stride)
    conc = int(os.environ.get("DPI_CONCURRENT_PATCHES", "1"))
    queue = PatchQueue.for_process_group(len(patches), static=bool(args.start_from_prev))
    if args.datadim == "3d" and vol.ndim == 3:
        if (args.imgchannel or 1) != 1:
            raise _lib.DpiError("the device overlap-add path re-assembles single-channel 3-D volumes (imgchannel = 1)")
        rec, mine = optimise_volume(args, patches, origins, vol.shape, pe, device, outpath, conc, queue)
    else:
        # 2-D / 2.5-D slabs: result files only; rank 0 re-assembles them on the host like the reference does
        from .data import reconstruct_patches
        T = Interpolator(args, outpath, device=device)
        mine = []
        while True:
            got = queue.claim(1)
            if not got:
                break
            i = got[0]
            std = T.load_data(patches[i])
            if np.isclose(std, 0.0, atol=1e-12):
                T.out_best, T.elapsed = T.img * T.mask, 0.0
            else:
                T.begin_patch(i)
                if T.net is None or not args.start_from_prev:
                    T.build_model(netpath=args.netdir[i]) if len(args.netdir) != 0 else T.build_model()
                T.build_input()
                T.build_regularizer()
                T.optimize(verbose=False)
            T.save_result()
            T.clean()
            mine.append(i)
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
