"""world_size-2 gloo tests (CPU) of the patch-parallel path: sharding, the single all-reduce of the overlap-add
accumulator and the normalisation must reproduce the single-process / oracle reassembly exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deep_prior_interpolation_amd import parallel as P
from deep_prior_interpolation_amd import utils as u
from oracle import dpi_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_optimise(i, patch):
    """Deterministic stand-in for the per-patch optimisation (a function of the patch and its index)."""
    return patch * 1.5 + 0.01 * i


def _worker(rank, world, port, shape, dim, stride, gain, outdir, dynamic=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    vol = rng.randn(*shape)
    pe = u.PatchExtractor(dim=dim, stride=stride)
    patches = pe.extract(vol).reshape((-1,) + dim)
    origins = u.window_origins(shape, dim, stride)
    queue, fn = None, _fake_optimise
    if dynamic:
        import time
        from torch.distributed import distributed_c10d
        queue = P.PatchQueue.for_process_group(len(patches))
        store = distributed_c10d._get_default_store()
        state = {"held": False}

        def fn(i, patch):
            # rank 0 is the slow rank, made slow by an explicit hand-shake instead of a sleep ratio (no wall-clock assumption): it
            # holds its FIRST patch until rank 1 has finished 30 of the 48 — the shared counter must have routed those to rank 1.
            # Rank 1 in turn does not finish its first patch before rank 0 holds one (on a loaded box rank 1 could otherwise drain the
            # whole queue before rank 0's process gets to its first claim, and "rank 0 took at least one" would fail for scheduling reasons)
            if rank == 1:
                if not state["held"]:
                    state["held"] = True
                    while int(store.add("dpi/test/rank0_holding", 0)) < 1:
                        time.sleep(0.001)
                store.add("dpi/test/rank1_done", 1)
            elif not state["held"]:
                state["held"] = True
                store.add("dpi/test/rank0_holding", 1)
                while int(store.add("dpi/test/rank1_done", 0)) < 30:
                    time.sleep(0.001)
            return _fake_optimise(i, patch)
    rec, mine = P.run_patches(list(patches), origins, shape, dim, stride, gain, fn, rank, world, queue=queue)
    np.save(os.path.join(outdir, "rec_%d.npy" % rank), rec)
    np.save(os.path.join(outdir, "mine_%d.npy" % rank), np.array(mine))
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,dim,stride", [((20, 18, 22), (8, 6, 10), (4, 4, 6)), ((16, 16, 16), (8, 8, 8), (8, 8, 8))])
def test_two_rank_reassembly_matches_oracle(tmp_path, shape, dim, stride):
    world, gain = 2, 40.0
    mp.spawn(_worker, args=(world, _free_port(), shape, dim, stride, gain, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.RandomState(0)
    vol = rng.randn(*shape)
    grid = O.patch_grid(shape, dim, stride)
    pa = O.extract_patches_nd(vol, dim, stride).reshape((-1,) + dim)
    outs = np.stack([_fake_optimise(i, p) for i, p in enumerate(pa)]).reshape(grid + dim)
    ref = O.reconstruct_nd(outs, dim, stride) / gain
    r0, r1 = np.load(tmp_path / "rec_0.npy"), np.load(tmp_path / "rec_1.npy")
    np.testing.assert_allclose(r0, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(r0, r1)                           # every rank holds the full volume after the all-reduce
    m0, m1 = np.load(tmp_path / "mine_0.npy"), np.load(tmp_path / "mine_1.npy")
    assert sorted(list(m0) + list(m1)) == list(range(len(pa))) and not set(m0) & set(m1)


def test_two_rank_shared_queue_balances_uneven_ranks(tmp_path):
    """PatchQueue over the c10d store: every patch is claimed exactly once, the fast rank takes more of them, and the
    re-assembled volume is the same as with the static split."""
    shape, dim, stride, world, gain = (20, 18, 22), (8, 6, 10), (4, 4, 6), 2, 40.0
    mp.spawn(_worker, args=(world, _free_port(), shape, dim, stride, gain, str(tmp_path), True), nprocs=world, join=True)
    vol = np.random.RandomState(0).randn(*shape)
    grid = O.patch_grid(shape, dim, stride)
    pa = O.extract_patches_nd(vol, dim, stride).reshape((-1,) + dim)
    outs = np.stack([_fake_optimise(i, p) for i, p in enumerate(pa)]).reshape(grid + dim)
    ref = O.reconstruct_nd(outs, dim, stride) / gain
    r0, r1 = np.load(tmp_path / "rec_0.npy"), np.load(tmp_path / "rec_1.npy")
    np.testing.assert_allclose(r0, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(r0, r1)
    m0, m1 = list(np.load(tmp_path / "mine_0.npy")), list(np.load(tmp_path / "mine_1.npy"))
    assert sorted(m0 + m1) == list(range(len(pa))) and not set(m0) & set(m1)
    assert len(pa) == 48 and len(m1) >= 30 and len(m0) >= 1, (len(m0), len(m1))


def test_patch_queue_local_and_static():
    q = P.PatchQueue(7)
    assert q.claim(3) == [0, 1, 2] and q.claim(3) == [3, 4, 5] and q.claim(3) == [6] and q.claim(3) == []
    q = P.PatchQueue(7, static=(1, 3))
    assert q.claim(2) == [1, 4] and q.claim(2) == [] and P.PatchQueue(7, static=(0, 3)).claim(9) == [0, 3, 6]


def test_shard_balance_config3():
    """configs[2]: 256^3 volume, 64^3 patches, stride 32 -> 343 patches; 8 ranks -> 43,43,...,42 (ideal speed-up 7.98x)."""
    n = u.count_patches((256, 256, 256), (64, 64, 64), (32, 32, 32))
    assert n == 343
    sizes = [len(P.shard_indices(n, r, 8)) for r in range(8)]
    assert sum(sizes) == n and max(sizes) == 43 and min(sizes) == 42
    assert sorted(sum((P.shard_indices(n, r, 8) for r in range(8)), [])) == list(range(n))


def test_single_process_host_accumulator_matches_reference_reconstruct():
    shape, dim, stride = (11, 7), (4, 3), (3, 2)
    vol = np.random.RandomState(1).randn(*shape)
    pe = u.PatchExtractor(dim=dim, stride=stride)
    pa = pe.extract(vol)
    rec, mine = P.run_patches(list(pa.reshape((-1,) + dim)), u.window_origins(shape, dim, stride), shape, dim, stride, 1.0,
                              lambda i, p: p)
    np.testing.assert_allclose(rec, pe.reconstruct(pa), rtol=1e-14)
    assert mine == list(range(int(np.prod(pa.shape[:2]))))


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_launcher_two_ranks():
    """`python bench.py --gpus 2` as the driver calls it, WITHOUT torch.distributed.run around it: the script starts its own rank
    processes, they rendezvous on 127.0.0.1, pull patches from the shared counter, all-reduce the accumulator, and rank 0's ONE
    JSON line comes back with n_gpus = 2 (round 2: `--gpus` was parsed and never read; the line said n_gpus 1).  CPU stand-in
    workload over gloo — the same launcher, queue and gather code the GPU workloads run."""
    rc, out, err = _bench("--gpus", "2", "--workload", "selftest", "--steps", "2")
    assert rc == 0, err
    assert out["n_gpus"] == 2 and out["gather"]["rccl_ranks"] == 2 and out["gather"]["backend"] == "gloo"
    assert sum(out["gather"]["patches_per_rank"]) == 5 and min(out["gather"]["patches_per_rank"]) >= 1
    assert out["gather"]["reconstruction_exact"] is True
    # under an external launcher (WORLD_SIZE set) the script is a plain rank and must not spawn again
    rc, out, err = _bench("--gpus", "1", "--workload", "selftest", "--steps", "1")
    assert rc == 0 and out["n_gpus"] == 1


def test_bench_launcher_reports_failure():
    """A rank that dies must make `bench.py --gpus N` exit non-zero instead of printing a line from the survivors."""
    rc, out, err = _bench("--gpus", "2", "--workload", "c2", "--steps", "1", env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert rc != 0 and out is None


def test_bench_launcher_stops_when_a_late_rank_dies():
    """rank 1 dies AFTER init_process_group while rank 0 sits in a barrier.  The launcher polls every rank, so it
    must return promptly (not after the 30-minute launch timeout / the c10d timeout), non-zero, with rank 1's stderr forwarded."""
    import time
    t0 = time.time()
    rc, out, err = _bench("--gpus", "2", "--workload", "selftest", "--steps", "2", env={"DPI_BENCH_TEST_FAIL_RANK": "1"})
    assert rc != 0 and out is None
    assert "rank 1 failed" in err and "exits on purpose" in err, err
    assert time.time() - t0 < 120


def test_launcher_parent_never_creates_a_hip_context():
    """the parent of `bench.py --gpus N` counts devices from the environment / sysfs — never through torch.cuda
    (hipGetDeviceCount on ROCm).  With torch.cuda.device_count / is_available poisoned in the parent only, the selftest job still
    spawns and finishes; and with an empty HIP_VISIBLE_DEVICES the GPU workload is refused before any rank starts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, runpy, torch\n"
            "def boom(*a, **k):\n"
            "    raise RuntimeError('launcher parent touched torch.cuda')\n"
            "if 'WORLD_SIZE' not in os.environ:\n"
            "    torch.cuda.device_count = boom; torch.cuda.is_available = boom; torch.cuda.init = boom; torch.cuda.set_device = boom\n"
            "sys.argv = [%r, '--gpus', '2', '--workload', 'c2', '--steps', '1']\n"
            "runpy.run_path(sys.argv[0], run_name='__main__')\n" % os.path.join(root, "bench.py"))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "only 0 HIP device(s) visible" in r.stderr and "touched torch.cuda" not in r.stderr, r.stderr
    import bench
    e2 = dict(os.environ)
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"
        os.environ.pop("ROCR_VISIBLE_DEVICES", None)
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
        n = bench.visible_gpu_count()
        assert n is not None and n <= 3
    finally:
        os.environ.clear()
        os.environ.update(e2)
