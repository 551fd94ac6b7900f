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


def _worker(rank, world, port, shape, dim, stride, gain, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    vol = rng.randn(*shape)
    pe = u.PatchExtractor(dim=dim, stride=stride)
    patches = pe.extract(vol).reshape((-1,) + dim)
    origins = u.window_origins(shape, dim, stride)
    rec, mine = P.run_patches(list(patches), origins, shape, dim, stride, gain, _fake_optimise, rank, world)
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
