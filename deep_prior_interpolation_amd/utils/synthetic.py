"""Synthetic seismic-like volumes for benchmarks and parity runs (the reference's hyperbolic3d dataset is absent
from its checkout): a (t, x, y) cube of hyperbolic events with a Ricker wavelet, apex at (x,y)=(0,0)."""
import numpy as np

__all__ = ["hyperbolic_volume", "random_trace_mask"]


def hyperbolic_volume(shape, seed=0, nev=5, dtype=np.float32):
    rng = np.random.RandomState(seed)
    nt, nx, ny = shape
    t = np.arange(nt, dtype=np.float32)[:, None, None]
    r2 = (np.arange(nx, dtype=np.float32)[None, :, None] / max(nx, 1)) ** 2 + \
         (np.arange(ny, dtype=np.float32)[None, None, :] / max(ny, 1)) ** 2
    vol = np.zeros(shape, dtype=np.float32)
    for _ in range(nev):
        t0 = rng.uniform(0.08, 0.6) * nt
        v = rng.uniform(0.6, 1.6)
        amp = rng.uniform(0.5, 1.0) * rng.choice([-1.0, 1.0])
        tt = np.sqrt(t0 ** 2 + r2 * (nt / v) ** 2)
        a = (np.pi * 0.25 * (t - tt)) ** 2
        vol += (amp * (1.0 - 2.0 * a) * np.exp(-a)).astype(np.float32)
    return (vol / np.abs(vol).max()).astype(dtype)


def random_trace_mask(shape, rate, seed=0, dtype=np.float32):
    """Binary mask with `rate` of the (x,y) traces deleted, constant along t (build_mask(regular=False) semantics)."""
    rng = np.random.RandomState(seed)
    ntr = int(np.prod(shape[1:]))
    keep = np.ones(ntr, dtype=dtype)
    keep[rng.choice(ntr, int(ntr * rate), replace=False)] = 0
    return np.broadcast_to(keep.reshape((1,) + tuple(shape[1:])), shape).copy()
