"""Sampling-mask synthesis (drop-in for reference utils/mask.py build_mask / add_rand_mask).
Uses numpy's global legacy RNG exactly like the reference so that np.random.seed(s) reproduces its masks."""
import numpy as np

__all__ = ["build_mask", "add_rand_mask"]


def build_mask(data, rate, regular=False):
    """Binary trace mask for a (t, x[, y]) cube with `rate` of the traces deleted (constant along t)."""
    if data.ndim == 2:
        nt, nx = data.shape
        ny = 1
    elif data.ndim == 3:
        nt, nx, ny = data.shape
    else:
        raise ValueError("data volume has to be either 2D or 3D")
    ntr = nx * ny
    ndel = int(ntr * rate)
    if regular:
        keep_few = rate >= 0.5
        n = ntr - ndel if keep_few else ndel
        m = int(np.ceil(ntr / n))
        tr = np.ones(ntr) if keep_few else np.zeros(ntr)
        for i in range(n):
            tr[i * m + 1:i * m + m] = 0 if keep_few else 1
    else:
        tr = np.ones(ntr)
        tr[np.random.choice(np.arange(ntr), ndel, replace=False)] = 0
    mask = np.broadcast_to(tr.astype(data.dtype)[None, :], (nt, ntr)).copy()
    return mask.reshape((nt, nx, ny)).squeeze()


def add_rand_mask(mask, perc=0.3):
    """Delete a further `perc` of the surviving traces (data.py:79-80, --adirandel)."""
    m = mask.copy()
    pts = np.argwhere(m[0] == 1)
    sel = np.random.choice(np.arange(pts.shape[0]), int(pts.shape[0] * perc), replace=False)
    for p in pts[sel]:
        m[(slice(None),) + tuple(p)] = 0
    return m
