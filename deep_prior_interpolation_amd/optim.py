"""Adam on the GPU in one launch for all parameter tensors (replaces torch.optim.Adam at reference main.py:200,213)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    """Same update rule and defaults as torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0).

    State (exp_avg, exp_avg_sq) lives in two flat device buffers; `step()` issues a single
    dpi_adam_multi launch.  The step counter and learning rate live on the device (`step_lr`), so a
    captured hipGraph of the iteration keeps advancing them on replay; `set_lr` / ReduceLROnPlateau-style
    schedulers only rewrite that device scalar."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdam supports a single parameter group")
        if not ps or not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
            raise _lib.DpiError("FusedAdam needs fp32 parameters on the GPU (no CPU path)")
        self._params = ps
        dev = ps[0].device
        sizes = [p.numel() for p in ps]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self.exp_avg = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self._m = [self.exp_avg[offs[i]:offs[i + 1]] for i in range(len(ps))]
        self._v = [self.exp_avg_sq[offs[i]:offs[i + 1]] for i in range(len(ps))]
        self._sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self.step_lr = torch.tensor([0.0, lr], dtype=torch.float32, device=dev)
        self.active = torch.ones(1, dtype=torch.int32, device=dev)
        self._table_host = torch.empty(len(ps) * 4, dtype=torch.int64).pin_memory()
        self._table = torch.empty(len(ps) * 4, dtype=torch.int64, device=dev)
        self._table_key = None
        self._table_copied = None       # event recorded after the last async H2D copy out of _table_host

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = lr
        self.step_lr[1] = lr

    def _refresh_table(self):
        key = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self._params)
        if key == self._table_key:
            return
        rows = []
        for p, m, v in zip(self._params, self._m, self._v):
            if p.grad is None:
                raise _lib.DpiError("FusedAdam.step: a parameter has no gradient")
            g = p.grad
            if not g.is_contiguous():
                raise _lib.DpiError("FusedAdam.step: non-contiguous gradient")
            rows += [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()]
        if torch.cuda.is_current_stream_capturing():
            # an H2D copy node would re-read the pinned staging buffer on every replay — after later eager steps rewrote it.
            # Callers run one eager iteration first (Interpolator.graph_prepare), so the table is normally current here.
            raise _lib.DpiError("FusedAdam.step: gradient buffers changed inside a graph capture; run one eager step first")
        if self._table_copied is not None:
            self._table_copied.synchronize()         # the previous async copy must have left the pinned buffer
        self._table_host.copy_(torch.tensor(rows, dtype=torch.int64))
        self._table.copy_(self._table_host, non_blocking=True)
        self._table_copied = torch.cuda.Event()
        self._table_copied.record()
        self._table_key = key

    @torch.no_grad()
    def step(self, closure=None):
        self._refresh_table()
        self.step_lr[0] += 1.0
        b1, b2 = self.param_groups[0]["betas"]
        check(_lib.load().dpi_adam_multi(ptr(self._table), ptr(self._sizes), len(self._params), ptr(self.step_lr),
                                         b1, b2, self.param_groups[0]["eps"], ptr(self.active), stream()), "dpi_adam_multi")


class DevicePlateau:
    """ReduceLROnPlateau(mode='min', threshold_mode='rel') semantics of reference main.py:201-204, evaluated on the
    host from the loss value the loop already reads back; writes the new lr into the optimiser's device scalar."""

    def __init__(self, optimizer, factor, threshold, patience, min_lr=0.0, eps=1e-8):
        self.opt, self.factor, self.threshold, self.patience, self.min_lr, self.eps = optimizer, factor, threshold, patience, min_lr, eps
        self.best = float("inf")
        self.bad = 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            lr = self.opt.param_groups[0]["lr"]
            new_lr = max(lr * self.factor, self.min_lr)
            if lr - new_lr > self.eps:
                self.opt.set_lr(new_lr)
            self.bad = 0
