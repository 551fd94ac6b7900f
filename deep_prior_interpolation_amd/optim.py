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
        self.step_lr = torch.tensor([0.0, lr], dtype=torch.float32, device=dev)
        self.active = torch.ones(1, dtype=torch.int32, device=dev)
        self._table_host = torch.empty(len(ps) * 5, dtype=torch.int64).pin_memory()     # 4 pointers + 1 size per tensor
        self._table = torch.empty(len(ps) * 5, dtype=torch.int64, device=dev)
        self._table_key = None
        self._n_active = 0
        self._capture_tables = []
        self._capture_ready = None
        self._table_copied = None       # event recorded after the last async H2D copy out of _table_host

    def zero_grad(self, set_to_none=True):
        """Gradients are always dropped, never zeroed in place: the fused nodes hand autograd freshly written tensors which AccumulateGrad adopts
        as .grad (no copy, no add pass); with the once-per-step join (ops.JOIN_AT) a surviving .grad would be accumulated into on the main stream
        before the side stream has produced the new gradient."""
        if not set_to_none:
            raise _lib.DpiError("FusedAdam.zero_grad(set_to_none=False) is not supported: gradients are adopted, not accumulated (ops.finish_backward)")
        super().zero_grad(set_to_none=True)

    def prepare_capture(self):
        """Allocate the pinned staging buffer + device table the next captured step() will use (see _refresh_table)."""
        n = len(self._params) * 5
        self._capture_ready = (torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(n, dtype=torch.int64, device=self._table.device))

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = lr
        self.step_lr[1] = lr

    def _rows(self):
        """(pointer rows, sizes) of the parameters that received a gradient.  Parameters whose gradient is None are skipped, as
        torch.optim.Adam does (main.py:200): the conv biases that feed a BatchNorm have an analytically zero gradient (SURVEY
        App. D) and the fused nodes return None for them — with zero moments their Adam update is exactly 0 either way."""
        rows, sizes = [], []
        for p, m, v in zip(self._params, self._m, self._v):
            g = p.grad
            if g is None:
                continue
            if not g.is_contiguous():
                raise _lib.DpiError("FusedAdam.step: non-contiguous gradient")
            rows += [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()]
            sizes.append(p.numel())
        if not sizes:
            raise _lib.DpiError("FusedAdam.step: no parameter has a gradient")
        return rows, sizes

    def _refresh_table(self):
        """Device table {p, g, m, v} x tensors for this step; returns (table, sizes, count)."""
        key = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self._params)
        if torch.cuda.is_current_stream_capturing():
            # Inside a hipGraph capture the gradients live in the graph's private pool (other addresses than in eager mode) and the
            # H2D copy becomes a graph node that re-reads its host buffer on EVERY replay: give the capture its own pinned staging
            # buffer and device table, kept alive with the optimiser, so later eager steps cannot overwrite what a replay reads.
            # (Pinned memory cannot be allocated while a stream is capturing: prepare_capture() did that beforehand.)
            if self._capture_ready is None:
                raise _lib.DpiError("FusedAdam.step inside a graph capture: call prepare_capture() before torch.cuda.graph(...)")
            host, dev = self._capture_ready
            self._capture_ready = None
            rows, sizes = self._rows()
            n = len(sizes)
            host[:5 * n].copy_(torch.tensor(rows + sizes, dtype=torch.int64))
            dev[:5 * n].copy_(host[:5 * n], non_blocking=True)
            self._capture_tables.append((host, dev))
            return dev[:4 * n], dev[4 * n:5 * n], n
        if key != self._table_key:
            rows, sizes = self._rows()
            n = len(sizes)
            if self._table_copied is not None:
                self._table_copied.synchronize()         # the previous async copy must have left the pinned buffer
            self._table_host[:5 * n].copy_(torch.tensor(rows + sizes, dtype=torch.int64))
            self._table[:5 * n].copy_(self._table_host[:5 * n], non_blocking=True)
            self._table_copied = torch.cuda.Event()
            self._table_copied.record()
            self._table_key, self._n_active = key, n
        n = self._n_active
        return self._table[:4 * n], self._table[4 * n:5 * n], n

    @torch.no_grad()
    def step(self, closure=None):
        table, sizes, n = self._refresh_table()
        self.step_lr[0] += 1.0
        b1, b2 = self.param_groups[0]["betas"]
        check(_lib.load().dpi_adam_multi(ptr(table), ptr(sizes), n, ptr(self.step_lr),
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
