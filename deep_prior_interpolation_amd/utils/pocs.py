"""POCS regulariser (drop-in for reference utils/pocs.py) with the transform on torch.fft (rocFFT).

The reference passes `torch.rfft(x, signal_ndim, onesided=False)` / `torch.irfft(...)` as transform pair (main_pocs.py:156-157);
both were removed in torch 1.8.  Their documented semantics — full two-sided spectrum as a real tensor with a trailing (re, im)
axis, unnormalised forward, 1/N inverse — are `fft_forward` / `fft_adjoint` below.  Thresholding therefore acts on real and
imaginary parts as independent reals, with threshold = max(real view) * perc / 100 (utils/pocs.py:5-19)."""
import torch

from .. import _lib

__all__ = ["POCS", "threshold", "compute_threshold", "fft_forward", "fft_adjoint"]


def fft_forward(x, signal_ndim):
    return torch.view_as_real(torch.fft.fftn(x, dim=tuple(range(-signal_ndim, 0)))).contiguous()


def fft_adjoint(X, signal_ndim):
    return torch.fft.ifftn(torch.view_as_complex(X.contiguous()), dim=tuple(range(-signal_ndim, 0))).real.contiguous()


def _req(t):
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.DpiError("POCS runs on fp32 GPU tensors (no CPU path)")
    return t.contiguous()


def compute_threshold(in_content: torch.Tensor, perc: float = 10) -> torch.Tensor:
    """max(in_content) * perc / 100 as a 1-element DEVICE tensor (the reference returns a python float: one host sync per
    iteration; here the value never leaves the GPU)."""
    x = _req(in_content)
    L = _lib.load()
    ws = torch.empty(L.dpi_max_ws_floats(x.numel()), dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.check(L.dpi_scaled_max(_lib.ptr(x), x.numel(), float(perc) / 100.0, _lib.ptr(ws), _lib.ptr(out), _lib.stream()), "dpi_scaled_max")
    return out


def threshold(in_content: torch.Tensor, thresh=None) -> torch.Tensor:
    x = _req(in_content)
    if thresh is None:
        thresh = compute_threshold(x)
    if not torch.is_tensor(thresh):
        thresh = torch.tensor([float(thresh)], dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    _lib.check(_lib.load().dpi_threshold(_lib.ptr(x), x.numel(), _lib.ptr(thresh), _lib.ptr(y), _lib.stream()), "dpi_threshold")
    return y


class POCS(torch.nn.Module):
    """weighted_data + weighted_mask * adjoint(threshold(forward(x)))  (utils/pocs.py:44-84).  forward_fn / adjoint_fn default
    to the FFT pair over the spatial axes of (1,C,...) tensors.  The result is used detached (main_pocs.py:183)."""

    def __init__(self, data, mask, weight, forward_fn=None, adjoint_fn=None, thresh_perc=None):
        super().__init__()
        nd = data.ndim - 2
        self.weighted_data = _req(weight * data)
        self.weighted_mask = _req(torch.ones_like(mask) - weight * mask)
        self.weight = weight
        self.forward_fn = forward_fn or (lambda x: fft_forward(x, nd))
        self.adjoint_fn = adjoint_fn or (lambda X: fft_adjoint(X, nd))
        self.thresh_perc = thresh_perc

    def __str__(self):
        return "POCS(weight=%.3f, fn=fft)" % self.weight

    __repr__ = __str__

    @torch.no_grad()
    def forward(self, x, thresh=None):
        X = _req(self.forward_fn(x))
        th = compute_threshold(X, self.thresh_perc) if self.thresh_perc is not None else thresh
        xr = _req(self.adjoint_fn(threshold(X, th)))
        y = torch.empty_like(xr)
        _lib.check(_lib.load().dpi_pocs_project(_lib.ptr(xr), _lib.ptr(self.weighted_data), _lib.ptr(self.weighted_mask), xr.numel(),
                                                _lib.ptr(y), _lib.stream()), "dpi_pocs_project")
        return y
