"""Vertical (time-axis) convolution with a wavelet (reference operators/signal.py:8-45)."""
import numpy as np
import torch

from .. import _lib
from .base import LinearOpFn

__all__ = ["VerticalConv"]


class VerticalConv(torch.nn.Module):
    """forward: every channel of a (1,C,H,W) section convolved along H with wavelet/2 ('same', zero padded); adjoint: the
    cross-correlation.  The reference builds an ntwav x ntwav Conv2d kernel whose only non-zero column is the centre one
    (kernel[:, :, n//2] = wavelet[::-1] / 2, signal.py:16-17), i.e. a 1-D filter along H: dpi_fir_axis0 with S = W."""

    def __init__(self, wavelet):
        super().__init__()
        w = np.asarray(wavelet, dtype=np.float64)
        assert w.ndim == 1 and w.size % 2 == 1, "odd-length 1-D wavelet expected"
        # Conv2d is a correlation with kernel k[i] = w[::-1][i] / 2  ==  a true convolution with w / 2 (what dpi_fir_axis0 computes)
        self._taps_fwd = (w / 2).astype(np.float32)
        self._taps_adj = (w[::-1] / 2).astype(np.float32).copy()
        self._dev = {}

    def _apply(self, x, adjoint):
        if x.ndim != 4:
            raise _lib.DpiError("VerticalConv expects a (B,C,H,W) tensor")
        key = (str(x.device), bool(adjoint))
        taps = self._dev.get(key)
        if taps is None:
            taps = torch.from_numpy(self._taps_adj if adjoint else self._taps_fwd).to(x.device)
            self._dev[key] = taps
        y = torch.empty_like(x)
        B, C_, H, W = x.shape
        _lib.check(_lib.load().dpi_fir_axis0(_lib.ptr(x), _lib.ptr(taps), int(taps.numel()), B * C_, H, W, _lib.ptr(y), _lib.stream()),
                   "dpi_fir_axis0")
        return y

    def forward(self, x):
        return LinearOpFn.apply(x, self, False)

    def adjoint(self, y):
        return LinearOpFn.apply(y, self, True)
