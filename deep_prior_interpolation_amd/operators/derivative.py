"""First-difference operators (reference operators/derivative.py:8-21)."""
import torch

from .. import _lib
from .base import LinearOpFn

__all__ = ["VerticalGrad", "AxisDerivative"]


class AxisDerivative(torch.nn.Module):
    """Finite difference along one axis of an N-d tensor as a linear operator with an exact adjoint (dpi_diff_axis).
    stencil: 'forward' | 'backward' | 'centered' | 'second' (reference utils/processing.py:139-181)."""
    _CODES = {"forward": 0, "backward": 1, "centered": 2, "second": 3}

    def __init__(self, axis, stencil="forward", spacing=1.0):
        super().__init__()
        if stencil not in self._CODES:
            raise ValueError("Stencil has to be centered, forward or backward")
        self.axis, self.stencil, self.spacing = int(axis), stencil, float(spacing)

    def _apply(self, x, adjoint):
        ax = self.axis % x.ndim
        outer = 1
        for s in x.shape[:ax]:
            outer *= int(s)
        n = int(x.shape[ax])
        inner = x.numel() // (outer * n)
        y = torch.empty_like(x)
        _lib.check(_lib.load().dpi_diff_axis(_lib.ptr(x), outer, n, inner, self._CODES[self.stencil], self.spacing, int(adjoint),
                                             _lib.ptr(y), _lib.stream()), "dpi_diff_axis")
        return y

    def forward(self, x):
        return LinearOpFn.apply(x, self, False)

    def adjoint(self, y):
        return LinearOpFn.apply(y, self, True)


class VerticalGrad(AxisDerivative):
    """y[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1] on BCHW tensors, last row zero; adjoint = its transpose (derivative.py:8-21)."""

    def __init__(self):
        super().__init__(axis=2, stencil="forward", spacing=1.0)
