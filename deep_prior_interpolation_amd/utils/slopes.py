"""Local dips and the directional Laplacian (drop-in for reference utils/slopes.py) on the GPU."""
from typing import Tuple

import torch

from .. import _lib
from ..operators.base import LinearOpFn
from .processing import GaussianFilter

__all__ = ["Hale2D", "directional_laplacian", "structure_tensor_dips"]


def _planes(t):
    if t.ndim != 4:
        raise _lib.DpiError("expected a BCHW tensor")
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.DpiError("slopes operators run on fp32 GPU tensors (no CPU path)")
    return t.contiguous(), t.shape[0] * t.shape[1], int(t.shape[2]), int(t.shape[3])


def structure_tensor_dips(in_content: torch.Tensor, dv: float = 1., dh: float = 1, smooth: float = 0.) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dip angle and anisotropy from the (optionally Gaussian-smoothed) structure tensor of a BCHW section
    (slopes.py:6-48).  No gradient is propagated: dips are a fixed field the regulariser is built on."""
    with torch.no_grad():
        x, N, H, W = _planes(in_content)
        L = _lib.load()
        gvv, gvh, ghh = (torch.empty_like(x) for _ in range(3))
        _lib.check(L.dpi_structure_tensor(_lib.ptr(x), N, H, W, float(dv), float(dh), _lib.ptr(gvv), _lib.ptr(gvh), _lib.ptr(ghh),
                                          _lib.stream()), "dpi_structure_tensor")
        if smooth > 0:
            G = GaussianFilter(channels=x.shape[1], kernel_size=2 * min(H, W) // 2 + 1, ndim=2, std=smooth)
            gvv, gvh, ghh = G(gvv), G(gvh), G(ghh)
        phi, aniso = torch.empty_like(x), torch.empty_like(x)
        _lib.check(L.dpi_dips(_lib.ptr(gvv), _lib.ptr(gvh), _lib.ptr(ghh), x.numel(), _lib.ptr(phi), _lib.ptr(aniso), _lib.stream()),
                   "dpi_dips")
        return phi, aniso


class Hale2D(torch.nn.Module):
    """Directional Laplacian built on a dip field (BCHW, same shape as the sections it is applied to) — slopes.py:72-105.
    forward(x) = -(Dh(a Dv x + b Dh x) + Dv(b Dv x + c Dh x)) with a = cos^2, b = -cos sin, c = sin^2; differentiable (the
    backward is the exact transpose kernel), plus an explicit `adjoint` the reference does not have."""

    def __init__(self, directions: torch.Tensor):
        super().__init__()
        with torch.no_grad():
            u1 = torch.cos(directions)
            u2 = -torch.sin(directions)
            self.a = (u1 * u1).contiguous()
            self.b = (u1 * u2).contiguous()
            self.c = (u2 * u2).contiguous()
            self.dips = directions

    def _apply(self, x, adjoint):
        x, N, H, W = _planes(x)
        if tuple(x.shape) != tuple(self.a.shape):
            raise _lib.DpiError("Hale2D: tensor shape %s differs from the dip field %s" % (tuple(x.shape), tuple(self.a.shape)))
        y = torch.empty_like(x)
        _lib.check(_lib.load().dpi_hale2d(_lib.ptr(x), _lib.ptr(self.a), _lib.ptr(self.b), _lib.ptr(self.c), N, H, W, int(adjoint),
                                          _lib.ptr(y), _lib.stream()), "dpi_hale2d")
        return y

    def forward(self, inputs):
        return LinearOpFn.apply(inputs, self, False)

    def adjoint(self, y):
        return LinearOpFn.apply(y, self, True)


def directional_laplacian(in_content: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    return Hale2D(theta)(in_content)
