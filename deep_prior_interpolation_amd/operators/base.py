"""Operator algebra (reference operators/base.py:10-67): Chain, Hessian, dottest."""
import torch

from .. import _lib

__all__ = ["Chain", "dottest", "Hessian", "LinearOpFn"]


class LinearOpFn(torch.autograd.Function):
    """y = A x for a linear operator object with `_apply(x, adjoint)`; backward = A^T dy (and A dy for the adjoint call)."""

    @staticmethod
    def forward(ctx, x, op, adjoint):
        if not x.is_cuda or x.dtype != torch.float32:
            raise _lib.DpiError("operators run on fp32 GPU tensors (no CPU path)")
        ctx.op, ctx.adjoint = op, bool(adjoint)
        return op._apply(x.contiguous(), bool(adjoint))

    @staticmethod
    def backward(ctx, dy):
        return ctx.op._apply(dy.contiguous(), not ctx.adjoint), None, None


class Chain(torch.nn.Module):
    """ops[-1](...ops[0](x)); adjoint applies the adjoints in reverse order (base.py:10-37)."""

    def __init__(self, ops: list):
        super().__init__()
        assert len(ops) >= 1
        self.ops = ops

    def forward(self, x):
        for op in self.ops:
            x = op(x)
        return x

    def adjoint(self, x):
        for op in self.ops[::-1]:
            x = op.adjoint(x)
        return x

    def __getitem__(self, item):
        return self.ops[item]


class Hessian(torch.nn.Module):
    """A^T A (base.py:40-50); self-adjoint."""

    def __init__(self, op):
        super().__init__()
        self.op = op

    def forward(self, x):
        return self.op.adjoint(self.op.forward(x))

    def adjoint(self, x):
        return self.forward(x)


def dottest(op, domain_tensor, range_tensor, verbose=True, generator=None):
    """Adjoint dot-product test <A d1, r1> = <d1, A^T r1> on random vectors of the given shapes (base.py:53-67).
    Runs on the device of `domain_tensor`; returns (absolute error, relative error) besides printing them like the reference."""
    dev = domain_tensor.device
    d1 = torch.randn(domain_tensor.shape, generator=generator).to(dev)
    r1 = torch.randn(range_tensor.shape, generator=generator).to(dev)
    r2 = op.forward(d1)
    d2 = op.adjoint(r1)
    d_ = torch.vdot(d1.double().view(-1), d2.double().view(-1))
    r_ = torch.vdot(r1.double().view(-1), r2.double().view(-1))
    err_abs = d_ - r_
    err_rel = err_abs / d_
    if verbose:
        print("Absolute error: %.6e" % abs(err_abs.item()))
        print("Relative error: %.6e \n" % abs(err_rel.item()))
    return abs(err_abs.item()), abs(err_rel.item())
