"""Building blocks shared by the network families (drop-in for reference architectures/base.py).

Factories are parametrised by `nd` (2 or 3) instead of being duplicated per dimensionality.  Child
names and construction order follow the reference so that `state_dict()` keys and — for equal seeds —
the initial parameter values are identical (cited per function).
"""
from torch import nn

from .. import nn as hnn

__all__ = ["get_activation", "conv", "conv3d", "conv2dbn", "conv3dbn", "conv_bn_act", "Concat", "Concat3D", "Seq"]

Seq = hnn.Seq
Concat = hnn.Concat
Concat3D = hnn.Concat


def get_activation(act_fun="LeakyReLU"):
    """reference base.py:97-114."""
    if act_fun == "LeakyReLU":
        return hnn.LeakyReLU(0.2)
    if act_fun == "ReLU":
        return hnn.LeakyReLU(0.0)
    if act_fun == "none":
        return nn.Sequential()
    if act_fun in ("ELU", "Tanh", "Sigmoid"):
        return hnn.Activation(act_fun)
    raise NotImplementedError("unknown activation function %r" % (act_fun,))


def conv_nd(nd, in_f, out_f, kernel_size, stride=1, bias=True):
    """'same' zero-padded convolution wrapped in a one-element Sequential (key '0'); base.py:117-126,169-180."""
    cls = hnn.Conv3d if nd == 3 else hnn.Conv2d
    return Seq(cls(in_f, out_f, kernel_size, stride, padding=int((kernel_size - 1) / 2), bias=bias))


def conv(in_f, out_f, kernel_size, stride=1, bias=True):
    return conv_nd(2, in_f, out_f, kernel_size, stride, bias)


def conv3d(in_f, out_f, kernel_size, stride=1, bias=True):
    return conv_nd(3, in_f, out_f, kernel_size, stride, bias)


def conv_bn_act(nd, in_f, out_f, kernel_size=3, stride=1, bias=True, act_fun="LeakyReLU"):
    """conv -> BatchNorm -> activation.
    3-D (base.py:211-216): Sequential(conv, BN, act)         -> child names 0, 1, 2
    2-D (base.py:162-166): conv Sequential + .add(BN, act)   -> child names 0, 2, 3"""
    if nd == 3:
        return ConvBnAct3d(conv_nd(3, in_f, out_f, kernel_size, stride, bias), hnn.BatchNorm3d(out_f), get_activation(act_fun))
    block = ConvBnAct2d(hnn.Conv2d(in_f, out_f, kernel_size, stride, padding=int((kernel_size - 1) / 2), bias=bias))
    block.add(hnn.BatchNorm2d(out_f))
    block.add(get_activation(act_fun))
    return block


def conv2dbn(in_f, out_f, kernel_size, stride=1, bias=True, act_fun="LeakyReLU"):
    return conv_bn_act(2, in_f, out_f, kernel_size, stride, bias, act_fun)


def conv3dbn(in_f, out_f, kernel_size=3, stride=1, bias=True, act_fun="LeakyReLU"):
    return conv_bn_act(3, in_f, out_f, kernel_size, stride, bias, act_fun)


class _ConvBnActMixin:
    """conv -> BN -> LeakyReLU with the activation folded into the BN-apply pass (one kernel less)."""

    def _parts(self):
        mods = list(self._modules.values())
        conv_m = mods[0][0] if isinstance(mods[0], nn.Sequential) else mods[0]
        return conv_m, mods[1], mods[2]

    def forward(self, x):
        conv_m, bn, act = self._parts()
        if isinstance(act, hnn.LeakyReLU):
            from .. import ops
            return ops.conv_bn_act(x, conv_m.weight, conv_m.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, conv_m._s, act.negative_slope)
        return act(bn(conv_m(x)))


class ConvBnAct3d(_ConvBnActMixin, Seq):
    pass


class ConvBnAct2d(_ConvBnActMixin, Seq):
    pass
