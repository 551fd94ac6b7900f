"""Leaf modules of the drop-in network tree.

They subclass the torch parameter containers (so `state_dict` keys, default initialisation, RNG
consumption order and `__class__.__name__` — which the reference's `init_weights` greps for 'Conv' /
'BatchNorm', utils/torch.py:34-53 — are exactly torch's), but every `forward` runs the HIP kernels of
libdpi_hip.so through `ops`; aten's conv / batch_norm / upsample are never called.
"""
import torch
from torch import nn

from . import ops


class Conv3d(nn.Conv3d):
    def __init__(self, in_f, out_f, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(in_f, out_f, kernel_size, stride, padding=padding, bias=bias)
        _check_conv(self, kernel_size, stride, padding)
        self._s = int(stride)

    def forward(self, x):
        return ops.conv(x, self.weight, self.bias, self._s)


class Conv2d(nn.Conv2d):
    def __init__(self, in_f, out_f, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(in_f, out_f, kernel_size, stride, padding=padding, bias=bias)
        _check_conv(self, kernel_size, stride, padding)
        self._s = int(stride)

    def forward(self, x):
        return ops.conv(x, self.weight, self.bias, self._s)


def _check_conv(m, k, stride, padding):
    if k not in (1, 3) or stride not in (1, 2) or padding != (k - 1) // 2 or (k == 1 and stride != 1):
        raise NotImplementedError("HIP conv supports k in {1,3}, stride in {1,2}, 'same' zero padding "
                                  "(got k=%r stride=%r pad=%r)" % (k, stride, padding))


class _BatchNormMixin:
    fused_slope = 1.0   # set by a parent container to fold the following LeakyReLU into the BN-apply pass

    def forward(self, x):
        # the reference never leaves training mode: always batch statistics + running update
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked, self.fused_slope)


class BatchNorm3d(_BatchNormMixin, nn.BatchNorm3d):
    pass


class BatchNorm2d(_BatchNormMixin, nn.BatchNorm2d):
    pass


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(C) with its defaults (affine=False, no running statistics): for a single patch this is the
    train-mode BatchNorm kernel with unit scale / zero shift.  No parameters, no state_dict entries (like torch's)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.register_buffer("_one", torch.ones(num_features), persistent=False)
        self.register_buffer("_zero", torch.zeros(num_features), persistent=False)
        self.fused_slope = 1.0

    def forward(self, x):
        return ops.batch_norm(x, self._one, self._zero, None, None, None, self.fused_slope)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size=2, stride=2):
        super().__init__()
        if kernel_size != 2 or stride != 2:
            raise NotImplementedError("only MaxPool2d(2, 2)")

    def forward(self, x):
        return ops.max_pool2x2(x)


class ConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, in_f, out_f, kernel_size=4, stride=2, padding=1):
        super().__init__(in_f, out_f, kernel_size, stride=stride, padding=padding)
        if (kernel_size, stride, padding) != (4, 2, 1):
            raise NotImplementedError("only ConvTranspose2d(k=4, stride=2, padding=1)")

    def forward(self, x):
        return ops.conv_transpose4x4s2(x, self.weight, self.bias)


class LeakyReLU(nn.Module):
    """LeakyReLU(slope); slope=0 gives ReLU.  (Out of place; the reference's inplace=True is a memory detail.)"""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = float(negative_slope)

    def forward(self, x):
        return ops.leaky_relu(x, self.negative_slope)

    def extra_repr(self):
        return "negative_slope=%g" % self.negative_slope


class Activation(nn.Module):
    """nn.ELU() / nn.Tanh() / nn.Sigmoid() of the reference's get_activation, as stand-alone HIP passes (they are not
    the default and are not fused into their neighbours like LeakyReLU is)."""

    def __init__(self, name):
        super().__init__()
        if name not in ops.ACT_KINDS:
            raise NotImplementedError("unknown activation %r" % (name,))
        self.name = name

    def forward(self, x):
        return ops.activation(x, self.name)

    def extra_repr(self):
        return self.name


class Upsample(nn.Module):
    def __init__(self, scale_factor=2, mode="nearest"):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("only scale_factor=2")
        self.scale_factor, self.mode = scale_factor, mode

    def forward(self, x):
        return ops.upsample2x(x, self.mode)

    def extra_repr(self):
        return "scale_factor=2, mode=%s" % self.mode


class Dropout(nn.Module):
    """nn.Dropout2d / nn.Dropout3d(p) of the reference (mulresunet.py:24,83,152,227): channel dropout, always in training
    mode.  p = 0 (the reference default) is the identity; p > 0 draws its mask from torch's device generator — a different
    random stream than the reference's, like the input noise."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)
        if not 0.0 <= self.p < 1.0:
            raise ValueError("dropout probability must be in [0, 1)")

    def forward(self, x):
        if self.p == 0.0:
            return x
        return ops.channel_dropout(x, self.p)

    def extra_repr(self):
        return "p=%g" % self.p


class Seq(nn.Sequential):
    """nn.Sequential whose `.add(m)` names the child str(len+1) — the naming rule the reference installs on
    torch.nn.Module (architectures/base.py:69-73) and that its state_dict keys depend on."""

    def add(self, module):
        self.add_module(str(len(self) + 1), module)


class Concat(nn.Module):
    """Run every child on the same input, centre-crop to the smallest spatial size, cat on dim 1
    (reference Concat / Concat3D, base.py:289-362)."""

    def __init__(self, dim, *mods):
        super().__init__()
        assert dim == 1
        self.dim = dim
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        return ops.concat_crop([m(x) for m in self._modules.values()])

    def __len__(self):
        return len(self._modules)
