"""DIP "skip" hourglass, 2-D and 3-D (drop-in for reference architectures/skip.py).

Only the configuration reachable through get_net is built natively: zero padding, stride-2
down-sampling, need1x1_up=True (skip.py:51-66 defaults).  Per scale i:
    Concat( skip: conv1 -> BN -> act  ||  deep: conv3 s2 -> BN -> act -> conv3 -> BN -> act -> [next scale] -> Upsample )
    -> BN -> conv3 -> BN -> act -> conv1 -> BN -> act
followed by a final 1x1 conv.  Child names follow the reference's `.add()` numbering.
"""
from torch import nn

from .. import nn as hnn
from .base import Concat, Seq, conv_nd, get_activation

__all__ = ["Skip", "Skip3D"]


def _bn(nd, c):
    return hnn.BatchNorm3d(c) if nd == 3 else hnn.BatchNorm2d(c)


def _build_skip(nd, num_input_channels, num_output_channels, num_channels_down, num_channels_up, num_channels_skip,
                filter_size_down=3, filter_size_up=3, filter_skip_size=1, last_act_fun=None, need_bias=True,
                upsample_mode="nearest", act_fun="LeakyReLU", need1x1_up=True, dropout=0.0):
    assert len(num_channels_down) == len(num_channels_up) == len(num_channels_skip)
    n_scales = len(num_channels_down)
    if not isinstance(upsample_mode, (list, tuple)):
        upsample_mode = [upsample_mode] * n_scales
    last = n_scales - 1
    model = Seq()
    cur = model
    depth = num_input_channels
    for i in range(n_scales):
        deeper, skip = Seq(), Seq()
        k = num_channels_up[i + 1] if i < last else num_channels_down[i]
        cur.add(Concat(1, skip, deeper) if num_channels_skip[i] != 0 else deeper)
        cur.add(_bn(nd, num_channels_skip[i] + k))
        if num_channels_skip[i] != 0:
            skip.add(conv_nd(nd, depth, num_channels_skip[i], filter_skip_size, bias=need_bias))
            skip.add(_bn(nd, num_channels_skip[i]))
            skip.add(get_activation(act_fun))
            skip.add(hnn.Dropout(dropout))
        deeper.add(conv_nd(nd, depth, num_channels_down[i], filter_size_down, 2, bias=need_bias))
        deeper.add(_bn(nd, num_channels_down[i]))
        deeper.add(get_activation(act_fun))
        deeper.add(hnn.Dropout(dropout))
        deeper.add(conv_nd(nd, num_channels_down[i], num_channels_down[i], filter_size_down, bias=need_bias))
        deeper.add(_bn(nd, num_channels_down[i]))
        deeper.add(get_activation(act_fun))
        deeper.add(hnn.Dropout(dropout))
        inner = Seq()
        if i != last:
            deeper.add(inner)
        deeper.add(hnn.Upsample(scale_factor=2, mode=upsample_mode[i]))
        cur.add(conv_nd(nd, num_channels_skip[i] + k, num_channels_up[i], filter_size_up, 1, bias=need_bias))
        cur.add(_bn(nd, num_channels_up[i]))
        cur.add(get_activation(act_fun))
        cur.add(hnn.Dropout(dropout))
        if need1x1_up:
            cur.add(conv_nd(nd, num_channels_up[i], num_channels_up[i], 1, bias=need_bias))
            cur.add(_bn(nd, num_channels_up[i]))
            cur.add(get_activation(act_fun))
            cur.add(hnn.Dropout(dropout))
        depth = num_channels_down[i]
        cur = inner
    model.add(conv_nd(nd, num_channels_up[0], num_output_channels, 1, bias=need_bias))
    if isinstance(last_act_fun, str) and last_act_fun.lower() == "none":
        last_act_fun = None
    if last_act_fun is not None:
        model.add(get_activation(last_act_fun))
    return model


def Skip3D(num_input_channels=2, num_output_channels=3, num_channels_down=(16, 32, 64, 128, 128),
           num_channels_up=(16, 32, 64, 128, 128), num_channels_skip=(4, 4, 4, 4, 4), last_act_fun=None, need_bias=True,
           upsample_mode="nearest", act_fun="LeakyReLU", need1x1_up=True, dropout=0.0, **unsupported):
    """reference skip.py:154-254."""
    _reject(unsupported)
    return _build_skip(3, num_input_channels, num_output_channels, list(num_channels_down), list(num_channels_up),
                       list(num_channels_skip), last_act_fun=last_act_fun, need_bias=need_bias, upsample_mode=upsample_mode,
                       act_fun=act_fun, need1x1_up=need1x1_up, dropout=dropout)


class Skip(nn.Module):
    """2-D variant (reference skip.py:5-48): the hourglass lives under `.model`."""

    def __init__(self, num_input_channels=2, num_output_channels=3, num_channels_down=(16, 32, 64, 128, 128),
                 num_channels_up=(16, 32, 64, 128, 128), num_channels_skip=(4, 4, 4, 4, 4), last_act_fun=None, need_bias=True,
                 upsample_mode="nearest", act_fun="LeakyReLU", need1x1_up=True, dropout=0.0, **unsupported):
        super().__init__()
        _reject(unsupported)
        self.model = _build_skip(2, num_input_channels, num_output_channels, list(num_channels_down), list(num_channels_up),
                                 list(num_channels_skip), last_act_fun=last_act_fun, need_bias=need_bias,
                                 upsample_mode=upsample_mode, act_fun=act_fun, need1x1_up=need1x1_up, dropout=dropout)

    def forward(self, x):
        return self.model(x)


def _reject(kw):
    defaults = {"filter_size_down": 3, "filter_size_up": 3, "filter_skip_size": 1, "pad": "zero", "downsample_mode": "stride"}
    for k, v in kw.items():
        if k not in defaults or defaults[k] != v:
            raise NotImplementedError("Skip: option %s=%r is outside the HIP path (only %r)" % (k, v, defaults))
