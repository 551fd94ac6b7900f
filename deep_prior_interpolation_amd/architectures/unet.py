"""Plain 2-D UNet (drop-in for reference architectures/unet.py:84-187): InstanceNorm, MaxPool down-sampling, and either a
ConvTranspose2d(4, 2, 1) ('deconv') or Upsample + 3x3 conv up path.

The reference class works but is unreachable through its own get_net (which names an undefined `UNetMod`,
architectures/__init__.py:13); here `--net unet` builds it.  Options outside the hot path (concat_x, more_layers,
reflection padding) are rejected loudly.
"""
import torch
from torch import nn

from .. import nn as hnn
from .. import ops
from .base import Seq, conv_nd, get_activation

__all__ = ["UNet"]


def _conv_block(in_f, out_f, norm, bias, act):
    """conv3x3 -> [norm] -> act as Sequential(conv Sequential, [norm], act)  (child names 0, 1, 2 / 0, 1)."""
    mods = [conv_nd(2, in_f, out_f, 3, bias=bias)]
    if norm:
        mods.append(hnn.InstanceNorm2d(out_f))
    mods.append(act)
    return nn.Sequential(*mods)


class unetConv(nn.Module):
    def __init__(self, in_size, out_size, norm, need_bias, act_fun, drop=0.0):
        super().__init__()
        self.conv1 = _conv_block(in_size, out_size, norm, need_bias, act_fun)
        self.conv2 = _conv_block(out_size, out_size, norm, need_bias, act_fun)
        self.dr = hnn.Dropout(drop)

    def forward(self, x):
        return self.dr(self.conv2(self.dr(self.conv1(x))))


class unetDown(nn.Module):
    def __init__(self, in_size, out_size, norm, need_bias, act_fun, drop=0.0):
        super().__init__()
        self.conv = unetConv(in_size, out_size, norm, need_bias, act_fun)
        self.down = hnn.MaxPool2d(2, 2)
        self.dr = hnn.Dropout(drop)

    def forward(self, x):
        return self.dr(self.conv(self.dr(self.down(x))))


class unetUp(nn.Module):
    def __init__(self, out_size, upsample_mode, need_bias, act_fun, drop=0.0, same_num_filt=False):
        super().__init__()
        num_filt = out_size if same_num_filt else out_size * 2
        if upsample_mode == "deconv":
            self.up = hnn.ConvTranspose2d(num_filt, out_size, 4, stride=2, padding=1)
        elif upsample_mode in ("bilinear", "nearest"):
            self.up = nn.Sequential(hnn.Upsample(scale_factor=2, mode=upsample_mode), conv_nd(2, num_filt, out_size, 3, bias=need_bias))
        else:
            raise NotImplementedError("UNet upsample_mode %r" % (upsample_mode,))
        self.conv = unetConv(out_size * 2, out_size, False, need_bias, act_fun, drop)
        self.dr = hnn.Dropout(drop)

    def forward(self, deep, skip):
        # the skip tensor is centre-cropped to the up-sampled one (unet.py:71-76); cat order [up, skip]
        return self.dr(self.conv(ops.concat_crop([self.up(deep), skip])))


class UNet(nn.Module):
    def __init__(self, num_input_channels=1, num_output_channels=1, filters=(16, 32, 64, 128, 256), more_layers=0, concat_x=False,
                 act_fun="ReLU", upsample_mode="deconv", pad="zero", dropout=0.0, norm_layer="instance", last_act_fun=None,
                 need_bias=True):
        super().__init__()
        if more_layers != 0 or concat_x or pad != "zero":
            raise NotImplementedError("UNet: more_layers / concat_x / non-zero padding are outside the HIP path")
        filters = list(filters)
        assert len(filters) == 5
        norm = norm_layer is not None
        act = get_activation(act_fun)                    # ONE shared module instance, like the reference (unet.py:99)
        self.start = unetConv(num_input_channels, filters[0], norm, need_bias, act, dropout)
        self.down1 = unetDown(filters[0], filters[1], norm, need_bias, act, dropout)
        self.down2 = unetDown(filters[1], filters[2], norm, need_bias, act, dropout)
        self.down3 = unetDown(filters[2], filters[3], norm, need_bias, act, dropout)
        self.down4 = unetDown(filters[3], filters[4], norm, need_bias, act, dropout)
        self.up4 = unetUp(filters[3], upsample_mode, need_bias, act, dropout)
        self.up3 = unetUp(filters[2], upsample_mode, need_bias, act, dropout)
        self.up2 = unetUp(filters[1], upsample_mode, need_bias, act, dropout)
        self.up1 = unetUp(filters[0], upsample_mode, need_bias, act, dropout)
        self.final = conv_nd(2, filters[0], num_output_channels, 1, bias=need_bias)
        if isinstance(last_act_fun, str) and last_act_fun.lower() == "none":
            last_act_fun = None
        if last_act_fun is not None:
            self.final = nn.Sequential(self.final, get_activation(last_act_fun))

    def forward(self, x):
        in64 = self.start(x)
        d1 = self.down1(in64)
        d2 = self.down2(d1)
        d3 = self.down3(d2)
        d4 = self.down4(d3)
        u4 = self.up4(d4, d3)
        u3 = self.up3(u4, d2)
        u2 = self.up2(u3, d1)
        u1 = self.up1(u2, in64)
        return self.final(u1)
