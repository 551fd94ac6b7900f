"""MultiRes-UNet, 2-D and 3-D (drop-in for reference architectures/mulresunet.py).

One implementation parametrised by `nd`; the structural differences between the reference's 2-D and
3-D variants are kept: the 3-D block has bn1/bn2 and a BatchNorm after the stride-2
encoder conv, the 2-D one has neither; the 3-D output conv is 3x3x3, the 2-D one 1x1.
Child registration and construction order follow the reference (cited inline) so that state_dict keys
and same-seed initial values coincide.
"""
from torch import nn

from .. import nn as hnn
from .. import ops
from .base import Concat, Seq, conv_bn_act, conv_nd, get_activation

__all__ = ["MulResUnet", "MulResUnet3D", "MultiResBlock", "ResPath", "multires_widths"]


FUSE_BLOCKS = True     # run Block3d / ResPath3d as single fused autograd nodes (ops.Block3dFn / ops.ResPath3dFn)


def multires_widths(U, alpha=1.67):
    """Channel split of a MultiRes block (mulresunet.py:14-23, 70-78)."""
    W = alpha * U
    return int(W * 0.167), int(W * 0.333), int(W * 0.5)


class MultiResBlock(nn.Module):
    """Block3d (mulresunet.py:67-96) / Block2d (11-36):
       o1 = CBA3(x); o2 = CBA3(o1); o3 = CBA3(o2)
       3-D: y = bn2(act(CBA1(x) + bn1(cat[o1,o2,o3])))      2-D: y = act(CBA1(x) + cat[o1,o2,o3])"""

    def __init__(self, nd, U, f_in, alpha=1.67, act_fun="LeakyReLU", bias=True, drop=0.0):
        super().__init__()
        a, b, c = multires_widths(U, alpha)
        self.nd = nd
        self.out_dim = a + b + c
        self.shortcut = conv_bn_act(nd, f_in, self.out_dim, 1, 1, bias=bias, act_fun=act_fun)
        self.conv3x3 = conv_bn_act(nd, f_in, a, 3, 1, bias=bias, act_fun=act_fun)
        self.conv5x5 = conv_bn_act(nd, a, b, 3, 1, bias=bias, act_fun=act_fun)
        self.conv7x7 = conv_bn_act(nd, b, c, 3, 1, bias=bias, act_fun=act_fun)
        if nd == 3:
            self.bn1 = hnn.BatchNorm3d(self.out_dim)
            self.bn2 = hnn.BatchNorm3d(self.out_dim)
            self.act = get_activation(act_fun)
            self.dr = hnn.Dropout(drop)
        else:
            self.dr = hnn.Dropout(drop)
            self.act = get_activation(act_fun)

    def _fusable(self):
        return (self.nd == 3 and isinstance(self.act, hnn.LeakyReLU) and self.dr.p == 0.0
                and all(isinstance(m._parts()[2], hnn.LeakyReLU) and m._parts()[2].negative_slope == self.act.negative_slope
                        for m in (self.conv3x3, self.conv5x5, self.conv7x7, self.shortcut)))

    def forward(self, x):
        if FUSE_BLOCKS and self._fusable():
            return ops.block3d(x, self, self.act.negative_slope)
        o1 = self.conv3x3(x)
        o2 = self.conv5x5(o1)
        o3 = self.conv7x7(o2)
        out = ops.concat_crop([o1, o2, o3])
        if self.nd == 3:
            out = self.dr(self.bn1(out))
        else:
            out = self.dr(out)
        out = ops.add(self.shortcut(x), out)
        if self.nd == 3 and isinstance(self.act, hnn.LeakyReLU):
            b = self.bn2                                    # act -> bn2 as one chained pass
            out = ops.batch_norm(out, b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked, 1.0,
                                 self.act.negative_slope)
        else:
            out = self.act(out)
            if self.nd == 3:
                out = self.bn2(out)
        return self.dr(out)


class ResPath(nn.Module):
    """ResPath3d (mulresunet.py:99-113): bn(act(CBA1(x) + CBA3(x))).
    ResPath2d (39-64, length 1 — the only length the reference instantiates): same maths, children live in `net`."""

    def __init__(self, nd, f_in, f_out, act_fun="LeakyReLU", bias=True, drop=0.0):
        super().__init__()
        self.nd = nd
        if nd == 3:
            self.conv3x3 = conv_bn_act(3, f_in, f_out, 3, 1, bias=bias, act_fun=act_fun)
            self.conv1x1 = conv_bn_act(3, f_in, f_out, 1, 1, bias=bias, act_fun=act_fun)
            self.bn = hnn.BatchNorm3d(f_out)
            self.act = get_activation(act_fun)
            self.dr = hnn.Dropout(drop)
        else:
            self.dr = hnn.Dropout(drop)
            c3 = conv_bn_act(2, f_in, f_out, 3, 1, bias=bias, act_fun=act_fun)
            c1 = conv_bn_act(2, f_in, f_out, 1, 1, bias=bias, act_fun=act_fun)
            bn = hnn.BatchNorm2d(f_out)
            self.act = get_activation(act_fun)
            self.length = 1
            self.net = nn.Sequential(c3, c1, bn, self.dr)

    def _fusable(self):
        return (self.nd == 3 and isinstance(self.act, hnn.LeakyReLU) and self.dr.p == 0.0
                and all(isinstance(m._parts()[2], hnn.LeakyReLU) and m._parts()[2].negative_slope == self.act.negative_slope
                        for m in (self.conv3x3, self.conv1x1)))

    def forward(self, x):
        if FUSE_BLOCKS and self._fusable():
            return ops.respath3d(x, self, self.act.negative_slope)
        if self.nd == 3:
            t, b = ops.add(self.conv1x1(x), self.conv3x3(x)), self.bn
        else:
            t, b = ops.add(self.net[0](x), self.net[1](x)), self.net[2]
        fuse_act_bn = isinstance(self.act, hnn.LeakyReLU) and (self.nd == 3 or self.dr.p == 0.0)
        if fuse_act_bn:                                         # act -> bn as one chained pass
            t = ops.batch_norm(t, b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked, 1.0,
                               self.act.negative_slope)
            return self.dr(t) if self.nd == 3 else t            # 3-D: add -> act -> bn -> dropout (mulresunet.py:109-112)
        if self.nd == 3:
            return self.dr(b(self.act(t)))
        return b(self.dr(self.act(t)))                          # 2-D: add -> act -> dropout -> bn (mulresunet.py:59-64)


class SkipConcat(Concat):
    """Concat(1, skip, deeper) of one U-Net level.  When the skip branch is a fusable ResPath3d and the deeper branch ends
    in an Upsample, both write directly into the concatenated tensor (ops.SkipJoinFn); otherwise plain Concat."""

    def forward(self, x):
        skip, deeper = list(self._modules.values())
        if FUSE_BLOCKS and isinstance(deeper, DownPath) and len(skip) == 1:
            rp = skip[0]
            mods = list(deeper._modules.values())
            if isinstance(rp, ResPath) and rp._fusable() and isinstance(mods[-1], hnn.Upsample):
                # the ResPath half only needs x: started on the branch stream before the deeper U runs (ops.skip_begin; None = serial)
                pre = ops.skip_begin(x, rp, rp.act.negative_slope, _deep_channels(mods[:-1])) if x.ndim == 5 else None
                # the two gradients of x (ResPath, stride-2 layer of the deeper U) meet inside the stride-2 layer's backward-data (ops.FanIn)
                fan = ops.FanIn() if (ops.FAN_IN and x.ndim == 5 and x.requires_grad) else None
                xs = ops.skip_tap(x) if (pre is not None or fan is not None) else x
                deep = deeper(x, stop_before_last=True, fan=fan)
                if deep.ndim == 5:
                    return ops.skip_join(xs, deep, rp, rp.act.negative_slope, mods[-1].mode, pre, fan)
                return ops.concat_crop([rp(x), mods[-1](deep)])
        return super().forward(x)


def _deep_channels(mods):
    """Channels the deeper branch hands to its Upsample: the last MultiRes block it runs (its own, or the decoder block of the level below)."""
    for m in reversed(mods):
        if isinstance(m, MultiResBlock):
            return m.out_dim
        if isinstance(m, nn.Sequential) and len(m) and isinstance(list(m._modules.values())[-1], MultiResBlock):
            return list(m._modules.values())[-1].out_dim
    raise ValueError("deeper branch without a MultiRes block")


class DownPath(Seq):
    """The `deeper` branch: stride-2 conv [-> BN] -> act -> dropout -> block -> [inner] -> upsample.  Same children and
    names as a plain Seq; only the conv -> BN -> LeakyReLU head (3-D) is executed as one fused op."""

    def forward(self, x, stop_before_last=False, fan=None):
        mods = list(self._modules.values())
        return self._run(mods[:-1] if stop_before_last else mods, x, fan)

    def _run(self, mods, x, fan=None):
        if (len(mods) >= 3 and isinstance(mods[1], (hnn.BatchNorm3d, hnn.BatchNorm2d)) and isinstance(mods[2], hnn.LeakyReLU)
                and isinstance(mods[0], nn.Sequential)):
            conv_m, bn, act = mods[0][0], mods[1], mods[2]
            x = ops.conv_bn_act(x, conv_m.weight, conv_m.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, conv_m._s, act.negative_slope, fan)
            mods = mods[3:]
        for m in mods:
            x = m(x)
        return x


def _mulresunet(nd, num_input_channels, num_output_channels, num_channels_down, num_channels_up, num_channels_skip,
                alpha, last_act_fun, need_bias, upsample_mode, act_fun, dropout):
    assert len(num_channels_down) == len(num_channels_up) == (len(num_channels_skip) + 1)
    n_scales = len(num_channels_down)
    if not isinstance(upsample_mode, (list, tuple)):
        upsample_mode = [upsample_mode] * n_scales

    model = Seq()
    cur = model
    block = MultiResBlock(nd, num_channels_down[0], num_input_channels, alpha, act_fun, need_bias, dropout)
    cur.add(block)
    depth = block.out_dim
    for i in range(1, n_scales):
        deeper, skip = DownPath(), Seq()
        # the encoder block is constructed before the stride-2 conv (mulresunet.py:221-224): RNG order
        block = MultiResBlock(nd, num_channels_down[i], depth, alpha, act_fun, need_bias, dropout)
        deeper.add(conv_nd(nd, depth, depth, 3, stride=2, bias=need_bias))
        if nd == 3:
            deeper.add(hnn.BatchNorm3d(depth))          # only the 3-D net normalises here (mulresunet.py:225 vs 150-152)
        deeper.add(get_activation(act_fun))
        deeper.add(hnn.Dropout(dropout))
        deeper.add(block)
        if num_channels_skip[i - 1] != 0:
            skip.add(ResPath(nd, depth, num_channels_skip[i - 1], act_fun, need_bias, dropout))
            cur.add(SkipConcat(1, skip, deeper))
        else:
            cur.add(deeper)
        inner = Seq()
        if i != n_scales - 1:
            deeper.add(inner)
        deeper.add(hnn.Upsample(scale_factor=2, mode=upsample_mode[i]))
        cur.add(MultiResBlock(nd, num_channels_up[i - 1], block.out_dim + num_channels_skip[i - 1], alpha, act_fun,
                              need_bias, dropout))
        depth = block.out_dim
        cur = inner
    last = sum(multires_widths(num_channels_up[0], alpha))
    model.add(conv_nd(nd, last, num_output_channels, 3 if nd == 3 else 1, bias=need_bias))
    if isinstance(last_act_fun, str) and last_act_fun.lower() == "none":
        last_act_fun = None
    if last_act_fun is not None:
        model.add(get_activation(last_act_fun))
    return model


def MulResUnet(num_input_channels=1, num_output_channels=1, num_channels_down=(16, 32, 64, 128, 256),
               num_channels_up=(16, 32, 64, 128, 256), num_channels_skip=(16, 32, 64, 128), alpha=1.67,
               last_act_fun=None, need_bias=True, upsample_mode="nearest", act_fun="LeakyReLU", dropout=0.0):
    """2-D MultiRes-UNet (reference mulresunet.py:116-185)."""
    return _mulresunet(2, num_input_channels, num_output_channels, list(num_channels_down), list(num_channels_up),
                       list(num_channels_skip), alpha, last_act_fun, need_bias, upsample_mode, act_fun, dropout)


def MulResUnet3D(num_input_channels=1, num_output_channels=1, num_channels_down=(16, 32, 64, 128, 256),
                 num_channels_up=(16, 32, 64, 128, 256), num_channels_skip=(16, 32, 64, 128), alpha=1.67,
                 last_act_fun=None, need_bias=True, upsample_mode="nearest", act_fun="LeakyReLU", dropout=0.0):
    """3-D MultiRes-UNet (reference mulresunet.py:188-259)."""
    return _mulresunet(3, num_input_channels, num_output_channels, list(num_channels_down), list(num_channels_up),
                       list(num_channels_skip), alpha, last_act_fun, need_bias, upsample_mode, act_fun, dropout)
