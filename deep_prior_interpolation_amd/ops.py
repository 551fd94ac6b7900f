"""torch.autograd.Function wrappers over the C ABI (leaf ops of the drop-in module tree).

Every op here runs a hand-written HIP kernel from libdpi_hip.so on the current HIP stream; there is no
aten / CPU fallback.  Tensors must be fp32, on a HIP device, batch size 1 (the reference optimises one
patch at a time, main.py:131-135), layout (1, C, [D,] H, W) contiguous.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, stream

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _req(t, name):
    if not t.is_cuda:
        raise _lib.DpiError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise _lib.DpiError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()


def _dims(x):
    """(C, D, H, W) of a (1,C,H,W) or (1,C,D,H,W) tensor."""
    if x.shape[0] != 1:
        raise _lib.DpiError("batch size must be 1 (got %d): BatchNorm statistics are per patch" % x.shape[0])
    if x.ndim == 5:
        return x.shape[1], x.shape[2], x.shape[3], x.shape[4]
    if x.ndim == 4:
        return x.shape[1], 1, x.shape[2], x.shape[3]
    raise _lib.DpiError("expected a 4-D or 5-D tensor, got shape %s" % (tuple(x.shape),))


def _like_spatial(x, C_, D, H, W):
    return (1, C_, D, H, W) if x.ndim == 5 else (1, C_, H, W)


def conv_out(n, k, s):
    return (n + 2 * ((k - 1) // 2) - k) // s + 1


def make_desc(x, w, stride):
    Cin, D, H, W = _dims(x)
    k = w.shape[-1]
    kd = w.shape[2] if w.ndim == 5 else 1
    if w.shape[1] != Cin:
        raise _lib.DpiError("conv: weight expects %d input channels, tensor has %d" % (w.shape[1], Cin))
    return ConvDesc(Cin, w.shape[0], D, H, W, k, kd, int(stride))


def desc_out_dims(d):
    sd = d.stride if d.kd > 1 else 1
    return conv_out(d.D, d.kd, sd), conv_out(d.H, d.k, d.stride), conv_out(d.W, d.k, d.stride)


# ------------------------------------------------------------------------------------------------
# raw (non-autograd) launches, shared with the fused engine
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Brackets selected launches with HIP events on the stream they are launched on (torch's current stream),
    for bench.py's roofline line.  `match(kind, desc)` selects launches; durations() gives milliseconds."""

    def __init__(self, match):
        self.match = match
        self.events = []

    def durations(self):
        return [a.elapsed_time(b) for a, b in self.events]


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def _timed(kind, d, launch):
    if _timer is not None and _timer.match(kind, d):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        _timer.events.append((e0, e1))
    else:
        launch()


def raw_conv_fwd(d, x, chain, w, bias, y, partials=None):
    L = _lib.load()
    _timed("conv_fwd", d, lambda: check(L.dpi_conv_fwd(C.byref(d), ptr(x), ptr(chain), ptr(w), ptr(bias), ptr(y),
                                                       ptr(partials), stream()), "dpi_conv_fwd"))


def raw_conv_bwd_data(d, dy, w, dx, accumulate=False):
    L = _lib.load()
    _timed("conv_bwd_data", d, lambda: check(L.dpi_conv_bwd_data(C.byref(d), ptr(dy), ptr(w), ptr(dx), int(accumulate),
                                                                 stream()), "dpi_conv_bwd_data"))


def raw_conv_bwd_weight(d, x, chain, dy, dw):
    L = _lib.load()
    n = L.dpi_conv_bwd_weight_ws_floats(C.byref(d))
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    _timed("conv_bwd_weight", d, lambda: check(L.dpi_conv_bwd_weight(C.byref(d), ptr(x), ptr(chain), ptr(dy), ptr(dw),
                                                                     ptr(ws), n, stream()), "dpi_conv_bwd_weight"))


def raw_channel_sum(x, C_, V, out):
    L = _lib.load()
    nblk = L.dpi_stat_blocks(C_, V)
    ws = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    check(L.dpi_channel_sum(ptr(x), C_, V, ptr(ws), ptr(out), stream()), "dpi_channel_sum")


def raw_bn_stats_finalize(x, chain_in, C_, V, gamma, beta, slope, running_mean, running_var, nbt, mean_invstd, chain_out,
                          eps=BN_EPS, momentum=BN_MOMENTUM, act_first=0):
    """channel stats of T(x) followed by finalize."""
    L = _lib.load()
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    check(L.dpi_channel_stats(ptr(x), ptr(chain_in), C_, V, ptr(part), stream()), "dpi_channel_stats")
    check(L.dpi_bn_finalize(ptr(part), nblk, C_, V, ptr(gamma), ptr(beta), eps, momentum, slope, act_first, ptr(running_mean),
                            ptr(running_var), ptr(nbt), ptr(mean_invstd), ptr(chain_out), stream()), "dpi_bn_finalize")


def raw_bn_finalize(part, nblk, C_, count, gamma, beta, slope, running_mean, running_var, nbt, mean_invstd, chain_out,
                    eps=BN_EPS, momentum=BN_MOMENTUM):
    L = _lib.load()
    check(L.dpi_bn_finalize(ptr(part), nblk, C_, count, ptr(gamma), ptr(beta), eps, momentum, slope, 0, ptr(running_mean),
                            ptr(running_var), ptr(nbt), ptr(mean_invstd), ptr(chain_out), stream()), "dpi_bn_finalize")


def raw_chain_apply(x, chain, C_, V, y):
    L = _lib.load()
    check(L.dpi_chain_apply(ptr(x), ptr(chain), C_, V, ptr(y), stream()), "dpi_chain_apply")


_slope_chains = {}


def slope_chain(C_, slope, device):
    """chain {1,0,slope,1,0} x C: a bare activation."""
    key = (C_, float(slope), str(device))
    t = _slope_chains.get(key)
    if t is None:
        t = torch.tensor([1.0, 0.0, slope, 1.0, 0.0], dtype=torch.float32).repeat(C_, 1).to(device).contiguous()
        _slope_chains[key] = t
    return t


# ------------------------------------------------------------------------------------------------
# autograd leaf ops
# ------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """nn.Conv3d / nn.Conv2d, k in {1,3}, stride in {1,2}, zero pad (k-1)//2 (reference base.py:123,176)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        x, w = _req(x, "conv input"), _req(w, "conv weight")
        b = _req(b, "conv bias") if b is not None else None
        d = make_desc(x, w, stride)
        Do, Ho, Wo = desc_out_dims(d)
        y = torch.empty(_like_spatial(x, d.Cout, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        raw_conv_fwd(d, x, None, w, b, y)
        ctx.save_for_backward(x, w)
        ctx.d = d
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d = ctx.d
        dy = _req(dy, "conv grad")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw_conv_bwd_data(d, dy, w, dx)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            raw_conv_bwd_weight(d, x, None, dy, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(d.Cout, dtype=torch.float32, device=x.device)
            raw_channel_sum(dy, d.Cout, dy.numel() // d.Cout, db)
        return dx, dw, db, None


def _bn_backward(dy, x, mi, gamma, beta, pre_slope, post_slope):
    """two-phase BatchNorm backward with the surrounding LeakyReLU folded in; returns (dx, dgamma, dbeta)."""
    L = _lib.load()
    C_ = x.shape[1]
    V = x.numel() // C_
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    check(L.dpi_bn_bwd_reduce(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), pre_slope, post_slope, C_, V, ptr(part), stream()),
          "dpi_bn_bwd_reduce")
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(gamma)
    check(L.dpi_bn_bwd_apply(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), pre_slope, post_slope, ptr(part), nblk, C_, V,
                             ptr(dx), ptr(dgamma), ptr(dbeta), stream()), "dpi_bn_bwd_apply")
    return dx, dgamma, dbeta


def _pre_chain(C_, pre_slope, device):
    return None if pre_slope == 1.0 else slope_chain(C_, pre_slope, device)


class BatchNormFn(torch.autograd.Function):
    """Train-mode BatchNorm over (D,H,W) of a single patch with optional fused LeakyReLU before (pre_slope: the
    act -> BN tail of Block3d / ResPath3d) and/or after (post_slope: conv -> BN -> act).  Only the BN input is saved:
    both activation masks are recomputed from it in the backward kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, post_slope, pre_slope=1.0):
        x = _req(x, "batchnorm input")
        C_ = x.shape[1]
        V = x.numel() // C_
        mi = torch.empty(2 * C_, dtype=torch.float32, device=x.device)
        chain = torch.empty(C_ * 5, dtype=torch.float32, device=x.device)
        if pre_slope != 1.0 and post_slope != 1.0:
            raise _lib.DpiError("batch_norm: pre and post activation cannot both be fused")
        # act -> BN runs as ONE chained pass: statistics of act(x), then T(x) = (gamma*invstd) * act(x) + shift
        raw_bn_stats_finalize(x, _pre_chain(C_, pre_slope, x.device), C_, V, gamma, beta,
                              pre_slope if pre_slope != 1.0 else post_slope, running_mean, running_var, nbt, mi, chain,
                              act_first=int(pre_slope != 1.0))
        y = torch.empty_like(x)
        raw_chain_apply(x, chain, C_, V, y)
        ctx.save_for_backward(x, gamma, beta, mi)
        ctx.slopes = (float(pre_slope), float(post_slope))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mi = ctx.saved_tensors
        dx, dgamma, dbeta = _bn_backward(_req(dy, "batchnorm grad"), x, mi, gamma, beta, *ctx.slopes)
        return dx, dgamma, dbeta, None, None, None, None, None


class ConvBnActFn(torch.autograd.Function):
    """conv -> BatchNorm(train) -> LeakyReLU(slope) in three launches: the conv's epilogue emits the {sum, sum^2}
    partials, dpi_bn_finalize turns them into the per-channel chain, dpi_chain_apply writes the activation.
    Backward: BN+act backward straight from (dy, raw conv output), then backward-data / backward-weight.
    The conv bias feeds a BatchNorm, so its gradient is analytically zero and returned as zeros."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, stride, slope):
        x, w = _req(x, "conv input"), _req(w, "conv weight")
        d = make_desc(x, w, stride)
        Do, Ho, Wo = desc_out_dims(d)
        L = _lib.load()
        r = torch.empty(_like_spatial(x, d.Cout, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
        part = torch.empty(nblk * d.Cout * 2, dtype=torch.float64, device=x.device)
        raw_conv_fwd(d, x, None, w, b, r, part)
        V = Do * Ho * Wo
        mi = torch.empty(2 * d.Cout, dtype=torch.float32, device=x.device)
        chain = torch.empty(d.Cout * 5, dtype=torch.float32, device=x.device)
        raw_bn_finalize(part, nblk, d.Cout, V, gamma, beta, slope, running_mean, running_var, nbt, mi, chain)
        y = torch.empty_like(r)
        raw_chain_apply(r, chain, d.Cout, V, y)
        ctx.save_for_backward(x, w, r, gamma, beta, mi)
        ctx.d, ctx.slope, ctx.has_bias = d, float(slope), b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, r, gamma, beta, mi = ctx.saved_tensors
        d = ctx.d
        dr, dgamma, dbeta = _bn_backward(_req(dy, "conv-bn-act grad"), r, mi, gamma, beta, 1.0, ctx.slope)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw_conv_bwd_data(d, dr, w, dx)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            raw_conv_bwd_weight(d, x, None, dr, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(d.Cout, dtype=torch.float32, device=x.device)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None


class LeakyReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = _req(x, "activation input")
        C_ = x.shape[1]
        y = torch.empty_like(x)
        raw_chain_apply(x, slope_chain(C_, slope, x.device), C_, x.numel() // C_, y)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _req(dy, "activation grad")
        dx = torch.empty_like(dy)
        L = _lib.load()
        check(L.dpi_lrelu_bwd(ptr(dy), ptr(y), ctx.slope, dy.numel(), ptr(dx), stream()), "dpi_lrelu_bwd")
        return dx, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a, "add lhs"), _req(b, "add rhs")
        if a.shape != b.shape:
            raise _lib.DpiError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        y = torch.empty_like(a)
        check(_lib.load().dpi_add(ptr(a), ptr(b), a.numel(), ptr(y), stream()), "dpi_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Upsample2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, nearest | bilinear | trilinear), optional crop of the output."""

    @staticmethod
    def forward(ctx, x, linear, out_size):
        x = _req(x, "upsample input")
        C_, D, H, W = _dims(x)
        if out_size is None:
            Do, Ho, Wo = (2 * D if x.ndim == 5 else 1), 2 * H, 2 * W
        else:
            Do, Ho, Wo = out_size
        y = torch.empty(_like_spatial(x, C_, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        check(_lib.load().dpi_upsample2x_fwd(ptr(x), None, C_, D, H, W, Do, Ho, Wo, int(linear), ptr(y), stream()),
              "dpi_upsample2x_fwd")
        ctx.geo = (C_, D, H, W, Do, Ho, Wo, int(linear))
        ctx.in_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "upsample grad")
        C_, D, H, W, Do, Ho, Wo, linear = ctx.geo
        dx = torch.empty(ctx.in_shape, dtype=torch.float32, device=dy.device)
        check(_lib.load().dpi_upsample2x_bwd(ptr(dy), C_, D, H, W, Do, Ho, Wo, linear, ptr(dx), stream()), "dpi_upsample2x_bwd")
        return dx, None, None


class ConcatCropFn(torch.autograd.Function):
    """Concat / Concat3D (reference base.py:289-362): centre-crop to the smallest spatial size, cat on dim 1."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_req(x, "concat input") for x in xs]
        dims = [_dims(x) for x in xs]
        tD, tH, tW = (min(d[i] for d in dims) for i in (1, 2, 3))
        Ct = sum(d[0] for d in dims)
        y = torch.empty(_like_spatial(xs[0], Ct, tD, tH, tW), dtype=torch.float32, device=xs[0].device)
        L = _lib.load()
        c0 = 0
        geo = []
        for x, (C_, D, H, W) in zip(xs, dims):
            od, oh, ow = (D - tD) // 2, (H - tH) // 2, (W - tW) // 2
            check(L.dpi_crop_copy(ptr(x), C_, D, H, W, od, oh, ow, tD, tH, tW, ptr(y[:, c0:c0 + C_]), stream()), "dpi_crop_copy")
            geo.append((c0, C_, D, H, W, od, oh, ow, x.shape))
            c0 += C_
        ctx.geo = geo
        ctx.t = (tD, tH, tW)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "concat grad")
        tD, tH, tW = ctx.t
        L = _lib.load()
        outs = []
        for (c0, C_, D, H, W, od, oh, ow, shape) in ctx.geo:
            g = dy[:, c0:c0 + C_]
            if (D, H, W) == (tD, tH, tW):
                outs.append(g)              # contiguous slice for N == 1
            else:
                dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
                check(L.dpi_crop_copy_bwd(ptr(g), C_, D, H, W, od, oh, ow, tD, tH, tW, ptr(dx), stream()), "dpi_crop_copy_bwd")
                outs.append(dx)
        return tuple(outs)


class MaskedLossFn(torch.autograd.Function):
    """loss_fn(out*mask, img*mask) of main.py:161 with the SNR / PCORR sums of main.py:166-167 in the same pass.
    Returns (loss, metrics) with metrics = device double[8] {loss, snr_dB, pcorr, ...}."""

    @staticmethod
    def forward(ctx, out, img, mask, kind):
        out, img, mask = _req(out, "loss output"), _req(img, "loss target"), _req(mask, "loss mask")
        if not (out.shape == img.shape == mask.shape):
            raise _lib.DpiError("loss: shape mismatch")
        L = _lib.load()
        n = out.numel()
        ws = torch.empty(L.dpi_loss_ws_doubles(n), dtype=torch.float64, device=out.device)
        res = torch.empty(8, dtype=torch.float64, device=out.device)
        dout = torch.empty_like(out)
        check(L.dpi_masked_loss(ptr(out), ptr(img), ptr(mask), n, int(kind), 1.0, ptr(dout), ptr(ws), ptr(res), stream()),
              "dpi_masked_loss")
        ctx.save_for_backward(dout)
        ctx.mark_non_differentiable(res)
        return res[0].to(torch.float32), res

    @staticmethod
    def backward(ctx, gloss, _gres):
        (dout,) = ctx.saved_tensors
        return dout * gloss, None, None, None


class MaxPool2x2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) on (1,C,H,W) (reference unet.py:42)."""

    @staticmethod
    def forward(ctx, x):
        x = _req(x, "maxpool input")
        if x.ndim != 4 or x.shape[0] != 1:
            raise _lib.DpiError("maxpool: expected (1,C,H,W)")
        C_, H, W = x.shape[1:]
        y = torch.empty((1, C_, H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(_lib.load().dpi_maxpool2x2_fwd(ptr(x), C_, H, W, ptr(y), stream()), "dpi_maxpool2x2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        C_, H, W = x.shape[1:]
        dx = torch.empty_like(x)
        check(_lib.load().dpi_maxpool2x2_bwd(ptr(_req(dy, "maxpool grad")), ptr(x), C_, H, W, ptr(dx), stream()), "dpi_maxpool2x2_bwd")
        return dx


class Deconv4x4s2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1) (reference unet.py:59); weight [Cin][Cout][4][4]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _req(x, "deconv input"), _req(w, "deconv weight")
        if x.ndim != 4 or x.shape[0] != 1 or tuple(w.shape[2:]) != (4, 4) or w.shape[0] != x.shape[1]:
            raise _lib.DpiError("deconv: expected (1,Cin,H,W) input and [Cin][Cout][4][4] weight")
        Cin, H, W = x.shape[1:]
        Cout = w.shape[1]
        y = torch.empty((1, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        check(_lib.load().dpi_deconv4x4s2_fwd(ptr(x), ptr(w), ptr(b), Cin, Cout, H, W, ptr(y), stream()), "dpi_deconv4x4s2_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _req(dy, "deconv grad")
        Cin, H, W = x.shape[1:]
        Cout = w.shape[1]
        L = _lib.load()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(L.dpi_deconv4x4s2_bwd_data(ptr(dy), ptr(w), Cin, Cout, H, W, ptr(dx), stream()), "dpi_deconv4x4s2_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            check(L.dpi_deconv4x4s2_bwd_weight(ptr(x), ptr(dy), Cin, Cout, H, W, ptr(dw), stream()), "dpi_deconv4x4s2_bwd_weight")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cout, dtype=torch.float32, device=x.device)
            raw_channel_sum(dy, Cout, dy.numel() // Cout, db)
        return dx, dw, db


def conv(x, w, b, stride=1):
    return ConvFn.apply(x, w, b, stride)


def max_pool2x2(x):
    return MaxPool2x2Fn.apply(x)


def conv_transpose4x4s2(x, w, b):
    return Deconv4x4s2Fn.apply(x, w, b)


def batch_norm(x, gamma, beta, running_mean=None, running_var=None, nbt=None, slope=1.0, pre_slope=1.0):
    return BatchNormFn.apply(x, gamma, beta, running_mean, running_var, nbt, float(slope), float(pre_slope))


def conv_bn_act(x, w, b, gamma, beta, running_mean, running_var, nbt, stride=1, slope=0.2):
    return ConvBnActFn.apply(x, w, b, gamma, beta, running_mean, running_var, nbt, int(stride), float(slope))


def leaky_relu(x, slope=0.2):
    return LeakyReLUFn.apply(x, float(slope))


def add(a, b):
    return AddFn.apply(a, b)


def upsample2x(x, mode="nearest", out_size=None):
    if mode not in ("nearest", "bilinear", "trilinear", "linear"):
        raise NotImplementedError("upsample mode %r" % mode)
    return Upsample2xFn.apply(x, mode != "nearest", out_size)


def concat_crop(xs):
    return ConcatCropFn.apply(*xs)


def masked_loss(out, img, mask, kind="mae"):
    return MaskedLossFn.apply(out, img, mask, 1 if kind == "mse" else 0)
