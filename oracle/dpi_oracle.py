"""CPU restatement of the reference's deep-prior hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Everything here is written from the reference's behaviour, as plain functions over a flat
`{state_dict key: tensor}` mapping — no nn.Module tree.  Arithmetic is torch-CPU (fp32 by default,
fp64 on request); gradients of the restated forward come from torch autograd on CPU.

Reference anchors (paths relative to /root/reference):
  conv+BN+act factories ........ architectures/base.py:117-126,162-166,169-180,211-216
  Block3d / Block2d ............ architectures/mulresunet.py:67-96 / 11-36
  ResPath3d / ResPath2d ........ architectures/mulresunet.py:99-113 / 39-64
  MulResUnet3D / MulResUnet .... architectures/mulresunet.py:188-259 / 116-185
  Skip3D / _build_skip ......... architectures/skip.py:154-254 / 51-151
  Concat3D / Concat ............ architectures/base.py:325-362 / 289-322
  init_weights ................. utils/torch.py:23-58
  masked loss .................. main.py:24-27,161
  snr / pcorr .................. utils/metrics.py:6-17,20-44
  optimize loop ................ main.py:141-220
  EarlyStopping ................ utils/torch.py:216-275
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.2


# --------------------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------------------
def activation(name, x):
    """architectures/base.py:97-114 (LeakyReLU slope 0.2)."""
    if name == "LeakyReLU":
        return torch.where(x >= 0, x, x * LRELU_SLOPE)
    if name == "ReLU":
        return torch.clamp_min(x, 0)
    if name == "ELU":
        return F.elu(x)
    if name == "Tanh":
        return torch.tanh(x)
    if name == "Sigmoid":
        return torch.sigmoid(x)
    if name in ("none", None):
        return x
    raise NotImplementedError(name)


def conv_nd(x, w, b, stride=1):
    """Zero-padded 'same' convolution, pad = (k-1)//2 (base.py:121,174)."""
    k = w.shape[-1]
    pad = int((k - 1) / 2)
    if w.ndim == 5:
        return F.conv3d(x, w, b, stride=stride, padding=pad)
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def batch_norm_train(x, gamma, beta, running_mean=None, running_var=None, nbt=None):
    """Train-mode BatchNorm (the reference never calls .eval()): biased batch variance for the
    normalisation, unbiased for the running estimate, momentum 0.1, eps 1e-5."""
    dims = [0] + list(range(2, x.ndim))
    n = x.numel() // x.shape[1]
    mean = x.mean(dims)
    var = ((x - _bc(mean, x)) ** 2).mean(dims)
    if running_mean is not None:
        with torch.no_grad():
            running_mean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(running_mean.dtype))
            unb = var.detach() * (n / max(n - 1, 1))
            running_var.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb.to(running_var.dtype))
            if nbt is not None:
                nbt.add_(1)
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    return (x - _bc(mean, x)) * _bc(invstd * gamma, x) + _bc(beta, x)


def _bc(v, like):
    return v.reshape((1, -1) + (1,) * (like.ndim - 2))


def _up2_linear_axis(x, axis):
    """x2 linear interpolation, align_corners=False, along one axis:
    out[2i] = .25 x[i-1] + .75 x[i];  out[2i+1] = .75 x[i] + .25 x[i+1]  (edge-clamped)."""
    n = x.shape[axis]
    idx = torch.arange(n, device=x.device)
    lo = x.index_select(axis, torch.clamp(idx - 1, min=0))
    hi = x.index_select(axis, torch.clamp(idx + 1, max=n - 1))
    even = 0.25 * lo + 0.75 * x
    odd = 0.75 * x + 0.25 * hi
    out = torch.stack([even, odd], dim=axis + 1)
    shp = list(x.shape)
    shp[axis] = 2 * n
    return out.reshape(shp)


def upsample2x(x, mode):
    """nn.Upsample(scale_factor=2, mode) (mulresunet.py:168,242; skip.py:128,231)."""
    if mode == "nearest":
        for ax in range(2, x.ndim):
            x = x.repeat_interleave(2, dim=ax)
        return x
    if mode in ("trilinear", "bilinear", "linear"):
        # torch accumulates the 2^d corners in one expression; separable passes differ by rounding only.
        for ax in range(x.ndim - 1, 1, -1):
            x = _up2_linear_axis(x, ax)
        return x
    raise NotImplementedError(mode)


def concat_crop(tensors):
    """Concat/Concat3D: centre-crop every input to the minimum spatial size, cat on dim 1."""
    nsp = tensors[0].ndim - 2
    tgt = [min(t.shape[2 + a] for t in tensors) for a in range(nsp)]
    out = []
    for t in tensors:
        sl = [slice(None), slice(None)]
        for a in range(nsp):
            d = (t.shape[2 + a] - tgt[a]) // 2
            sl.append(slice(d, d + tgt[a]))
        out.append(t[tuple(sl)])
    return torch.cat(out, dim=1)


# --------------------------------------------------------------------------------------
# network restatements over a flat parameter mapping
# --------------------------------------------------------------------------------------
class NetState:
    """Flat parameter/buffer store keyed exactly like the reference's state_dict."""

    def __init__(self, state_dict, dtype=torch.float32, requires_grad=True):
        self.P = {}
        self.B = {}
        for k, v in state_dict.items():
            t = torch.from_numpy(np.array(v)) if not torch.is_tensor(v) else v.detach().clone()
            if k.endswith("running_mean") or k.endswith("running_var"):
                self.B[k] = t.to(dtype)
            elif k.endswith("num_batches_tracked"):
                self.B[k] = t.to(torch.int64)
            else:
                self.P[k] = t.to(dtype).requires_grad_(requires_grad)
        self.track_running = True
        self.used = set()

    def state_dict(self):
        out = {k: v.detach().clone() for k, v in self.P.items()}
        out.update({k: v.clone() for k, v in self.B.items()})
        return out

    def params(self):
        return list(self.P.values())

    # leaf helpers ------------------------------------------------------------------
    def conv(self, key, x, stride=1):
        self.used.update((key + ".weight", key + ".bias"))
        return conv_nd(x, self.P[key + ".weight"], self.P.get(key + ".bias"), stride)

    def bn(self, key, x):
        self.used.update((key + ".weight", key + ".bias"))
        if self.track_running:
            return batch_norm_train(x, self.P[key + ".weight"], self.P[key + ".bias"],
                                    self.B[key + ".running_mean"], self.B[key + ".running_var"],
                                    self.B.get(key + ".num_batches_tracked"))
        return batch_norm_train(x, self.P[key + ".weight"], self.P[key + ".bias"])


def multires_widths(U, alpha=1.67):
    """mulresunet.py:14-23 / 70-78."""
    W = alpha * U
    return int(W * 0.167), int(W * 0.333), int(W * 0.5)


def _cba3(S, pre, x, act, stride=1):
    """conv3dbn (base.py:211-216): Sequential[ Sequential[Conv3d] , BN , act ] -> keys .0.0 / .1"""
    return activation(act, S.bn(pre + ".1", S.conv(pre + ".0.0", x, stride)))


def _cba2(S, pre, x, act, stride=1):
    """conv2dbn (base.py:162-166): conv() Sequential then .add(BN) .add(act) -> keys .0 / .2"""
    return activation(act, S.bn(pre + ".2", S.conv(pre + ".0", x, stride)))


def block3d(S, pre, x, act):
    o1 = _cba3(S, pre + ".conv3x3", x, act)
    o2 = _cba3(S, pre + ".conv5x5", o1, act)
    o3 = _cba3(S, pre + ".conv7x7", o2, act)
    out = S.bn(pre + ".bn1", torch.cat([o1, o2, o3], dim=1))
    out = _cba3(S, pre + ".shortcut", x, act) + out
    out = activation(act, out)
    return S.bn(pre + ".bn2", out)


def block2d(S, pre, x, act):
    o1 = _cba2(S, pre + ".conv3x3", x, act)
    o2 = _cba2(S, pre + ".conv5x5", o1, act)
    o3 = _cba2(S, pre + ".conv7x7", o2, act)
    out = _cba2(S, pre + ".shortcut", x, act) + torch.cat([o1, o2, o3], dim=1)
    return activation(act, out)


def respath3d(S, pre, x, act):
    out = _cba3(S, pre + ".conv1x1", x, act) + _cba3(S, pre + ".conv3x3", x, act)
    return S.bn(pre + ".bn", activation(act, out))


def respath2d(S, pre, x, act):
    """ResPath2d with length=1 (the only length the reference builds, mulresunet.py:157)."""
    out = _cba2(S, pre + ".net.0", x, act) + _cba2(S, pre + ".net.1", x, act)
    return S.bn(pre + ".net.2", activation(act, out))


def mulresunet_forward(S, x, cfg, taps=None):
    """MulResUnet3D (cfg['ndim']==3) or MulResUnet (2).  cfg keys: ndim, filters, skip, upsample,
    act, last_act.  `taps`, if a dict, receives named intermediate activations."""
    nd = cfg["ndim"]
    act = cfg.get("act", "LeakyReLU")
    filters, skips = cfg["filters"], cfg["skip"]
    assert len(filters) == len(skips) + 1
    n_scales = len(filters)
    blk = block3d if nd == 3 else block2d
    rpath = respath3d if nd == 3 else respath2d

    def level(pre, x, i):
        """Everything hanging off model_tmp at scale i>=1: [Concat(skip, deeper)] then decoder block."""
        cpre = pre + ("1" if i > 1 else "2")        # top level: '1' is the first Block, Concat is '2'
        dpre = pre + ("2" if i > 1 else "3")        # decoder block
        has_skip = skips[i - 1] != 0
        deep_pre = cpre + ".1" if has_skip else cpre
        if nd == 3:
            d = S.conv(deep_pre + ".1.0", x, stride=2)
            d = activation(act, S.bn(deep_pre + ".2", d))
            d = blk(S, deep_pre + ".5", d, act)
            nxt = 6
        else:
            d = activation(act, S.conv(deep_pre + ".1.0", x, stride=2))
            d = blk(S, deep_pre + ".4", d, act)
            nxt = 5
        if taps is not None:
            taps["enc%d" % i] = d
        if i != n_scales - 1:
            d = level(deep_pre + ".%d." % nxt, d, i + 1)
        d = upsample2x(d, cfg["upsample"])
        if has_skip:
            s = rpath(S, cpre + ".0.1", x, act)
            d = concat_crop([s, d])
        out = blk(S, dpre, d, act)
        if taps is not None:
            taps["dec%d" % i] = out
        return out

    h = blk(S, "1", x, act)
    if taps is not None:
        taps["enc0"] = h
    if n_scales > 1:
        h = level("", h, 1)
        okey = "4.0"
    else:
        okey = "2.0"
    out = S.conv(okey, h)
    la = cfg.get("last_act")
    if isinstance(la, str) and la.lower() == "none":
        la = None
    if la is not None:
        out = activation(la, out)
    return out


def skip3d_forward(S, x, cfg):
    """Skip3D (skip.py:154-254), zero pad, stride downsampling, need1x1_up=True.
    cfg keys: filters (down == up), skip, upsample, act, last_act.  2-D `Skip` (cfg['ndim']==2,
    keys prefixed 'model.') shares the structure."""
    nd = cfg.get("ndim", 3)
    act = cfg.get("act", "LeakyReLU")
    filters, skips = cfg["filters"], cfg["skip"]
    assert len(filters) == len(skips)
    n_scales = len(filters)
    root = "model." if nd == 2 else ""

    def level(pre, x, i):
        has_skip = skips[i] != 0
        deep = pre + "1.1" if has_skip else pre + "1"
        d = activation(act, S.bn(deep + ".2", S.conv(deep + ".1.0", x, stride=2)))
        d = activation(act, S.bn(deep + ".6", S.conv(deep + ".5.0", d)))
        if i != n_scales - 1:
            d = level(deep + ".9.", d, i + 1)
        d = upsample2x(d, cfg["upsample"])
        if has_skip:
            s = activation(act, S.bn(pre + "1.0.2", S.conv(pre + "1.0.1.0", x)))
            d = concat_crop([s, d])
        d = S.bn(pre + "2", d)
        d = activation(act, S.bn(pre + "4", S.conv(pre + "3.0", d)))
        d = activation(act, S.bn(pre + "8", S.conv(pre + "7.0", d)))
        return d

    h = level(root, x, 0)
    out = S.conv(root + "11.0", h)
    la = cfg.get("last_act")
    if isinstance(la, str) and la.lower() == "none":
        la = None
    if la is not None:
        out = activation(la, out)
    return out


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d defaults (affine=False, batch statistics): per-channel normalisation of the single patch."""
    dims = list(range(2, x.ndim))
    mean = x.mean(dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dims, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def unet_forward(S, x, cfg):
    """Plain 2-D UNet (unet.py:84-187) with more_layers=0, concat_x=False, zero padding, InstanceNorm.
    cfg keys: upsample ('deconv' | 'bilinear' | 'nearest'), act, last_act."""
    act = cfg.get("act", "LeakyReLU")
    mode = cfg["upsample"]

    def conv_block(pre, t, norm):
        t = S.conv(pre + ".0.0", t)
        if norm:
            t = instance_norm(t)
        return activation(act, t)

    def unet_conv(pre, t, norm=True):
        return conv_block(pre + ".conv2", conv_block(pre + ".conv1", t, norm), norm)

    def down(pre, t):
        return unet_conv(pre + ".conv", F.max_pool2d(t, 2, 2))

    def up(pre, deep, skip):
        if mode == "deconv":
            S.used.update((pre + ".up.weight", pre + ".up.bias"))
            u = F.conv_transpose2d(deep, S.P[pre + ".up.weight"], S.P[pre + ".up.bias"], stride=2, padding=1)
        else:
            u = S.conv(pre + ".up.1.0", upsample2x(deep, mode))
        return unet_conv(pre + ".conv", concat_crop([u, skip]), norm=False)

    in64 = unet_conv("start", x)
    d1 = down("down1", in64)
    d2 = down("down2", d1)
    d3 = down("down3", d2)
    d4 = down("down4", d3)
    u = up("up4", d4, d3)
    u = up("up3", u, d2)
    u = up("up2", u, d1)
    u = up("up1", u, in64)
    la = cfg.get("last_act")
    if isinstance(la, str) and la.lower() == "none":
        la = None
    if la is not None:
        return activation(la, S.conv("final.0.0", u))
    return S.conv("final.0", u)


def net_forward(S, x, cfg, taps=None):
    if cfg.get("net", "multiunet") == "skip":
        return skip3d_forward(S, x, cfg)
    if cfg.get("net") == "unet":
        return unet_forward(S, x, cfg)
    return mulresunet_forward(S, x, cfg, taps)


# --------------------------------------------------------------------------------------
# init (utils/torch.py:23-58)
# --------------------------------------------------------------------------------------
def xavier_std(weight_shape, gain):
    """xavier_normal_: std = gain * sqrt(2 / (fan_in + fan_out)), fans include the receptive field."""
    rf = int(np.prod(weight_shape[2:])) if len(weight_shape) > 2 else 1
    fan_in, fan_out = weight_shape[1] * rf, weight_shape[0] * rf
    return gain * math.sqrt(2.0 / (fan_in + fan_out))


# --------------------------------------------------------------------------------------
# loss + metrics
# --------------------------------------------------------------------------------------
def masked_loss(out, img, mask, kind="mae"):
    """main.py:161: loss_fn(out*mask, img*mask), reduction='mean' over ALL elements."""
    d = out * mask - img * mask
    return (d * d).mean() if kind == "mse" else d.abs().mean()


def snr(output, target):
    """utils/metrics.py:15."""
    return 10 * torch.log10(torch.sum(target ** 2) / torch.sum((target - output) ** 2))


def pcorr(output, target):
    """utils/metrics.py:32-36."""
    td = target - target.mean()
    od = output - output.mean()
    return torch.sum(td * od) / (torch.sqrt(torch.sum(td ** 2)) * torch.sqrt(torch.sum(od ** 2)))


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults used at main.py:200)
# --------------------------------------------------------------------------------------
def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam step (step counts from 1). Returns (p, m, v) new tensors; inputs untouched."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


class PlateauLR:
    """torch ReduceLROnPlateau(mode='min', threshold_mode='rel') as configured at main.py:201-204."""

    def __init__(self, lr, factor, threshold, patience, min_lr=0.0, eps=1e-8):
        self.lr, self.factor, self.threshold, self.patience = lr, factor, threshold, patience
        self.min_lr, self.eps = min_lr, eps
        self.best = math.inf
        self.bad = 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best = metric
            self.bad = 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.bad = 0
        return self.lr


class EarlyStop:
    """utils/torch.py:216-275 with percentage=True, mode min (main.py:206-208)."""

    def __init__(self, patience, min_delta):
        self.patience, self.min_delta = patience, min_delta
        self.best = None
        self.bad = 0

    def step(self, metric):
        metric = float(metric)
        if self.patience == 0:
            return False
        if self.best is None:
            self.best = metric
            return False
        if math.isnan(metric):
            return True
        if metric < self.best - (self.best * self.min_delta / 100):
            self.bad = 0
            self.best = metric
        else:
            self.bad += 1
        return self.bad >= self.patience


# --------------------------------------------------------------------------------------
# the optimisation loop (main.py:141-220) with host-supplied net inputs
# --------------------------------------------------------------------------------------
def optimize(S, cfg, z, img, mask, epochs, lr=1e-3, loss_kind="mae", reg_noise_std=0.03,
             net_inputs=None, generator=None, reduce_lr=None, earlystop=None):
    """Runs `epochs` Adam iterations.  `net_inputs[k]` (if given) is the already-perturbed input of
    iteration k; otherwise input_k = z + reg_noise_std * N(0,1) drawn from `generator`.
    Returns dict(loss, snr, pcorr, lr lists, out_best, loss_min)."""
    params = S.params()
    keys = list(S.P.keys())
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    hist = {"loss": [], "snr": [], "pcorr": [], "lr": []}
    out_best, loss_min = None, None
    sched = PlateauLR(lr, *reduce_lr) if reduce_lr else None
    stopper = EarlyStop(*earlystop) if earlystop else None
    cur_lr = lr
    for it in range(epochs):
        if net_inputs is not None:
            inp = torch.as_tensor(net_inputs[it]).to(z.dtype)
        else:
            inp = z.detach().clone()
            if reg_noise_std > 0:
                inp = inp + reg_noise_std * torch.randn(inp.shape, generator=generator, dtype=inp.dtype)
        for p in params:
            p.grad = None
        out = net_forward(S, inp, cfg)
        loss = masked_loss(out, img, mask, loss_kind)
        loss.backward()
        l = loss.item()
        hist["loss"].append(l)
        hist["snr"].append(snr(out.detach(), img).item())
        hist["pcorr"].append(pcorr(out.detach(), img).item())
        hist["lr"].append(cur_lr)
        if it == 0 or l <= loss_min:
            loss_min = l
            out_best = out.detach().clone()
        with torch.no_grad():
            for i, k in enumerate(keys):
                p = S.P[k]
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                newp, m[i], v[i] = adam_update(p.detach(), g, m[i], v[i], it + 1, cur_lr)
                p.copy_(newp)
        if sched is not None:
            cur_lr = sched.step(l)
        if stopper is not None and stopper.step(l):
            break
    hist["out_best"] = out_best
    hist["loss_min"] = loss_min
    return hist


# --------------------------------------------------------------------------------------
# patch extraction / overlap-add reassembly (utils/patch_extractor.py:299-428, data.py:44-130)
# --------------------------------------------------------------------------------------
def patch_grid(in_shape, dim, stride):
    """Number of windows per axis: (in - dim)//stride + 1  (patch_extractor.py:153-155)."""
    return tuple((int(n) - int(d)) // int(s) + 1 for n, d, s in zip(in_shape, dim, stride))


def extract_patches_nd(vol, dim, stride):
    """Window w (C-order over the window grid) starts at w*stride.  dim==stride crops to whole
    blocks first (patch_extractor.py:320-325) — the same set of windows."""
    grid = patch_grid(vol.shape, dim, stride)
    out = np.empty(grid + tuple(dim), dtype=vol.dtype)
    for idx in np.ndindex(*grid):
        sl = tuple(slice(i * s, i * s + d) for i, s, d in zip(idx, stride, dim))
        out[idx] = vol[sl]
    return out


def reconstruct_nd(patch_array, dim, stride):
    """Overlap-add in C-order of window indices, divided by the hit count (patch_extractor.py:395-426)."""
    nd = len(dim)
    grid = patch_array.shape[:nd]
    shape = tuple((g - 1) * s + d for g, s, d in zip(grid, stride, dim))
    acc = np.zeros(shape)
    cnt = np.zeros(shape)
    for idx in np.ndindex(*grid):
        sl = tuple(slice(i * s, i * s + d) for i, s, d in zip(idx, stride, dim))
        acc[sl] += patch_array[idx]
        cnt[sl] += 1
    return (acc / cnt).astype(patch_array.dtype)


def nan_to_binary_mask(a):
    """utils/processing.py:27-31 bool2bin(logic=True): finite -> 1, NaN -> 0."""
    return np.where(np.isnan(a), 0.0, 1.0).astype(a.dtype)


# ------------------------------------------------------------------------------------------
# Anti-aliasing add-on operators and POCS, numpy float64 restatements
# ------------------------------------------------------------------------------------------
def first_derivative_np(x, spacing=1.0, axis=0, stencil="forward"):
    """reference utils/processing.py:139-162."""
    x = np.moveaxis(np.asarray(x), axis, 0)
    g = np.zeros_like(x)
    if stencil == "centered":
        g[1:-1] = (0.5 * x[2:] - 0.5 * x[:-2]) / spacing
    elif stencil == "forward":
        g[:-1] = (x[1:] - x[:-1]) / spacing
    elif stencil == "backward":
        g[1:] = (x[1:] - x[:-1]) / spacing
    else:
        raise ValueError(stencil)
    return np.moveaxis(g, 0, axis)


def second_derivative_np(x, spacing=1.0, axis=0):
    """reference utils/processing.py:165-181."""
    x = np.moveaxis(np.asarray(x), axis, 0)
    g = np.zeros_like(x)
    g[1:-1] = (x[2:] - 2 * x[1:-1] + x[:-2]) / spacing ** 2
    return np.moveaxis(g, 0, axis)


def vertical_grad_np(x, adjoint=False):
    """reference operators/derivative.py:8-21 (BCHW, difference along H)."""
    x = np.asarray(x)
    y = np.zeros_like(x)
    if not adjoint:
        y[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
    else:
        y[:, :, :-1] -= x[:, :, :-1]
        y[:, :, 1:] += x[:, :, :-1]
    return y


def hale2d_np(x, theta):
    """reference utils/slopes.py:51-69 / 72-105: -(Dh(a Dv x + b Dh x) + Dv(b Dv x + c Dh x)), forward differences."""
    u1, u2 = np.cos(theta), -np.sin(theta)
    gv = first_derivative_np(x, axis=2)
    gh = first_derivative_np(x, axis=3)
    p1 = u1 * u1 * gv + u1 * u2 * gh
    p2 = u1 * u2 * gv + u2 * u2 * gh
    return -(first_derivative_np(p1, axis=3) + first_derivative_np(p2, axis=2))


def linear_operator_matrix(fn, shape):
    """Dense matrix of a linear map on tensors of `shape` (tiny shapes only): column k = fn(e_k).  Used to check adjoint
    kernels against the TRANSPOSE of the forward operator."""
    n = int(np.prod(shape))
    cols = []
    for k in range(n):
        e = np.zeros(n)
        e[k] = 1.0
        cols.append(np.asarray(fn(e.reshape(shape))).reshape(-1))
    return np.stack(cols, axis=1)


def gaussian_kernel_np(M, std):
    n = np.arange(0, M) - (M - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


def conv_same_axis_np(x, taps, axis):
    """y[i] = sum_k taps[k] x[i + K//2 - k], zero padded (the ConvTransposeNd with padding K//2 of utils/processing.py:122-130)."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    K = len(taps)
    p = K // 2
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(p, p)])
    y = np.zeros_like(x)
    n = x.shape[-1]
    for k in range(K):
        y += taps[k] * xp[..., 2 * p - k:2 * p - k + n]
    return np.moveaxis(y, -1, axis)


def gaussian_filter_np(x, kernel_size, std):
    """Separable Gaussian blur over all axes after (B, C) (reference GaussianFilter, utils/processing.py:112-136)."""
    w = gaussian_kernel_np(kernel_size, std)
    y = np.asarray(x, dtype=np.float64)
    for ax in range(2, y.ndim):
        y = conv_same_axis_np(y, w, ax)
    return y


def structure_tensor_dips_np(x, dv=1.0, dh=1.0, smooth=0.0, dtype=np.float64):
    """reference utils/slopes.py:6-48.  dtype=np.float32 mimics the reference's own arithmetic on fp32 tensors: the dip angle
    atan((l1 - gvv) / gvh) cancels catastrophically where the tensor is nearly diagonal, so fp32 and fp64 legitimately differ there."""
    x = np.asarray(x, dtype=dtype)
    gv = first_derivative_np(x, dv, 2)
    gh = first_derivative_np(x, dh, 3)
    gvv, gvh, ghh = gv * gv, gv * gh, gh * gh
    if smooth > 0:
        K = 2 * min(x.shape[2], x.shape[3]) // 2 + 1
        gvv, gvh, ghh = (gaussian_filter_np(t, K, smooth).astype(dtype) for t in (gvv, gvh, ghh))
    t1 = 0.5 * (gvv + ghh)
    t2 = 0.5 * np.sqrt((gvv - ghh) ** 2 + 4 * gvh ** 2)
    l1, l2 = t1 + t2, t1 - t2
    with np.errstate(all="ignore"):
        phi = np.arctan((l1 - gvv) / gvh)
        phi[np.isnan(phi)] = 0.0
        aniso = 1 - l2 / l1
    return phi, aniso


def pocs_np(x, wdata_weight, data, mask, perc):
    """reference utils/pocs.py:5-19,80-84 with the two-sided FFT pair of main_pocs.py:156-157 (torch.rfft / irfft, onesided=False):
    threshold real and imaginary parts as independent reals at max * perc / 100."""
    x = np.asarray(x, dtype=np.float64)
    axes = tuple(range(2, x.ndim))
    X = np.fft.fftn(x, axes=axes)
    re, im = X.real.copy(), X.imag.copy()
    th = max(re.max(), im.max()) * perc / 100.0
    re = re * ((re > th).astype(float) + (re < -th).astype(float))
    im = im * ((im > th).astype(float) + (im < -th).astype(float))
    xr = np.fft.ifftn(re + 1j * im, axes=axes).real
    return wdata_weight * data + (1.0 - wdata_weight * mask) * xr, th
