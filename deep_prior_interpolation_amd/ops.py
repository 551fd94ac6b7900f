"""torch.autograd.Function wrappers over the C ABI (leaf ops of the drop-in module tree).

Every op here runs a hand-written HIP kernel from libdpi_hip.so on the current HIP stream; there is no
aten / CPU fallback.  Tensors must be fp32, on a HIP device, batch size 1 (the reference optimises one
patch at a time, main.py:131-135), layout (1, C, [D,] H, W) contiguous.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, stream

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _req(t, name, act=False):
    """act=True: an ACTIVATION (or the gradient of one) of a fused 3-D node — float32, or bfloat16 in the bf16-storage mode."""
    if not t.is_cuda:
        raise _lib.DpiError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32 and not (act and t.dtype == torch.bfloat16):
        raise _lib.DpiError("%s must be float32%s (got %s)" % (name, " or bfloat16" if act else "", t.dtype))
    return t.contiguous()


# ---- storage type of activations in HBM (BASELINE configs[4]: "bf16 activations + fp32 Adam master weights") ---------------------------
# False: every tensor fp32 (the reference's storage, main.py:112).  True: the fused 3-D nodes (Block3dFn, ResPath3dFn, SkipJoinFn,
# ConvBnActFn) allocate their forward tensors (R, S, t, y, cat) AND the gradients of those as torch.bfloat16; the kernels widen on load and
# round to nearest-even on store (dpi_conv_desc.io, the *_io entry points), arithmetic / BatchNorm statistics / weights / weight gradients /
# Adam stay fp32.  A leaf convolution (the network's output layer) reads bf16 and writes fp32, so loss and metrics see fp32.
STORAGE_BF16 = os.environ.get("DPI_STORAGE", "fp32") == "bf16"      # process DEFAULT (tools / tests); an Interpolator's own mode is a mode_scope
_tls = threading.local()


def set_storage(name):
    """Process default (tools / tests).  The product path does not call it: an Interpolator runs every iteration inside its own mode_scope."""
    global STORAGE_BF16
    if name not in ("fp32", "bf16"):
        raise ValueError("storage must be fp32 or bf16")
    STORAGE_BF16 = name == "bf16"


class mode_scope:
    """with mode_scope(precision, storage): the arithmetic mode / activation storage type of the fused nodes BUILT inside the block, on THIS host
    thread only (round 6: they used to be module globals flipped per iteration, which two Interpolators of different --precision driven from two
    threads would race on).  None leaves that half to the process default (set_precision / set_storage, DPI_PRECISION / DPI_STORAGE).  Everything
    that depends on the mode is decided in the FORWARD of a node, on the calling thread (make_desc, act_dtype); the backward — which autograd runs on
    its own device thread — only uses the descriptors and tensor types the forward saved.  Scopes nest; leaving one restores the previous."""

    def __init__(self, precision=None, storage=None):
        if precision is not None and precision not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        if storage is not None and storage not in ("fp32", "bf16"):
            raise ValueError("storage must be fp32 or bf16")
        self.mode = (None if precision is None else _PRECISIONS[precision], None if storage is None else storage == "bf16")

    def __enter__(self):
        self.prev = getattr(_tls, "mode", None)
        _tls.mode = self.mode
        return self

    def __exit__(self, *exc):
        _tls.mode = self.prev
        return False


def precision():
    """Arithmetic mode (dpi_conv_desc.precision) of the layers built now: the innermost mode_scope of this thread, else the process default."""
    m = getattr(_tls, "mode", None)
    return PRECISION if (m is None or m[0] is None) else m[0]


def storage_bf16():
    m = getattr(_tls, "mode", None)
    return STORAGE_BF16 if (m is None or m[1] is None) else m[1]


def act_dtype():
    return torch.bfloat16 if storage_bf16() else torch.float32


def _bf(t):
    return t is not None and t.dtype == torch.bfloat16


def _io(fwd=None, grad=None):
    """`io` mask of an *_io entry point from the tensors themselves."""
    return (_lib.STORE_FWD_BF16 if _bf(fwd) else 0) | (_lib.STORE_GRAD_BF16 if _bf(grad) else 0)


def _same_type(a, b, what):
    if a.dtype != b.dtype:
        raise _lib.DpiError("%s: tensors of one call must share their storage type (got %s and %s)" % (what, a.dtype, b.dtype))


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


# 0: fp32 arithmetic (the reference's).  1: BASELINE configs[4] mixed precision — 3x3(x3) stride-1 convolutions feed bf16-rounded
# operands to the matrix cores with fp32 accumulation; tensors, master weights, BatchNorm statistics and Adam stay fp32.
# 2: "split" mode — fp32 operands split exactly into three bf16 terms, six partial products accumulated in fp32: fp32-class accuracy
# on the bf16 matrix cores (forward / backward-data of the shapes where the kernel wins).
# "bf16mm" is mode 1 under its round-2/3 meaning (operands only); "bf16" is the same arithmetic and, where the net supports it, bf16
# STORAGE of the activations on top (Interpolator.precision_scope; round 4).
_PRECISIONS = {"fp32": 0, "bf16": 1, "bf16mm": 1, "split": 2}
PRECISION = _PRECISIONS.get(os.environ.get("DPI_PRECISION", "fp32"), 0)


def set_precision(name):
    """Process default (tools / tests); see mode_scope."""
    global PRECISION
    if name not in _PRECISIONS:
        raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
    PRECISION = _PRECISIONS[name]


def make_desc(x, w, stride, ydt=torch.float32):
    """Descriptor of one layer.  `ydt`: storage type of the layer's OUTPUT; the gradient of a tensor shares the tensor's type
    (autograd's rule), so x decides the X and DX bits and ydt the Y and DY bits of dpi_conv_desc.io."""
    Cin, D, H, W = _dims(x)
    k = w.shape[-1]
    kd = w.shape[2] if w.ndim == 5 else 1
    if w.shape[1] != Cin:
        raise _lib.DpiError("conv: weight expects %d input channels, tensor has %d" % (w.shape[1], Cin))
    io = ((_lib.IO_X_BF16 | _lib.IO_DX_BF16) if _bf(x) else 0) | ((_lib.IO_Y_BF16 | _lib.IO_DY_BF16) if ydt == torch.bfloat16 else 0)
    return ConvDesc(Cin, w.shape[0], D, H, W, k, kd, int(stride), precision(), io)


def desc_out_dims(d):
    sd = d.stride if d.kd > 1 else 1
    return conv_out(d.D, d.kd, sd), conv_out(d.H, d.k, d.stride), conv_out(d.W, d.k, d.stride)


# ------------------------------------------------------------------------------------------------
# raw (non-autograd) launches, shared with the fused engine
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Brackets selected launches with HIP events on the stream they are launched on (torch's current stream),
    for bench.py's roofline line.  `match(kind, desc)` selects launches: a falsy result skips the launch, anything else is
    the key the launch is filed under.  durations() gives all milliseconds, by_key() groups them."""

    def __init__(self, match):
        self.match = match
        self.events = []

    def durations(self):
        return [a.elapsed_time(b) for _, a, b in self.events]

    def by_key(self):
        out = {}
        for key, a, b in self.events:
            out.setdefault(key, []).append(a.elapsed_time(b))
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


# ---- profile tags: DPI_PROFILE_TAGS=<json path> brackets every conv launch with dpi_profile_marker(id) dispatches and writes the
# id -> layer table at exit, so that tools/rocpd_stats.py can give every layer its own row in a rocprofv3 kernel trace.
_PROFILE_TAGS = os.environ.get("DPI_PROFILE_TAGS")
_tag_ids = {}


def _tag_id(kind, d):
    key = "%s %d->%d k%d%s s%d @%dx%dx%d" % (kind, d.Cin, d.Cout, d.k, " (2-D)" if d.D == 1 else "", d.stride, d.D, d.H, d.W)
    i = _tag_ids.get(key)
    if i is None:
        i = len(_tag_ids) + 2
        _tag_ids[key] = i
    return i


if _PROFILE_TAGS:
    import atexit
    import json

    def _dump_tags():
        with open(_PROFILE_TAGS, "w") as fp:
            json.dump({str(i): k for k, i in _tag_ids.items()}, fp, indent=0)
    atexit.register(_dump_tags)


def _timed(kind, d, launch):
    if _PROFILE_TAGS:
        L = _lib.load()
        L.dpi_profile_marker(_tag_id(kind, d), stream())
        launch()
        L.dpi_profile_marker(1, stream())
        return
    key = _timer.match(kind, d) if _timer is not None else None
    if key:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        _timer.events.append((key, e0, e1))
    else:
        launch()


_ws_cache = {}


def _ws_floats(kind, d):
    """dpi_conv_*_ws_floats(d), memoised per (kind, descriptor): the call re-runs descriptor validation and launch planning, ~150 host
    round trips per eager iteration otherwise.  Forward / backward-data only, where a stale answer is harmless: the
    library ignores a workspace it does not need and runs the unsplit launch when it gets none (include/dpi_hip.h); the split /
    kernel-selection knobs (dpi_set_splitk, dpi_set_q4, ...) are test / tool hooks — `reset_ws_cache()` after flipping one."""
    key = (kind, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.kd, d.stride, d.precision, d.io)
    n = _ws_cache.get(key)
    if n is None:
        L = _lib.load()
        n = {"fwd": L.dpi_conv_fwd_ws_floats, "bwd_data": L.dpi_conv_bwd_data_ws_floats}[kind](C.byref(d))
        _ws_cache[key] = n
    return n


def reset_ws_cache():
    _ws_cache.clear()


def _conv_ws(n, like):
    """Workspace of the input-channel split (coarse levels; dpi_conv_*_ws_floats is 0 for every other launch)."""
    return torch.empty(n, dtype=torch.float32, device=like.device) if n else None


def raw_conv_fwd(d, x, chain, w, bias, y, partials=None):
    L = _lib.load()
    n = _ws_floats("fwd", d)
    ws = _conv_ws(n, x)
    _timed("conv_fwd", d, lambda: check(L.dpi_conv_fwd_ws(C.byref(d), ptr(x), ptr(chain), ptr(w), ptr(bias), ptr(y),
                                                          ptr(partials), ptr(ws), n, stream()), "dpi_conv_fwd"))


def raw_conv_bwd_data(d, dy, w, dx, accumulate=False):
    L = _lib.load()
    n = _ws_floats("bwd_data", d)
    ws = _conv_ws(n, dy)
    _timed("conv_bwd_data", d, lambda: check(L.dpi_conv_bwd_data_ws(C.byref(d), ptr(dy), ptr(w), ptr(dx), int(accumulate),
                                                                    ptr(ws), n, stream()), "dpi_conv_bwd_data"))


def raw_conv_bwd_data_dual(d3, dy3, w3, d1, dy1, w1, dx, accumulate=False):
    """dx (+)= conv_transpose(dy3, w3) + conv_transpose(dy1, w1): a 3x3(x3) layer and a 1x1(x1) layer that read the same tensor
    (Block3d.conv1 + shortcut, ResPath3d.conv3x3 + conv1x1).  One pass over dx where the MFMA stencil kernel serves the 3x3(x3) layer
    (the 1x1x1 term is a few extra MFMAs on operands read straight from dy1), two launches (write, then add) otherwise."""
    L = _lib.load()
    n = _ws_floats("bwd_data", d3)
    ws = _conv_ws(n, dy3)

    def launch():
        check(L.dpi_conv_bwd_data_dual(C.byref(d3), ptr(dy3), ptr(w3), C.byref(d1), ptr(dy1), ptr(w1), ptr(dx), int(accumulate),
                                       ptr(ws), n, stream()), "dpi_conv_bwd_data_dual")
    _timed("conv_bwd_data", d3, launch)


# ------------------------------------------------------------------------------------------------
# side stream for weight gradients: dW of a layer depends only on (x, dy) and nothing downstream depends on it before the
# optimiser step, so it runs concurrently with the backward-data / BatchNorm-backward chain.  At the coarse levels of the
# U-Net a single kernel fills only part of the chip (tens to hundreds of workgroups), the pair fills more of it.
# ------------------------------------------------------------------------------------------------
# Off by default: it pays when the iteration is GPU-bound (patches of >= 2^20 voxels, eager loop: 40.7 -> 39.6 ms at
# 256x128x128) and costs host time when it is launch-bound (64^3 eager: 8.8 -> 11.3 ms).  Interpolator.optimize / bench.py
# switch it on by patch size; DPI_OVERLAP_WGRAD=0/1 forces it.
OVERLAP_WEIGHT_GRADS = os.environ.get("DPI_OVERLAP_WGRAD", "0") == "1"
OVERLAP_IN_GRAPH = os.environ.get("DPI_OVERLAP_IN_GRAPH", "0") == "1"
OVERLAP_MAX_VOXELS = int(os.environ.get("DPI_OVERLAP_MAX_VOXELS", str(1 << 40)))


def set_weight_grad_overlap(on, in_graph=None):
    """in_graph: keep the side stream inside a hipGraph capture too (fork / join edges in the graph).  Pays on GPU-bound patches
    (256x128x128: 34.2 ms eager with overlap, 34.2 ms graph without, 33.8 ms graph with), costs on small ones (64^3: 7.6 vs 7.9 ms)."""
    global OVERLAP_WEIGHT_GRADS, OVERLAP_IN_GRAPH
    if "DPI_OVERLAP_WGRAD" not in os.environ:
        OVERLAP_WEIGHT_GRADS = bool(on)
    if in_graph is not None and "DPI_OVERLAP_IN_GRAPH" not in os.environ:
        OVERLAP_IN_GRAPH = bool(in_graph)
_side_streams = {}
N_SIDE_STREAMS = int(os.environ.get("DPI_SIDE_STREAMS", "2"))     # weight gradients round-robin over this many side streams (four alternating bench runs each: 32.43 / 32.12 / 32.33 ms with 1 / 2 / 3)
_side_rr = [0]
_side_used = set()
# Where the main stream waits for the weight-gradient streams.  "node": at the end of every fused node's backward (rounds 1-4).
# "step": once, after the whole backward (finish_backward(), before the optimiser step): the main chain never stalls behind a
# weight-gradient kernel, the side streams drain their backlog under the BatchNorm-backward / up-sampling passes of the nodes that
# follow.  The tensors a side-stream launch reads or writes were allocated on the main stream's pool; they are kept referenced in
# _side_keep until the join so that the caching allocator cannot hand their memory to a later main-stream tensor.
# Only between begin_iteration() and finish_backward() (the Interpolator's loops); a bare loss.backward() joins per node.
JOIN_AT = os.environ.get("DPI_JOIN_AT", "step")
_side_keep = []
_in_iteration = [False]
# Gradients handed back to autograd BEFORE the stream that writes them has run (weight gradients on the side streams; with the branch stream also the
# ResPath's BatchNorm gradients).  That is only sound if autograd's AccumulateGrad ADOPTS the tensor as .grad — which needs .grad to be None (zero_grad with
# set_to_none=True), no hook, no second use of the parameter; anything else makes it clone or add on the main stream, ahead of the producer.
# finish_backward(params) checks exactly that: every deferred tensor must BE some parameter's .grad storage afterwards.
_deferred = []


def begin_iteration():
    """Top of every iteration (eager or captured): the weight-gradient launches are dealt to the side streams from stream 0 again, so
    the kernel-to-stream assignment is the same in every iteration and every run."""
    _side_rr[0] = 0
    _in_iteration[0] = True
    del _deferred[:]


def abort_iteration():
    """An iteration raised between begin_iteration() and finish_backward(): join every stream and forget the per-iteration state, so that a later
    bare loss.backward() joins per node again instead of believing it is still inside an Interpolator iteration."""
    try:
        if torch.cuda.is_available():
            join_weight_grads(final=True)
            join_branch()
    finally:
        _in_iteration[0] = False
        _side_used.clear()
        del _side_keep[:]
        del _deferred[:]
        _branch_open[0] = False


def _defer(*tensors):
    if JOIN_AT == "step" and _in_iteration[0]:
        _deferred.extend(t.data_ptr() for t in tensors if t is not None)


def _side_stream():
    dev = torch.cuda.current_device()
    sts = _side_streams.get(dev)
    if sts is None:
        sts = [torch.cuda.Stream(device=dev) for _ in range(max(N_SIDE_STREAMS, 1))]
        _side_streams[dev] = sts
    i = _side_rr[0] % len(sts)
    _side_rr[0] += 1
    _side_used.add(i)
    return sts[i]


def conv_bwd_weight_async(d, x, chain, dy, dw):
    """raw_conv_bwd_weight on the side stream (ordered after everything already queued on the current stream).
    Callers must `join_weight_grads()` before returning to autograd."""
    # (DPI_OVERLAP_MAX_VOXELS: A/B knob — restricting the side stream to the coarse levels, whose kernels leave CUs idle, measured
    #  35.76 ms per iteration against 35.10 with every layer on it, round 3)
    if not OVERLAP_WEIGHT_GRADS or d.D * d.H * d.W >= OVERLAP_MAX_VOXELS or (torch.cuda.is_current_stream_capturing() and not OVERLAP_IN_GRAPH):
        # (inside a hipGraph capture the fork / join edges cost more than the overlap wins on the small patches that are
        #  run as graphs: measured 7.6 vs 7.9 ms per iteration at 64^3)
        return raw_conv_bwd_weight(d, x, chain, dy, dw)
    side = _side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        raw_conv_bwd_weight(d, x, chain, dy, dw)
    if JOIN_AT == "step" and _in_iteration[0]:
        # (not dw: the parameter's .grad keeps it alive until the next zero_grad, and a second reference would make autograd's
        #  AccumulateGrad CLONE it on the main stream — before the side stream has written it — instead of adopting it)
        _side_keep.append((x, chain, dy))
        _deferred.append(dw.data_ptr())


def join_weight_grads(final=False):
    """End of a fused node's backward (final=False) / end of the whole backward (final=True, finish_backward)."""
    if JOIN_AT == "step" and _in_iteration[0] and not final:
        return
    if _side_used:
        sts = _side_streams.get(torch.cuda.current_device()) or []
        for i in sorted(_side_used):
            torch.cuda.current_stream().wait_stream(sts[i])
        _side_used.clear()
    _side_keep.clear()


def finish_backward(params=None):
    """After loss.backward(), before the optimiser step: every weight gradient launched on a side stream (and everything the branch
    stream still runs) is ordered in front of whatever the current stream does next.
    params (the optimised parameters): checks that every gradient that was handed to autograd ahead of its producer stream was ADOPTED as a
    parameter's .grad — not cloned, not accumulated into an existing .grad (see _deferred) — and fails loudly otherwise."""
    join_weight_grads(final=True)
    join_branch()
    _in_iteration[0] = False
    if _deferred and params is not None:
        have = {p.grad.data_ptr() for p in params if p.grad is not None}
        lost = [q for q in _deferred if q not in have]
        if lost:
            n = len(_deferred)
            del _deferred[:]
            raise _lib.DpiError("%d of %d gradients written on a side / branch stream were copied or accumulated by autograd before that stream ran "
                                "(zero_grad(set_to_none=False), a gradient hook, retain_grad or a shared parameter?): with the once-per-step join "
                                "(DPI_JOIN_AT=step) .grad must be None when backward starts; use DPI_JOIN_AT=node otherwise" % (len(lost), n))
    del _deferred[:]


# ------------------------------------------------------------------------------------------------
# branch stream (round 5): work that is OFF the critical path of the U-Net runs beside it instead of in front of it.
#   forward:  the ResPath of level l (skip branch: conv3x3 + conv1x1 + join + BatchNorm, mulresunet.py:99-113) depends only on the
#             encoder output of its level, while its consumer — the level's concat — also waits for the whole deeper U;  the 1x1x1
#             shortcut of a MultiRes block (HBM-bound) depends only on the block input, while the three 3x3x3 layers (matrix-bound)
#             form the chain the block output waits for.
#   backward: the ResPath's backward (BatchNorm backward, backward-data, two weight gradients) feeds the encoder block of its level,
#             which the deeper U's backward reaches much later.
# An HBM-bound kernel beside a matrix-bound one costs ~20 % of its stand-alone time (tools/overlap_probe.py: 25->16 forward 0.80 ms +
# elementwise 0.43 ms = 0.89 ms side by side), and the coarse levels' launches leave most of the chip idle.
# Same kernels, same operands, same order per tensor: results are bit-identical to the serial schedule.
# Memory: tensors are allocated from the launching (main) stream's pool; each one a branch-stream kernel touches is marked with
# record_stream, so the caching allocator does not hand its memory out again before that kernel has run.
# Only inside an Interpolator iteration with the weight-gradient overlap on (patches >= 2^20 voxels, eager loop).
# ------------------------------------------------------------------------------------------------
BRANCH_STREAMS = os.environ.get("DPI_BRANCH", "1") == "1"
BRANCH_SHORTCUT = int(os.environ.get("DPI_BRANCH_SHORT", "0"))   # 0: OFF.  2 (round 6): beside the SECOND and THIRD 3x3x3 layer of the block (4->8, 8->13: matrix-bound, they read 4 / 8 channels), started once the first one has drained — measured +0.3 ms (29.7-30.1 against 29.3-29.5, profiles/r06/ab_fanin_shortcut.txt).  1: beside the first layer:     # OFF: the first 3x3x3 layer of a block (64->4, 67->4, 25->8: few output channels) is itself at ~half the HBM bandwidth, the 1x1x1 layer beside it gains nothing (64->4 0.70 -> 1.05 ms with the 0.35 ms shortcut beside it; 31.1-31.5 ms per iteration with, 30.6-30.8 without)
BRANCH_SKIP = os.environ.get("DPI_BRANCH_SKIP", "1") == "1"
BRANCH_SKIP_BWD = os.environ.get("DPI_BRANCH_SKIP_BWD", "1") == "1"
_branch_streams = {}
_branch_open = [False]


def branch_on():
    return (BRANCH_STREAMS and OVERLAP_WEIGHT_GRADS and _in_iteration[0] and JOIN_AT == "step"
            and not torch.cuda.is_current_stream_capturing())


def _branch_stream(kind="skip"):
    key = (torch.cuda.current_device(), kind)
    st = _branch_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=key[0])
        _branch_streams[key] = st
    return st


class _Branch:
    """with _Branch(kind, tensors...): launches go to the branch stream, ordered after everything queued on the current stream so
    far; every tensor listed (allocated by the current stream's pool) is marked as used there."""

    def __init__(self, kind, *tensors):
        self.st = _branch_stream(kind)
        self.tensors = tensors

    def __enter__(self):
        self.st.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            if t is not None:
                t.record_stream(self.st)
        self.ctx = torch.cuda.stream(self.st)
        self.ctx.__enter__()
        _branch_open[0] = True
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def join_branch(kind=None):
    """The current stream waits for the branch stream(s)."""
    if not _branch_open[0]:
        return
    dev = torch.cuda.current_device()
    for (d, k), st in _branch_streams.items():
        if d == dev and (kind is None or k == kind):
            torch.cuda.current_stream().wait_stream(st)
    if kind is None:
        _branch_open[0] = False


def raw_conv_bwd_weight(d, x, chain, dy, dw):
    L = _lib.load()
    n = L.dpi_conv_bwd_weight_ws_floats(C.byref(d))     # not memoised: the planner knobs (dpi_set_bw_tuning / _pair) change it and too small a workspace is an error
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    _timed("conv_bwd_weight", d, lambda: check(L.dpi_conv_bwd_weight(C.byref(d), ptr(x), ptr(chain), ptr(dy), ptr(dw),
                                                                     ptr(ws), n, stream()), "dpi_conv_bwd_weight"))


def raw_channel_sum(x, C_, V, out):
    L = _lib.load()
    nblk = L.dpi_stat_blocks(C_, V)
    ws = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    check(L.dpi_channel_sum(ptr(x), C_, V, ptr(ws), ptr(out), stream()), "dpi_channel_sum")


def raw_bn_stats_finalize(x, chain_in, C_, V, gamma, beta, slope, running_mean, running_var, nbt, mean_invstd, chain_out,
                          eps=BN_EPS, momentum=BN_MOMENTUM, act_first=0, compose=False):
    """channel stats of T(x) followed by finalize (compose=True: chain_out = BN o T_in instead of BN alone)."""
    L = _lib.load()
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    check(L.dpi_channel_stats_io(ptr(x), ptr(chain_in), C_, V, ptr(part), _io(x), stream()), "dpi_channel_stats")
    check(L.dpi_bn_finalize(ptr(part), nblk, C_, V, ptr(gamma), ptr(beta), eps, momentum, slope, act_first,
                            ptr(chain_in) if compose else None, ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean_invstd),
                            ptr(chain_out), stream()), "dpi_bn_finalize")


def raw_bn_finalize(part, nblk, C_, count, gamma, beta, slope, running_mean, running_var, nbt, mean_invstd, chain_out,
                    eps=BN_EPS, momentum=BN_MOMENTUM, act_first=0):
    L = _lib.load()
    check(L.dpi_bn_finalize(ptr(part), nblk, C_, count, ptr(gamma), ptr(beta), eps, momentum, slope, act_first, None,
                            ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean_invstd), ptr(chain_out), stream()),
          "dpi_bn_finalize")


def raw_upsample2x_bwd(dy, C_, D, H, W, Do, Ho, Wo, linear, dx):
    L = _lib.load()
    n = L.dpi_upsample2x_bwd_ws_floats(C_, D, H, W, Do, Ho, Wo, linear)
    ws = torch.empty(n, dtype=torch.float32, device=dy.device) if n else None
    _same_type(dy, dx, "upsample2x_bwd")
    check(L.dpi_upsample2x_bwd_io(ptr(dy), C_, D, H, W, Do, Ho, Wo, linear, ptr(dx), ptr(ws), _io(None, dy), stream()), "dpi_upsample2x_bwd")


def raw_chain_apply(x, chain, C_, V, y):
    L = _lib.load()
    _same_type(x, y, "chain_apply")
    check(L.dpi_chain_apply_io(ptr(x), ptr(chain), C_, V, ptr(y), _io(x), stream()), "dpi_chain_apply")


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
        # (a bf16 input — the last fused block's output in the bf16-storage mode — is read as it is; the output of a leaf convolution is
        #  always fp32: it is the network's output layer, and the loss / metrics kernels take fp32)
        x, w = _req(x, "conv input", act=True), _req(w, "conv weight")
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
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            conv_bwd_weight_async(d, x, None, dy, dw)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw_conv_bwd_data(d, dy, w, dx)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(d.Cout, dtype=torch.float32, device=x.device)
            raw_channel_sum(dy, d.Cout, dy.numel() // d.Cout, db)
        join_weight_grads()
        return dx, dw, db, None


def _bn_backward(dy, x, mi, gamma, beta, pre_slope, post_slope, in_chain=None, dx=None):
    """two-phase BatchNorm backward with the surrounding LeakyReLU folded in; returns (dx, dgamma, dbeta).
    in_chain: the normalised tensor was T_in(x) and dx is the gradient w.r.t. T_in(x); dx: optional output buffer."""
    L = _lib.load()
    C_ = x.shape[1]
    V = x.numel() // C_
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    io = _io(x, dy)
    check(L.dpi_bn_bwd_reduce_io(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), ptr(in_chain), pre_slope, post_slope, C_, V,
                                 ptr(part), io, stream()), "dpi_bn_bwd_reduce")
    if dx is None:
        dx = torch.empty_like(dy)
    _same_type(dy, dx, "bn_backward")
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(gamma)
    check(L.dpi_bn_bwd_apply_io(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), ptr(in_chain), pre_slope, post_slope, ptr(part), nblk,
                                C_, V, ptr(dx), ptr(dgamma), ptr(dbeta), io, stream()), "dpi_bn_bwd_apply")
    return dx, dgamma, dbeta


def _bn_backward_fork(dy, x, mi, gamma, beta, pre_slope, post_slope, forks):
    """BatchNorm backward of (dy, x) whose result dx is the incoming gradient of the BatchNorms in `forks`
    (list of up to two (x_k, mi_k, gamma_k, beta_k, in_chain_k, post_slope_k)).  Returns dx, dgamma, dbeta and, per fork,
    its already-reduced phase-1 partials (nblk, tensor) so that `_bn_backward_apply` can finish it without a reduce pass."""
    L = _lib.load()
    C_ = x.shape[1]
    V = x.numel() // C_
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device)
    io = _io(x, dy)
    check(L.dpi_bn_bwd_reduce_io(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), None, pre_slope, post_slope, C_, V, ptr(part),
                                 io, stream()), "dpi_bn_bwd_reduce")
    dx = torch.empty_like(dy)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    fparts = [torch.empty(nblk * C_ * 2, dtype=torch.float64, device=x.device) for _ in forks]
    fa = []
    for k in range(2):
        if k < len(forks):
            xk, mik, gk, ek, chk, postk = forks[k]
            _same_type(x, xk, "bn_backward_fork")
            fa += [ptr(xk), ptr(mik), ptr(gk), ptr(ek), ptr(chk), postk, ptr(fparts[k])]
        else:
            fa += [None, None, None, None, None, 1.0, None]
    check(L.dpi_bn_bwd_apply_fork_io(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), None, pre_slope, post_slope, ptr(part), nblk, C_, V,
                                     ptr(dx), ptr(dgamma), ptr(dbeta), *fa, io, stream()), "dpi_bn_bwd_apply_fork")
    return dx, dgamma, dbeta, [(nblk, fp) for fp in fparts]


def _bn_backward_apply(dy, x, mi, gamma, beta, pre_slope, post_slope, red, in_chain=None, dx=None):
    """phase 2 only, from partials `red` = (nblk, tensor) produced by _bn_backward_fork."""
    L = _lib.load()
    C_ = x.shape[1]
    V = x.numel() // C_
    nblk, part = red
    if dx is None:
        dx = torch.empty_like(dy)
    _same_type(dy, dx, "bn_backward_apply")
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
    check(L.dpi_bn_bwd_apply_io(ptr(dy), ptr(x), ptr(mi), ptr(gamma), ptr(beta), ptr(in_chain), pre_slope, post_slope, ptr(part), nblk,
                                C_, V, ptr(dx), ptr(dgamma), ptr(dbeta), _io(x, dy), stream()), "dpi_bn_bwd_apply")
    return dx, dgamma, dbeta


def _bn_backward_apply_dual(dy, side_a, side_b, fork=None):
    """Phase 2 of two BatchNorm backward passes sharing the incoming gradient `dy`, in one kernel.
    side = (x, mi, gamma, beta, in_chain, post_slope, red) with red = (nblk, partials) from _bn_backward_fork.
    fork = (c_lo, c_hi, mi_f, gamma_f, beta_f, post_f): additionally take the phase-1 partials of the BatchNorm that dx_b
    feeds on channels [c_lo, c_hi) (its input being x_b itself).  Returns (dx_a, dgamma_a, dbeta_a), (dx_b, ...), red_f."""
    L = _lib.load()
    xa, mia, ga, ea, cha, posta, (nblk, parta) = side_a
    xb, mib, gb, eb, chb, postb, (nblk_b, partb) = side_b
    assert nblk == nblk_b
    C_ = xa.shape[1]
    V = xa.numel() // C_
    _same_type(xa, xb, "bn_backward_apply_dual")
    dxa, dxb = torch.empty_like(dy), torch.empty_like(dy)
    dga, dea, dgb, deb = (torch.empty_like(g) for g in (ga, ga, gb, gb))
    red_f = None
    fargs = [0, 0, None, None, None, 1.0, None]
    if fork is not None:
        lo, hi, mif, gf, ef, postf = fork
        nb = L.dpi_stat_blocks(C_, V)
        pf = torch.empty(nb * (hi - lo) * 2, dtype=torch.float64, device=xa.device)
        fargs = [lo, hi, ptr(mif), ptr(gf), ptr(ef), postf, ptr(pf)]
        red_f = (nb, pf)
    check(L.dpi_bn_bwd_apply_dual_io(ptr(dy), nblk, C_, V,
                                     ptr(xa), ptr(mia), ptr(ga), ptr(ea), ptr(cha), posta, ptr(parta), ptr(dxa), ptr(dga), ptr(dea),
                                     ptr(xb), ptr(mib), ptr(gb), ptr(eb), ptr(chb), postb, ptr(partb), ptr(dxb), ptr(dgb), ptr(deb),
                                     *fargs, _io(xa, dy), stream()), "dpi_bn_bwd_apply_dual")
    return (dxa, dga, dea), (dxb, dgb, deb), red_f


# The residual join's BatchNorm backward in two passes (dpi_join_bwd, ABI 403) instead of reduce / apply+fork / dual apply / apply:
# 10.5 instead of 13.6 tensor passes per Block3d, 10 instead of 12 per ResPath3d, and 3 launches instead of 4.  fp32 or bf16 storage (the
# rounds 1-4 bf16 kernels took their statistics of the ROUNDED intermediate gradient they stored; this path stores none, its sums describe
# the fp32 values).  DPI_JOIN_BWD=0: the rounds 1-4 sequence (A/B and test knob).
JOIN_BWD_FUSED = os.environ.get("DPI_JOIN_BWD", "1") == "1"
EARLY_SHORTCUT_WGRAD = os.environ.get("DPI_EARLY_SHORTCUT_WGRAD", "0") == "1"     # measured: 29.47-29.57 ms with, 29.36-29.43 without (profiles/r05/ab_early_shortcut.txt): off


def _join_backward(dy, t, mi, gamma, beta, pre_slope, side_a, side_b, fork=None, fwd_chains=None):
    """side = (x, mi, gamma, beta, in_chain, post_slope); fork = (lo, hi, mi_f, gamma_f, beta_f, post_f, dxf) with dxf the [hi - lo]-channel
    tensor (a slice of the block's dR) that receives the fork BatchNorm's input gradient.  t = None: the join's sum was never stored
    and is recomputed from the two sides through fwd_chains = (chain_a, chain_b) of the forward join.
    Returns (dxa, dgamma_a, dbeta_a), (dxb, dgamma_b, dbeta_b), (dgamma, dbeta) of the top BatchNorm, (dgamma_f, dbeta_f) or None."""
    L = _lib.load()
    xa, mia, ga, ea, cha, posta = side_a
    xb, mib, gb, eb, chb, postb = side_b
    C_ = dy.shape[1]
    V = dy.numel() // C_
    dev = dy.device
    fwa, fwb = fwd_chains if fwd_chains is not None else (None, None)
    ws = torch.empty(L.dpi_join_bwd_ws_doubles(C_, V), dtype=torch.float64, device=dev)
    coef = torch.empty(C_ * 8, dtype=torch.float32, device=dev)
    dgb = torch.empty((6, C_), dtype=torch.float32, device=dev)
    dxa, dxb = torch.empty_like(dy), torch.empty_like(dy)
    if fork is not None:
        lo, hi, mif, gf, ef, postf, dxf = fork
        dgbf = torch.empty((2, hi - lo), dtype=torch.float32, device=dev)
        fa = [lo, hi, ptr(mif), ptr(gf), ptr(ef), postf]
    else:
        dxf = dgbf = None
        fa = [0, 0, None, None, None, 1.0]
    check(L.dpi_join_bwd(ptr(dy), ptr(t), ptr(mi), ptr(gamma), ptr(beta), pre_slope, C_, V,
                         ptr(xa), ptr(mia), ptr(ga), ptr(ea), ptr(cha), posta, ptr(xb), ptr(mib), ptr(gb), ptr(eb), ptr(chb), postb,
                         ptr(fwa), ptr(fwb), *fa, ptr(ws), ptr(coef), ptr(dxa), ptr(dxb), ptr(dxf), ptr(dgb), ptr(dgbf), _io(xa, dy), stream()),
          "dpi_join_bwd")
    # (rows of one buffer: contiguous gradient vectors, no copies)
    f = None if dgbf is None else (dgbf[0], dgbf[1])
    return (dxa, dgb[2], dgb[3]), (dxb, dgb[4], dgb[5]), (dgb[0], dgb[1]), f


def _join_fused_ok(*tensors):
    """fp32 tensors, or — every tensor of the node alike — bf16 storage (the kernels widen on load and round on store)."""
    return JOIN_BWD_FUSED and (all(x.dtype == torch.float32 for x in tensors) or all(x.dtype == torch.bfloat16 for x in tensors))


def _join_forward(a, ch_a, b, ch_b, C_, V, slope, bn, gamma, beta, mi_out, y, keep_t):
    """t = T_a(a) + T_b(b);  y = bn(act(t)) written to `y` (a tensor or a channel slice of one).  Returns t, or None when it was not stored:
    with fp32 tensors and the two-pass join backward (keep_t False) the statistics pass only reads, the apply pass re-forms the sum from a
    and b, and the backward does the same — one tensor write less forward, two tensor reads less backward, C x V floats less held."""
    L = _lib.load()
    dev = a.device
    t = torch.empty_like(a) if keep_t else None
    nblk = L.dpi_stat_blocks(C_, V)
    part = torch.empty(nblk * C_ * 2, dtype=torch.float64, device=dev)
    check(L.dpi_chain_add_stats_io(ptr(a), ptr(ch_a), ptr(b), ptr(ch_b), C_, V, slope, ptr(t), ptr(part), _io(a), stream()),
          "dpi_chain_add_stats")
    ch_out = torch.empty(C_ * 5, dtype=torch.float32, device=dev)
    raw_bn_finalize(part, nblk, C_, V, gamma, beta, slope, bn.running_mean, bn.running_var, bn.num_batches_tracked, mi_out, ch_out,
                    act_first=1)
    if keep_t:
        raw_chain_apply(t, ch_out, C_, V, y)
    else:
        check(L.dpi_chain_add_apply(ptr(a), ptr(ch_a), ptr(b), ptr(ch_b), ptr(ch_out), C_, V, ptr(y), _io(a), stream()), "dpi_chain_add_apply")
    return t


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


# Gradient fan-in of an encoder output (round 6).  The output x of the encoder block of a U-Net level has two consumers — the level's ResPath
# (skip branch) and the stride-2 convolution that opens the deeper U (reference mulresunet.py:227-243) — so autograd used to ADD their two gradients
# in a pass of its own (four aten `add` launches per iteration, 1.26 GB moved for the finest level).  With a FanIn handle shared by the two nodes
# the stride-2 layer's backward-data ACCUMULATES into the ResPath's gradient instead (the kernels' `accumulate` flag: old + new in the epilogue,
# the same fp32 sum the add pass formed) and returns no gradient of its own; the tap node in front of the ResPath then hands autograd the total.
# Order (autograd runs ready nodes by descending creation order): SkipJoinFn.backward (last node of the level: first) publishes its dx, the deeper
# U's backward follows, its last node — ConvBnActFn.backward of the stride-2 layer — accumulates, SkipTapFn.backward (created before the deeper U:
# after it) returns the sum.  If the order ever differs (fan.dx not published yet), the node falls back to its own dx and autograd adds as before.
FAN_IN = os.environ.get("DPI_FAN_IN", "1") == "1"


class FanIn:
    __slots__ = ("dx", "branched")

    def __init__(self):
        self.dx = None
        self.branched = False


class ConvBnActFn(torch.autograd.Function):
    """conv -> BatchNorm(train) -> LeakyReLU(slope) in three launches: the conv's epilogue emits the {sum, sum^2}
    partials, dpi_bn_finalize turns them into the per-channel chain, dpi_chain_apply writes the activation.
    Backward: BN+act backward straight from (dy, raw conv output), then backward-data / backward-weight.
    The conv bias feeds a BatchNorm, so its gradient is analytically zero and returned as zeros."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, nbt, stride, slope, fan=None):
        x, w = _req(x, "conv input", act=True), _req(w, "conv weight")
        ctx.fan = fan
        adt = act_dtype() if x.ndim == 5 else torch.float32
        if x.ndim != 5 and _bf(x):
            raise _lib.DpiError("conv_bn_act: bf16 storage is built for the 3-D nets only")
        d = make_desc(x, w, stride, adt)
        Do, Ho, Wo = desc_out_dims(d)
        L = _lib.load()
        r = torch.empty(_like_spatial(x, d.Cout, Do, Ho, Wo), dtype=adt, device=x.device)
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
        ctx.d, ctx.slope, ctx.has_bias, ctx.bias_ref = d, float(slope), b is not None, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, r, gamma, beta, mi = ctx.saved_tensors
        d = ctx.d
        dr, dgamma, dbeta = _bn_backward(_req(dy, "conv-bn-act grad", act=True), r, mi, gamma, beta, 1.0, ctx.slope)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            conv_bwd_weight_async(d, x, None, dr, dw)
        if ctx.needs_input_grad[0]:
            fan = ctx.fan
            if fan is not None and fan.dx is not None and fan.dx.shape == x.shape and fan.dx.dtype == x.dtype:
                if fan.branched:
                    join_branch("skip")          # the ResPath's backward-data wrote fan.dx on the branch stream
                raw_conv_bwd_data(d, dr, w, fan.dx, accumulate=True)
                fan.dx = None                    # consumed: x's whole gradient now reaches autograd through the tap node (FanIn above)
            else:
                dx = torch.empty_like(x)
                raw_conv_bwd_data(d, dr, w, dx)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _zeros_like_or_none(ctx.bias_ref, dr)
        join_weight_grads()
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None


def _cba_raw(d, x, in_chain, w, b, bn, slope, r_out, mi_out, chain_out):
    """conv(T_in(x)) -> r_out with {sum, sum^2} epilogue -> finalize of `bn` (+ LeakyReLU slope) into (mi_out, chain_out)."""
    L = _lib.load()
    nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
    part = torch.empty(nblk * d.Cout * 2, dtype=torch.float64, device=x.device)
    raw_conv_fwd(d, x, in_chain, w, b, r_out, part)
    Do, Ho, Wo = desc_out_dims(d)
    raw_bn_finalize(part, nblk, d.Cout, Do * Ho * Wo, bn.weight, bn.bias, slope, bn.running_mean, bn.running_var,
                    bn.num_batches_tracked, mi_out, chain_out)


# Bisect knob (tools/snr_protocol_gpu.py --dead-bias; never set by the product path): what the ~3.3 k conv biases that feed a
# BatchNorm receive as "gradient".  "off" (default): None — they never move.  "sum": the per-channel sum of the pre-BatchNorm gradient, which is
# what the reference's autograd hands Adam (main.py:200,213): analytically zero, numerically the rounding residue of that sum.  "noise": N(0, 1e-6^2)
# — far above Adam's eps, so every dead bias takes full-size (~0.2 lr) random Adam steps: the upper bound of what the reference's residues can do.
DEAD_BIAS = "off"


def _zeros_like_or_none(b, dr=None):
    """Gradient of a conv bias that feeds a BatchNorm: identically zero — reported as None, which autograd and
    the optimisers treat as "no gradient" (torch.optim.Adam and FusedAdam skip such parameters; with zero moments the Adam
    update of a zero gradient is exactly 0, so the trajectory is the same).  Handing out cached zero tensors instead made
    autograd clone each of them every iteration (48 device-to-device copies).  (DEAD_BIAS above: the bisect variants.)"""
    if b is None or DEAD_BIAS == "off":
        return None
    if DEAD_BIAS == "noise":
        return torch.randn_like(b) * 1e-6
    return dr.sum(dim=[0] + list(range(2, dr.ndim)), dtype=torch.float32)


class Block3dFn(torch.autograd.Function):
    """Block3d (reference mulresunet.py:67-96) as ONE autograd node:
         y = bn2(act(CBA1(x) + bn1(cat[o1, o2, o3]))),  o1 = CBA3(x), o2 = CBA3(o1), o3 = CBA3(o2).
    The three 3x3x3 convs write their RAW outputs into channel slices of one tensor R (zero-copy concat); BN + LeakyReLU
    of each is a chain applied by whoever loads R (the next conv, the bn1 statistics, the residual join), bn1 is composed
    into that chain, and the join t = T_s(S) + T_bn1(R) emits the bn2 statistics in the same pass.  Backward accumulates
    the fan-in gradients inside conv backward-data instead of separate add passes.
    Elementwise traffic per block: 6 tensor passes forward instead of 15 (in units of out_dim x V floats)."""

    @staticmethod
    def forward(ctx, x, blk, slope, *p):
        x = _req(x, "block input", act=True)
        (w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, ws, bs, gs, es, gA, eA, gB, eB) = p
        L = _lib.load()
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        adt = act_dtype()                                    # storage type of R, S, t, y (and of their gradients in the backward)
        c1, c2, c3 = w1.shape[0], w2.shape[0], w3.shape[0]
        Ct = c1 + c2 + c3
        bn1_, bn2_, bn3_, bns_ = (m._parts()[1] for m in (blk.conv3x3, blk.conv5x5, blk.conv7x7, blk.shortcut))
        d1 = make_desc(x, w1, 1, adt)
        Do, Ho, Wo = desc_out_dims(d1)
        V = Do * Ho * Wo
        R = torch.empty(_like_spatial(x, Ct, Do, Ho, Wo), dtype=adt, device=dev)
        CH = torch.empty(Ct * 5, **f32)
        r1, r2, r3 = R[:, :c1], R[:, c1:c1 + c2], R[:, c1 + c2:]
        ch1, ch2, ch3 = CH[:c1 * 5], CH[c1 * 5:(c1 + c2) * 5], CH[(c1 + c2) * 5:]
        mi1, mi2, mi3 = (torch.empty(2 * c, **f32) for c in (c1, c2, c3))
        miA, miS, miB = (torch.empty(2 * Ct, **f32) for _ in range(3))
        dsc = make_desc(x, ws, 1, adt)
        S = torch.empty_like(R)
        chSA = torch.empty((2, Ct * 5), **f32)       # the two chains of the residual join, rows of one buffer (saved as a whole for the backward)
        chS, chA = chSA[0], chSA[1]
        side_shortcut = BRANCH_SHORTCUT if branch_on() else 0

        def shortcut_on_branch():   # the 1x1x1 shortcut (HBM-bound) beside the 3x3x3 chain (matrix-bound): it only needs the block input
            with _Branch("short", x, S, miS, chS):
                _cba_raw(dsc, x, None, ws, bs, bns_, slope, S, miS, chS)
        if side_shortcut == 1:
            shortcut_on_branch()
        _cba_raw(d1, x, None, w1, b1, bn1_, slope, r1, mi1, ch1)
        if side_shortcut == 2:      # ordered behind the first layer (which reads the same 64-67 channels at half the HBM bandwidth already)
            shortcut_on_branch()
        d2 = make_desc(r1, w2, 1, adt)
        _cba_raw(d2, r1, ch1, w2, b2, bn2_, slope, r2, mi2, ch2)
        d3 = make_desc(r2, w3, 1, adt)
        _cba_raw(d3, r2, ch2, w3, b3, bn3_, slope, r3, mi3, ch3)
        # bn1 over the virtual concat T_CH(R); chA = bn1 o T_CH
        A = blk.bn1
        raw_bn_stats_finalize(R, CH, Ct, V, gA, eA, 1.0, A.running_mean, A.running_var, A.num_batches_tracked, miA, chA,
                              compose=True)
        if side_shortcut:
            join_branch("short")
        else:
            _cba_raw(dsc, x, None, ws, bs, bns_, slope, S, miS, chS)
        # residual join + statistics of act(t) for bn2, then y = bn2(act(t))
        y = torch.empty_like(R)
        tless = _join_fused_ok(R, S)
        t = _join_forward(S, chS, R, chA, Ct, V, slope, blk.bn2, gB, eB, miB, y, keep_t=not tless)
        if tless:
            t = chSA                          # what the backward re-forms t from (2 x Ct x 5 floats instead of Ct x V)
        ctx.tless = tless
        ctx.save_for_backward(x, R, S, t, CH, mi1, mi2, mi3, miA, miS, miB, *[q for q in p if q is not None])
        ctx.none_mask = [q is None for q in p]
        ctx.descs = (d1, d2, d3, dsc)
        ctx.slope = float(slope)
        ctx.split = (c1, c2, c3)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "block grad", act=True)
        x, R, S, t, CH, mi1, mi2, mi3, miA, miS, miB = ctx.saved_tensors[:11]
        it = iter(ctx.saved_tensors[11:])
        p = [None if isnone else next(it) for isnone in ctx.none_mask]
        (w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, ws, bs, gs, es, gA, eA, gB, eB) = p
        d1, d2, d3, dsc = ctx.descs
        slope = ctx.slope
        c1, c2, c3 = ctx.split
        s1, s2, s3 = slice(0, c1), slice(c1, c1 + c2), slice(c1 + c2, c1 + c2 + c3)
        ch1, ch2 = CH[:c1 * 5], CH[c1 * 5:(c1 + c2) * 5]
        dR = torch.empty_like(R)
        if ctx.tless or _join_fused_ok(dy, t, S, R):
            # bn2, shortcut-BN, bn1 and conv7x7's BatchNorm (fork range = the last channel slice: nothing accumulates into it afterwards)
            # in two passes; dcat's last slice is not written (its consumer was the fork's apply pass)
            (dS, dgs, des), (dcat, dgA, deA), (dgB, deB), (dg3, de3) = _join_backward(
                dy, None if ctx.tless else t, miB, gB, eB, slope, (S, miS, gs, es, None, slope), (R, miA, gA, eA, CH, 1.0),
                fork=(c1 + c2, c1 + c2 + c3, mi3, g3, e3, slope, dR[:, s3]), fwd_chains=(t[0], t[1]) if ctx.tless else None)
        else:
            dt, dgB, deB, (redS, redA) = _bn_backward_fork(dy, t, miB, gB, eB, slope, 1.0,
                                                           [(S, miS, gs, es, None, slope), (R, miA, gA, eA, CH, 1.0)])
            # shortcut-BN and bn1 share dt: one pass; it also takes the phase-1 partials of conv7x7's BatchNorm (its incoming
            # gradient is bn1's dx on the last channel slice — nothing accumulates into that slice afterwards)
            (dS, dgs, des), (dcat, dgA, deA), red3 = _bn_backward_apply_dual(
                dt, (S, miS, gs, es, None, slope, redS), (R, miA, gA, eA, CH, 1.0, redA), fork=(c1 + c2, c1 + c2 + c3, mi3, g3, e3, slope))
            del dt
            # o3 -> o2 -> o1: each conv's backward-data ACCUMULATES into the concat gradient of its input slice
            _, dg3, de3 = _bn_backward_apply(dcat[:, s3], R[:, s3], mi3, g3, e3, 1.0, slope, red3, dx=dR[:, s3])
        # the shortcut's weight gradient (1x1x1: HBM-bound) only needs dS: issued FIRST, it runs beside the matrix-bound 3x3x3 chain below
        # instead of beside conv1's weight gradient at the end of the node (round 5; DPI_EARLY_SHORTCUT_WGRAD=0: the old place)
        dws = torch.empty_like(ws)
        if EARLY_SHORTCUT_WGRAD:
            conv_bwd_weight_async(dsc, x, None, dS, dws)
        dw3 = torch.empty_like(w3)
        conv_bwd_weight_async(d3, R[:, s2], ch2, dR[:, s3], dw3)
        raw_conv_bwd_data(d3, dR[:, s3], w3, dcat[:, s2], accumulate=True)
        _, dg2, de2 = _bn_backward(dcat[:, s2], R[:, s2], mi2, g2, e2, 1.0, slope, dx=dR[:, s2])
        dw2 = torch.empty_like(w2)
        conv_bwd_weight_async(d2, R[:, s1], ch1, dR[:, s2], dw2)
        raw_conv_bwd_data(d2, dR[:, s2], w2, dcat[:, s1], accumulate=True)
        _, dg1, de1 = _bn_backward(dcat[:, s1], R[:, s1], mi1, g1, e1, 1.0, slope, dx=dR[:, s1])
        dw1 = torch.empty_like(w1)
        conv_bwd_weight_async(d1, x, None, dR[:, s1], dw1)
        if not EARLY_SHORTCUT_WGRAD:
            conv_bwd_weight_async(dsc, x, None, dS, dws)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw_conv_bwd_data_dual(d1, dR[:, s1], w1, dsc, dS, ws, dx)
        join_weight_grads()
        z = _zeros_like_or_none      # conv biases feed a BatchNorm: analytically zero gradient
        return (dx, None, None, dw1, z(b1, dR[:, s1]), dg1, de1, dw2, z(b2, dR[:, s2]), dg2, de2, dw3, z(b3, dR[:, s3]), dg3, de3,
                dws, z(bs, dS), dgs, des, dgA, deA, dgB, deB)


def _respath_bn_backward(dy, t, miB, gB, eB, slope, r3, mi3, g3, e3, r1, mi1, g1, e1, tless=False):
    """BatchNorm backward of y = bn(act(act(bn3(r3)) + act(bn1(r1)))): (dr3, dg3, de3), (dr1, dg1, de1), (dgB, deB).
    tless: `t` holds the two forward chains [ch1, ch3] instead of the sum (the forward join did not store it)."""
    if tless or _join_fused_ok(dy, t, r3, r1):
        # (side A = r3, side B = r1; the forward join formed t = T_ch1(r1) + T_ch3(r3))
        a, b, top, _ = _join_backward(dy, None if tless else t, miB, gB, eB, slope, (r3, mi3, g3, e3, None, slope), (r1, mi1, g1, e1, None, slope),
                                      fwd_chains=(t[1], t[0]) if tless else None)
        return a, b, top
    dt, dgB, deB, (red3, red1) = _bn_backward_fork(dy, t, miB, gB, eB, slope, 1.0,
                                                   [(r3, mi3, g3, e3, None, slope), (r1, mi1, g1, e1, None, slope)])
    a, b, _ = _bn_backward_apply_dual(dt, (r3, mi3, g3, e3, None, slope, red3), (r1, mi1, g1, e1, None, slope, red1))
    return a, b, (dgB, deB)


class ResPath3dFn(torch.autograd.Function):
    """ResPath3d (reference mulresunet.py:99-113) as one autograd node: y = bn(act(CBA1(x) + CBA3(x)))."""

    @staticmethod
    def forward(ctx, x, rp, slope, *p):
        x = _req(x, "respath input", act=True)
        (w3, b3, g3, e3, w1, b1, g1, e1, gB, eB) = p
        L = _lib.load()
        f32 = dict(dtype=torch.float32, device=x.device)
        adt = act_dtype()
        Ct = w3.shape[0]
        bn3_, bn1_ = rp.conv3x3._parts()[1], rp.conv1x1._parts()[1]
        d3, d1 = make_desc(x, w3, 1, adt), make_desc(x, w1, 1, adt)
        Do, Ho, Wo = desc_out_dims(d3)
        V = Do * Ho * Wo
        r3 = torch.empty(_like_spatial(x, Ct, Do, Ho, Wo), dtype=adt, device=x.device)
        r1 = torch.empty_like(r3)
        mi3, mi1, miB = (torch.empty(2 * Ct, **f32) for _ in range(3))
        ch13 = torch.empty((2, 5 * Ct), **f32)
        ch1, ch3 = ch13[0], ch13[1]
        _cba_raw(d3, x, None, w3, b3, bn3_, slope, r3, mi3, ch3)
        _cba_raw(d1, x, None, w1, b1, bn1_, slope, r1, mi1, ch1)
        y = torch.empty_like(r3)
        tless = _join_fused_ok(r3, r1)
        t = _join_forward(r1, ch1, r3, ch3, Ct, V, slope, rp.bn, gB, eB, miB, y, keep_t=not tless)
        if tless:
            t = ch13
        ctx.tless = tless
        ctx.save_for_backward(x, r3, r1, t, mi3, mi1, miB, *[q for q in p if q is not None])
        ctx.none_mask = [q is None for q in p]
        ctx.descs = (d3, d1)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "respath grad", act=True)
        x, r3, r1, t, mi3, mi1, miB = ctx.saved_tensors[:7]
        it = iter(ctx.saved_tensors[7:])
        (w3, b3, g3, e3, w1, b1, g1, e1, gB, eB) = [None if isnone else next(it) for isnone in ctx.none_mask]
        d3, d1 = ctx.descs
        slope = ctx.slope
        (dr3, dg3, de3), (dr1, dg1, de1), (dgB, deB) = _respath_bn_backward(dy, t, miB, gB, eB, slope, r3, mi3, g3, e3, r1, mi1, g1, e1, ctx.tless)
        dw3, dw1 = torch.empty_like(w3), torch.empty_like(w1)
        conv_bwd_weight_async(d3, x, None, dr3, dw3)
        conv_bwd_weight_async(d1, x, None, dr1, dw1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            raw_conv_bwd_data_dual(d3, dr3, w3, d1, dr1, w1, dx)
        join_weight_grads()
        z = _zeros_like_or_none
        return dx, None, None, dw3, z(b3, dr3), dg3, de3, dw1, z(b1, dr1), dg1, de1, dgB, deB


def _skip_alloc(x, p, Cd):
    """Tensors of one level join (ResPath3d(x) -> cat[:, :Cs], up-sampled deep branch -> cat[:, Cs:]), allocated by the CURRENT stream."""
    (w3, b3, g3, e3, w1, b1, g1, e1, gB, eB) = p
    f32 = dict(dtype=torch.float32, device=x.device)
    adt = act_dtype()
    Cs = w3.shape[0]
    d3, d1 = make_desc(x, w3, 1, adt), make_desc(x, w1, 1, adt)
    Do, Ho, Wo = desc_out_dims(d3)
    T = dict(adt=adt, Cs=Cs, Cd=Cd, d3=d3, d1=d1, dims=(Do, Ho, Wo))
    T["r3"] = torch.empty(_like_spatial(x, Cs, Do, Ho, Wo), dtype=adt, device=x.device)
    T["r1"] = torch.empty_like(T["r3"])
    T["tless"] = _join_fused_ok(T["r3"])
    T["mi3"], T["mi1"], T["miB"] = (torch.empty(2 * Cs, **f32) for _ in range(3))
    T["ch13"] = torch.empty((2, 5 * Cs), **f32)
    T["ch1"], T["ch3"], T["chB"] = T["ch13"][0], T["ch13"][1], torch.empty(5 * Cs, **f32)
    # the join's sum t, or — when it is not stored — the two chains it is re-formed from ([ch1, ch3], filled by _skip_launch)
    T["t"] = T["ch13"] if T["tless"] else torch.empty_like(T["r3"])
    T["cat"] = torch.empty(_like_spatial(x, Cs + Cd, Do, Ho, Wo), dtype=adt, device=x.device)
    return T


def _skip_tensors(T):
    return [T[k] for k in ("r3", "r1", "t", "mi3", "mi1", "miB", "ch13", "chB", "cat")]


def _skip_launch(x, rp, slope, p, T):
    """The ResPath half of the join on the current stream: conv3x3 / conv1x1 with statistics, residual join + statistics, bn -> cat[:, :Cs]."""
    (w3, b3, g3, e3, w1, b1, g1, e1, gB, eB) = p
    L = _lib.load()
    Cs = T["Cs"]
    Do, Ho, Wo = T["dims"]
    V = Do * Ho * Wo
    bn3_, bn1_ = rp.conv3x3._parts()[1], rp.conv1x1._parts()[1]
    _cba_raw(T["d3"], x, None, w3, b3, bn3_, slope, T["r3"], T["mi3"], T["ch3"])
    _cba_raw(T["d1"], x, None, w1, b1, bn1_, slope, T["r1"], T["mi1"], T["ch1"])
    nblk = L.dpi_stat_blocks(Cs, V)
    part = torch.empty(nblk * Cs * 2, dtype=torch.float64, device=x.device)
    tless = T["tless"]
    check(L.dpi_chain_add_stats_io(ptr(T["r1"]), ptr(T["ch1"]), ptr(T["r3"]), ptr(T["ch3"]), Cs, V, slope, None if tless else ptr(T["t"]),
                                   ptr(part), _io(T["r3"]), stream()), "dpi_chain_add_stats")
    B = rp.bn
    raw_bn_finalize(part, nblk, Cs, V, gB, eB, slope, B.running_mean, B.running_var, B.num_batches_tracked, T["miB"], T["chB"],
                    act_first=1)
    if tless:
        check(L.dpi_chain_add_apply(ptr(T["r1"]), ptr(T["ch1"]), ptr(T["r3"]), ptr(T["ch3"]), ptr(T["chB"]), Cs, V, ptr(T["cat"][:, :Cs]),
                                    _io(T["r3"]), stream()), "dpi_chain_add_apply")
    else:
        raw_chain_apply(T["t"], T["chB"], Cs, V, T["cat"][:, :Cs])


def skip_begin(x, rp, slope, Cd):
    """Start the ResPath half of a level join on the branch stream BEFORE the deeper U runs (it only needs the encoder output x).
    Returns the handle skip_join() takes, or None when the serial schedule applies."""
    if not (branch_on() and BRANCH_SKIP) or x.ndim != 5:
        return None
    x = _req(x, "skip input", act=True)
    p = _cba_params(rp.conv3x3) + _cba_params(rp.conv1x1) + [rp.bn.weight, rp.bn.bias]
    with torch.no_grad():
        T = _skip_alloc(x, p, Cd)
        with _Branch("skip", x, *_skip_tensors(T)):
            _skip_launch(x, rp, slope, p, T)
    return T


class SkipTapFn(torch.autograd.Function):
    """Identity in front of a level join whose ResPath half runs on the branch stream.  Created BEFORE the deeper U's nodes, so the
    autograd engine (highest sequence number first) runs its backward AFTER the deeper U's backward: that is where the main stream
    waits for the branch stream's ResPath backward — right before the encoder block of the level needs the gradient."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        join_branch("skip")
        return g


def skip_tap(x):
    return SkipTapFn.apply(x)


class SkipJoinFn(torch.autograd.Function):
    """One U-Net level join of the 3-D MultiRes-UNet as a single node (reference mulresunet.py:227-243 + Concat3D):
         cat[:, :Cs] = ResPath3d(x)            cat[:, Cs:] = Upsample(x2)(deep), cropped to x's spatial size
    Both producers write straight into the concat buffer (no crop-copy pass); the backward hands the two channel slices
    of d(cat) to the ResPath backward and the up-sampling adjoint.  `pre`: the handle of skip_begin() — the ResPath half is
    already running on the branch stream (its input must then come through skip_tap())."""

    @staticmethod
    def forward(ctx, x, deep, rp, slope, linear, pre, fan, *p):
        x, deep = _req(x, "skip input", act=True), _req(deep, "deep input", act=True)
        ctx.fan = fan
        L = _lib.load()
        Cd, Dd, Hd, Wd = _dims(deep)
        if pre is None:
            T = _skip_alloc(x, p, Cd)
        else:
            T = pre
            if T["Cd"] != Cd or T["adt"] != act_dtype():
                raise _lib.DpiError("skip_join: the branch-stream half was started for another deep branch / storage type")
        adt, Cs = T["adt"], T["Cs"]
        if deep.dtype != adt:
            raise _lib.DpiError("skip_join: the up-sampled branch is %s, this node stores %s (storage mode switched between nodes?)" % (deep.dtype, adt))
        Do, Ho, Wo = T["dims"]
        if not (Do <= 2 * Dd and Ho <= 2 * Hd and Wo <= 2 * Wd):
            raise _lib.DpiError("skip_join: the up-sampled branch is smaller than the skip branch")
        if pre is None:
            _skip_launch(x, rp, slope, p, T)
        cat = T["cat"]
        check(L.dpi_upsample2x_fwd_io(ptr(deep), None, Cd, Dd, Hd, Wd, Do, Ho, Wo, int(linear), ptr(cat[:, Cs:]), _io(deep), stream()),
              "dpi_upsample2x_fwd")
        if pre is not None:
            join_branch("skip")          # the two producers wrote disjoint channel slices; the consumer needs both
        ctx.save_for_backward(x, T["r3"], T["r1"], T["t"], T["mi3"], T["mi1"], T["miB"], *[q for q in p if q is not None])
        ctx.none_mask = [q is None for q in p]
        ctx.descs = (T["d3"], T["d1"])
        ctx.slope = float(slope)
        ctx.up = (Cs, Cd, Dd, Hd, Wd, Do, Ho, Wo, int(linear), deep.shape, deep.dtype)
        ctx.branched = pre is not None
        ctx.tless = T["tless"]
        return cat

    @staticmethod
    def backward(ctx, dcat):
        dcat = _req(dcat, "skip-join grad", act=True)
        x, r3, r1, t, mi3, mi1, miB = ctx.saved_tensors[:7]
        it = iter(ctx.saved_tensors[7:])
        (w3, b3, g3, e3, w1, b1, g1, e1, gB, eB) = [None if isnone else next(it) for isnone in ctx.none_mask]
        d3, d1 = ctx.descs
        slope = ctx.slope
        Cs, Cd, Dd, Hd, Wd, Do, Ho, Wo, linear, deep_shape, deep_dtype = ctx.up
        ddeep = None
        if ctx.needs_input_grad[1]:
            ddeep = torch.empty(deep_shape, dtype=deep_dtype, device=dcat.device)
            raw_upsample2x_bwd(dcat[:, Cs:], Cd, Dd, Hd, Wd, Do, Ho, Wo, linear, ddeep)

        def respath_backward():
            (dr3, dg3, de3), (dr1, dg1, de1), (dgB, deB) = _respath_bn_backward(dcat[:, :Cs], t, miB, gB, eB, slope,
                                                                                  r3, mi3, g3, e3, r1, mi1, g1, e1, ctx.tless)
            dw3, dw1 = torch.empty_like(w3), torch.empty_like(w1)
            conv_bwd_weight_async(d3, x, None, dr3, dw3)
            conv_bwd_weight_async(d1, x, None, dr1, dw1)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                raw_conv_bwd_data_dual(d3, dr3, w3, d1, dr1, w1, dx)
            return dx, dw3, dg3, de3, dw1, dg1, de1, dgB, deB, _zeros_like_or_none(b3, dr3), _zeros_like_or_none(b1, dr1)

        if ctx.branched and branch_on() and BRANCH_SKIP_BWD:
            # the deeper U's backward (critical path, main stream) does not wait for this: SkipTapFn.backward joins, just before the
            # encoder block of the level consumes dx.  Tensors allocated in here come from the branch stream's pool.
            with _Branch("skip", dcat, x, r3, r1, t, mi3, mi1, miB):
                dx, dw3, dg3, de3, dw1, dg1, de1, dgB, deB, db3, db1 = respath_backward()
            _defer(dg3, de3, dg1, de1, dgB, deB, db3, db1)
        else:
            dx, dw3, dg3, de3, dw1, dg1, de1, dgB, deB, db3, db1 = respath_backward()
        join_weight_grads()
        if ctx.fan is not None and dx is not None:
            ctx.fan.dx, ctx.fan.branched = dx, bool(ctx.branched and branch_on() and BRANCH_SKIP_BWD)
        return dx, ddeep, None, None, None, None, None, dw3, db3, dg3, de3, dw1, db1, dg1, de1, dgB, deB


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


ACT_KINDS = {"ELU": 1, "Tanh": 2, "Sigmoid": 3}


class ActFn(torch.autograd.Function):
    """nn.ELU() / nn.Tanh() / nn.Sigmoid() (reference base.py:104-112); the output is what the backward needs."""

    @staticmethod
    def forward(ctx, x, kind):
        x = _req(x, "activation input")
        y = torch.empty_like(x)
        check(_lib.load().dpi_act_fwd(ptr(x), x.numel(), kind, ptr(y), stream()), "dpi_act_fwd")
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _req(dy, "activation grad")
        dx = torch.empty_like(dy)
        check(_lib.load().dpi_act_bwd(ptr(dy), ptr(y), dy.numel(), ctx.kind, ptr(dx), stream()), "dpi_act_bwd")
        return dx, None


class ChannelScaleFn(torch.autograd.Function):
    """y[c] = s[c] * x[c] (channel dropout: s = Bernoulli mask / (1 - p)); linear, so the backward is the same pass on dy."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _req(x, "channel-scale input")
        C_ = x.shape[1]
        chain = torch.zeros(C_, 5, dtype=torch.float32, device=x.device)
        chain[:, 0] = 1.0
        chain[:, 2] = 1.0
        chain[:, 3] = scale
        y = torch.empty_like(x)
        raw_chain_apply(x, chain, C_, x.numel() // C_, y)
        ctx.save_for_backward(chain)
        return y

    @staticmethod
    def backward(ctx, dy):
        (chain,) = ctx.saved_tensors
        dy = _req(dy, "channel-scale grad")
        C_ = dy.shape[1]
        dx = torch.empty_like(dy)
        raw_chain_apply(dy, chain, C_, dy.numel() // C_, dx)
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
        raw_upsample2x_bwd(dy, C_, D, H, W, Do, Ho, Wo, linear, dx)
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


def conv_bn_act(x, w, b, gamma, beta, running_mean, running_var, nbt, stride=1, slope=0.2, fan=None):
    return ConvBnActFn.apply(x, w, b, gamma, beta, running_mean, running_var, nbt, int(stride), float(slope), fan)


def _cba_params(m):
    conv_m, bn, _ = m._parts()
    return [conv_m.weight, conv_m.bias, bn.weight, bn.bias]


def block3d(x, blk, slope):
    """fused Block3d; `blk` is the MultiResBlock module (parameters are passed explicitly so autograd tracks them)."""
    p = (_cba_params(blk.conv3x3) + _cba_params(blk.conv5x5) + _cba_params(blk.conv7x7) + _cba_params(blk.shortcut)
         + [blk.bn1.weight, blk.bn1.bias, blk.bn2.weight, blk.bn2.bias])
    return Block3dFn.apply(x, blk, slope, *p)


def respath3d(x, rp, slope):
    p = _cba_params(rp.conv3x3) + _cba_params(rp.conv1x1) + [rp.bn.weight, rp.bn.bias]
    return ResPath3dFn.apply(x, rp, slope, *p)


def skip_join(x, deep, rp, slope, mode, pre=None, fan=None):
    """cat[ResPath3d(x), Upsample(deep)] written in place (zero-copy concat).  pre: handle of skip_begin(x, ...) (x through skip_tap).
    fan: the FanIn handle this node shares with the stride-2 layer of the deeper U (x through skip_tap as well)."""
    p = _cba_params(rp.conv3x3) + _cba_params(rp.conv1x1) + [rp.bn.weight, rp.bn.bias]
    return SkipJoinFn.apply(x, deep, rp, slope, mode != "nearest", pre, fan, *p)


def leaky_relu(x, slope=0.2):
    return LeakyReLUFn.apply(x, float(slope))


def channel_dropout(x, p):
    """nn.Dropout2d / Dropout3d in training mode (the reference never leaves it): whole channels are zeroed with
    probability p, the rest scaled by 1/(1-p).  The mask comes from torch's device generator (not the reference's stream)."""
    keep = (torch.rand(x.shape[1], device=x.device) >= p).to(torch.float32) / (1.0 - p)
    return ChannelScaleFn.apply(x, keep)


def activation(x, name):
    return ActFn.apply(x, ACT_KINDS[name])


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
