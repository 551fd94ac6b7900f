"""GPU parity of the leaf ops: HIP kernels (through the C ABI / ctypes / autograd wrappers) against
 (1) the golden vectors recorded from the reference, (2) the CPU oracle on seeded random inputs,
 (3) size-independent properties (adjoint dot-tests, linearity) at BASELINE sizes.
Tolerances: fp32; forward rtol 1e-5..1e-4, gradients norm-wise 2e-4 (different summation order)."""
import numpy as np
import pytest
import torch

from oracle import dpi_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def G(a, grad=False):
    t = torch.from_numpy(np.array(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(True) if grad else t


def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


@pytest.fixture(scope="module")
def ops():
    from deep_prior_interpolation_amd import ops as _ops
    return _ops


CONVS = ["conv3d_k3s1", "conv3d_k3s1_wide", "conv3d_k3s2_odd", "conv3d_k3s2_even", "conv3d_k1",
         "conv2d_k3s1", "conv2d_k3s2", "conv2d_k1"]


@pytest.mark.parametrize("name", CONVS)
def test_conv_golden(golden, ops, name):
    g = golden("ops")[name]
    stride = 2 if "s2" in name else 1
    x, w, b = G(g["x"], True), G(g["state"]["0.weight"], True), G(g["state"]["0.bias"], True)
    y = ops.conv(x, w, b, stride)
    close(y, g["y"], 1e-5, 2e-5, "y")
    y.backward(G(g["dy"]))
    close(x.grad, g["dx"], 1e-5, 2e-5, "dx")
    close(w.grad, g["grads"]["0.weight"], 1e-5, 5e-5, "dw")
    close(b.grad, g["grads"]["0.bias"], 1e-5, 5e-5, "db")


CONV_RANDOM = [
    # Cin, Cout, (D,H,W), k, stride
    (64, 4, (16, 20, 36), 3, 1), (4, 8, (9, 17, 33), 3, 1), (8, 13, (8, 8, 32), 3, 1), (25, 16, (12, 9, 40), 3, 1),
    (67, 4, (6, 10, 34), 3, 1), (25, 1, (8, 16, 32), 3, 1), (35, 71, (8, 8, 8), 3, 1), (71, 142, (4, 4, 4), 3, 1),
    (25, 25, (16, 16, 32), 3, 2), (51, 51, (9, 11, 13), 3, 2), (3, 5, (2, 2, 2), 3, 2), (5, 3, (1, 1, 1), 3, 1),
    (64, 25, (8, 16, 32), 1, 1), (137, 51, (5, 7, 9), 1, 1), (554, 35, (4, 4, 4), 1, 1), (25, 16, (8, 8, 33), 1, 1),
    # MFMA paths: small-tile variants (few voxels), stride 2 with >= 8 channels, odd sizes, channel tails
    (12, 9, (7, 9, 11), 3, 2), (9, 20, (8, 8, 40), 3, 2), (105, 64, (4, 8, 16), 3, 1), (17, 26, (6, 6, 6), 3, 1),
    (16, 16, (3, 5, 17), 3, 1), (9, 33, (12, 16, 64), 3, 1), (212, 212, (2, 2, 2), 3, 2),
    # few-output-channel backward-weight MFMA kernel (Cout <= 5, >= 32768 voxels), channel / size tails
    (13, 4, (32, 32, 40), 3, 1), (7, 1, (33, 31, 37), 3, 1), (6, 5, (32, 32, 32), 3, 1), (64, 4, (16, 48, 64), 3, 1),
    # few-output-channel FORWARD MFMA kernel ((co, kw) rows, 46-column tiles): Cout 1..4, ragged W / H / D tile edges
    (25, 1, (33, 31, 50), 3, 1), (9, 3, (32, 33, 47), 3, 1), (12, 2, (34, 17, 93), 3, 1), (67, 4, (30, 24, 46), 3, 1),
]


@pytest.mark.parametrize("cin,cout,shape,k,stride", CONV_RANDOM)
def test_conv_vs_oracle(ops, cin, cout, shape, k, stride):
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, k, k, k), generator=gen) * (1.0 / np.sqrt(cin * k ** 3))
    b = torch.randn(cout, generator=gen)
    # fp64 oracle: only the GPU's own fp32 rounding is measured (an fp32 CPU reference is itself ~1e-6 off, box-dependent)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = O.conv_nd(xr, wr, br, stride)
    dy = torch.randn(yr.shape, generator=gen)
    yr.backward(dy.double())
    xg, wg, bg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.conv(xg, wg, bg, stride)
    assert rel(y, yr) < 2e-6
    y.backward(dy.to(DEV))
    assert rel(xg.grad, xr.grad) < 2e-6
    assert rel(wg.grad, wr.grad) < 5e-6
    assert rel(bg.grad, br.grad) < 5e-6


@pytest.mark.parametrize("cin,cout,shape,k,stride", [(5, 7, (1, 37, 41), 3, 1), (6, 6, (1, 33, 30), 3, 2), (9, 4, (1, 20, 24), 1, 1),
                                                     (9, 12, (1, 33, 30), 3, 2), (13, 17, (1, 70, 45), 3, 1), (16, 8, (1, 9, 12), 3, 1),
                                                     (10, 10, (1, 64, 64), 3, 2),
                                                     # few output channels in 2-D: swapped-orientation backward-weight MFMA path
                                                     (16, 4, (1, 40, 48), 3, 1), (24, 1, (1, 33, 50), 3, 1), (67, 4, (1, 64, 64), 3, 1),
                                                     (137, 51, (1, 16, 16), 1, 1), (300, 9, (1, 8, 8), 1, 1)])
def test_conv2d_vs_oracle(ops, cin, cout, shape, k, stride):
    gen = torch.Generator().manual_seed(7)
    x = torch.randn((1, cin) + shape[1:], generator=gen)
    w = torch.randn((cout, cin, k, k), generator=gen) * 0.2
    b = torch.randn(cout, generator=gen)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = O.conv_nd(xr, wr, br, stride)
    dy = torch.randn(yr.shape, generator=gen)
    yr.backward(dy.double())
    xg, wg, bg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.conv(xg, wg, bg, stride)
    assert rel(y, yr) < 2e-6
    y.backward(dy.to(DEV))
    assert rel(xg.grad, xr.grad) < 2e-6 and rel(wg.grad, wr.grad) < 5e-6 and rel(bg.grad, br.grad) < 5e-6


@pytest.mark.parametrize("cin,cout,shape", [(13, 9, (7, 12, 37)), (20, 4, (32, 33, 47)), (25, 16, (12, 16, 40)), (9, 3, (40, 30, 50))])
def test_conv_fused_chain_and_stats(ops, cin, cout, shape):
    """conv(T(x)) with the BN-apply+LeakyReLU chain fused into the load, and the {sum, sum^2} epilogue
    (MFMA, few-output-channel MFMA and VALU kernels)."""
    import ctypes as C
    from deep_prior_interpolation_amd import _lib
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((1, cin) + shape, generator=gen)
    chain = torch.stack([torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen)], dim=1).contiguous()
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.1
    b = torch.randn(cout, generator=gen)
    bc = lambda v: v.reshape(1, -1, 1, 1, 1)
    tx = bc(chain[:, 3]) * O.activation("LeakyReLU", bc(chain[:, 0]) * x + bc(chain[:, 1])) + bc(chain[:, 4])
    yr = O.conv_nd(tx, w, b, 1)
    L = _lib.load()
    xg, wg, bg, cg = x.to(DEV), w.to(DEV), b.to(DEV), chain.to(DEV)
    d = ops.make_desc(xg, wg, 1)
    nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
    part = torch.zeros(nblk * cout * 2, dtype=torch.float64, device=DEV)
    y = torch.empty(yr.shape, device=DEV)
    ops.raw_conv_fwd(d, xg, cg, wg, bg, y, part)
    assert rel(y, yr) < 2e-6
    p = part.view(nblk, cout, 2).sum(0).cpu()
    yr64 = yr.double()
    np.testing.assert_allclose(p[:, 0].numpy(), yr64.sum((0, 2, 3, 4)).numpy(), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(p[:, 1].numpy(), (yr64 ** 2).sum((0, 2, 3, 4)).numpy(), rtol=1e-5)
    # backward-weight with the same chain on x
    dy = torch.randn(yr.shape, generator=gen)
    txr = tx.clone().requires_grad_(False)
    wr = w.clone().requires_grad_(True)
    O.conv_nd(txr, wr, None, 1).backward(dy)
    dw = torch.empty_like(wg)
    ops.raw_conv_bwd_weight(d, xg, cg, dy.to(DEV), dw)
    assert rel(dw, wr.grad) < 5e-6


SPLIT_CASES = [(140, 35, (8, 16, 16), 1), (554, 35, (4, 8, 16), 1), (71, 20, (5, 7, 9), 1), (212, 71, (4, 4, 8), 1),
               (100, 24, (8, 8, 16), 2), (133, 17, (3, 9, 11), 2), (77, 9, (1, 8, 16), 1)]


@pytest.mark.parametrize("cin,cout,shape,stride", SPLIT_CASES)
def test_conv_input_channel_split_vs_oracle(ops, cin, cout, shape, stride):
    """Coarse-level launches (few output tiles, many input channels) split the input-channel loop over blockIdx.z into a workspace
    and sum the partial outputs in a fixed order (dpi_conv_fwd_ws / dpi_conv_bwd_data_ws): forward with chain, bias and the
    statistics epilogue, backward-data with fan-in, against the fp64 oracle; the unsplit launch of the same problem
    (dpi_set_option("splitk", 0), and the entry points without a workspace) agrees to rounding; two runs are bit-identical."""
    import ctypes as C
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    gen = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    chain = torch.stack([torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen)], dim=1).contiguous()
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * (1.0 / np.sqrt(cin * 27))
    b = torch.randn(cout, generator=gen)
    bc = lambda v: v.reshape(1, -1, 1, 1, 1)
    tx = bc(chain[:, 3]) * O.activation("LeakyReLU", bc(chain[:, 0]) * x + bc(chain[:, 1])) + bc(chain[:, 4])
    txr = tx.double().requires_grad_(True)
    yr = O.conv_nd(txr, w.double(), b.double(), stride)
    dy = torch.randn(yr.shape, generator=gen)
    yr.backward(dy.double())
    xg, wg, bg, cg, dyg = x.to(DEV), w.to(DEV), b.to(DEV), chain.to(DEV), dy.to(DEV)
    d = ops.make_desc(xg, wg, stride)
    assert L.dpi_conv_fwd_ws_floats(C.byref(d)) > 0, "case does not exercise the split"
    nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))

    def fwd():
        part = torch.zeros(nblk * cout * 2, dtype=torch.float64, device=DEV)
        y = torch.full(yr.shape, float("nan"), device=DEV)
        ops.raw_conv_fwd(d, xg, cg, wg, bg, y, part)
        return y, part
    y, part = fwd()
    assert rel(y, yr) < 2e-6
    p = part.view(nblk, cout, 2).sum(0).cpu()
    y64 = yr.detach()
    np.testing.assert_allclose(p[:, 0].numpy(), y64.sum((0, 2, 3, 4)).numpy(), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(p[:, 1].numpy(), (y64 ** 2).sum((0, 2, 3, 4)).numpy(), rtol=1e-5)
    y2, part2 = fwd()
    assert torch.equal(y, y2) and torch.equal(part, part2)
    # backward-data (stride 1: the flipped launch splits over the OUTPUT channels of the layer), plain and with fan-in
    dxr = None
    if stride == 1:
        xr2 = x.double().requires_grad_(True)
        O.conv_nd(xr2, w.double(), None, 1).backward(dy.double())
        dxr = xr2.grad
        dx = torch.full(x.shape, float("nan"), device=DEV)
        ops.raw_conv_bwd_data(d, dyg, wg, dx)
        assert rel(dx, dxr) < 2e-6
        base = torch.randn(x.shape, generator=gen)
        dx2 = base.to(DEV).clone()
        ops.raw_conv_bwd_data(d, dyg, wg, dx2, accumulate=True)
        assert rel(dx2, dxr + base.double()) < 2e-6
    # the same launches without the split
    L.set_option("splitk", 0)
    try:
        assert L.dpi_conv_fwd_ws_floats(C.byref(d)) == 0
        y0, part0 = fwd()
        if dxr is not None:
            dx0 = torch.empty(x.shape, device=DEV)
            ops.raw_conv_bwd_data(d, dyg, wg, dx0)
            assert rel(dx0, dxr) < 2e-6
    finally:
        L.set_option("splitk", 1)
    assert rel(y0, yr) < 2e-6 and rel(y, y0) < 2e-6
    np.testing.assert_allclose(part0.view(nblk, cout, 2).sum(0).cpu().numpy(), p.numpy(), rtol=1e-6, atol=1e-3)
    # an ABI-300 caller (no workspace) gets the unsplit launch
    y3 = torch.empty(yr.shape, device=DEV)
    _lib.check(L.dpi_conv_fwd(C.byref(d), _lib.ptr(xg), _lib.ptr(cg), _lib.ptr(wg), _lib.ptr(bg), _lib.ptr(y3), None, _lib.stream()), "dpi_conv_fwd")
    assert torch.equal(y3, y0)


DUAL_CASES = [(67, 4, 25, (12, 16, 40)), (25, 16, 16, (8, 16, 32)), (25, 16, 16, (33, 31, 50)), (137, 8, 51, (6, 10, 18)),
              (51, 32, 32, (16, 16, 64)), (8, 13, 9, (8, 16, 32)), (105, 64, 64, (4, 6, 10)), (13, 9, 7, (1, 20, 24)),
              (67, 4, 25, (40, 48, 64)), (554, 35, 212, (4, 8, 16))]


@pytest.mark.parametrize("cin,c3,c1,shape", DUAL_CASES)
def test_conv_bwd_data_dual_vs_oracle(ops, cin, c3, c1, shape):
    """dpi_conv_bwd_data_dual: input gradient of a 3x3(x3) layer and a 1x1(x1) layer reading the same tensor (Block3d conv1 + shortcut,
    ResPath3d) in one pass — fused into the MFMA stencil kernel's epilogue where that kernel serves the 3x3(x3) layer, two launches
    elsewhere (few-channel 4x4x1 kernel, input-channel split, 2-D big tiles): against the fp64 oracle, with and without fan-in, and
    against the two separate launches."""
    import ctypes as C
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    gen = torch.Generator().manual_seed(cin + 31 * c3 + c1)
    kd = 1 if shape[0] == 1 else 3
    w3 = torch.randn((c3, cin, kd, 3, 3), generator=gen) * (1.0 / np.sqrt(c3 * 9 * kd))
    w1 = torch.randn((c1, cin, 1, 1, 1), generator=gen) * (1.0 / np.sqrt(c1))
    dy3 = torch.randn((1, c3) + shape, generator=gen)
    dy1 = torch.randn((1, c1) + shape, generator=gen)
    xr = torch.zeros((1, cin) + shape, dtype=torch.float64, requires_grad=True)
    if kd == 3:
        y = O.conv_nd(xr, w3.double(), None, 1), O.conv_nd(xr, w1.double(), None, 1)
    else:
        y = O.conv_nd(xr[:, :, 0], w3[:, :, 0].double(), None, 1)[:, :, None], O.conv_nd(xr[:, :, 0], w1[:, :, 0].double(), None, 1)[:, :, None]
    (y[0] * dy3.double()).sum().backward(retain_graph=True)
    (y[1] * dy1.double()).sum().backward()
    ref = xr.grad
    w3g, w1g, dy3g, dy1g = w3.to(DEV), w1.to(DEV), dy3.to(DEV), dy1.to(DEV)
    xg = torch.empty((1, cin) + shape, device=DEV)
    d3 = ops.make_desc(xg, w3g if kd == 3 else w3g[:, :, 0], 1)
    d1 = ops.make_desc(xg, w1g if kd == 3 else w1g[:, :, 0], 1)
    dx = torch.full(xg.shape, float("nan"), device=DEV)
    ops.raw_conv_bwd_data_dual(d3, dy3g, w3g, d1, dy1g, w1g, dx)
    assert rel(dx, ref) < 2e-6
    base = torch.randn(xg.shape, generator=gen)
    dxa = base.to(DEV).clone()
    ops.raw_conv_bwd_data_dual(d3, dy3g, w3g, d1, dy1g, w1g, dxa, accumulate=True)
    assert rel(dxa, ref + base.double()) < 2e-6
    dx2 = torch.full(xg.shape, float("nan"), device=DEV)
    L.set_option("dual_bwd_data", 0)
    try:
        ops.raw_conv_bwd_data_dual(d3, dy3g, w3g, d1, dy1g, w1g, dx2)
    finally:
        L.set_option("dual_bwd_data", 1)
    assert rel(dx2, ref) < 2e-6 and rel(dx, dx2) < 2e-6
    dx3 = torch.empty_like(dx)
    ops.raw_conv_bwd_data(d1, dy1g, w1g, dx3)
    ops.raw_conv_bwd_data(d3, dy3g, w3g, dx3, accumulate=True)
    assert torch.equal(dx2, dx3)


@pytest.mark.parametrize("cin,cout,shape", [(25, 16, (12, 9, 40)), (8, 13, (8, 8, 32)), (17, 26, (6, 16, 32)), (12, 20, (5, 11, 33)), (51, 32, (4, 8, 32)),
                                            (137, 8, (8, 8, 32)), (16, 40, (9, 17, 40)), (21, 16, (7, 10, 48))])
def test_conv_bwd_weight_pair_kernel_vs_oracle(ops, cin, cout, shape):
    """conv_bwd_weight_mfma_pair_kernel (two 4-channel groups per workgroup) forced onto every layer with >= 2 full groups: even / odd
    numbers of full groups, the column-trimmed one-channel tail (4m + 1 staged channels), both orientations (137 -> 8 stages dY), a
    chain on X, ragged tiles — against the fp64 oracle and against the single-group kernels."""
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    gen = torch.Generator().manual_seed(cin * 13 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    chain = torch.stack([torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen)], dim=1).contiguous()
    dy = torch.randn((1, cout) + shape, generator=gen)
    bc = lambda v: v.reshape(1, -1, 1, 1, 1)
    tx = bc(chain[:, 3]) * O.activation("LeakyReLU", bc(chain[:, 0]) * x + bc(chain[:, 1])) + bc(chain[:, 4])
    w = torch.zeros((cout, cin, 3, 3, 3))
    xg, dyg, cg = x.to(DEV), dy.to(DEV), chain.to(DEV)
    d = ops.make_desc(xg, w.to(DEV), 1)
    res = {}
    for mode in (1, 0):
        L.set_option("bw_pair", mode)
        try:
            for use_chain in (False, True):
                dw = torch.full(w.shape, float("nan"), device=DEV)
                ops.raw_conv_bwd_weight(d, xg, cg if use_chain else None, dyg, dw)
                res[(mode, use_chain)] = dw
        finally:
            L.set_option("bw_pair", 2)
    for use_chain in (False, True):
        wr = w.double().requires_grad_(True)
        O.conv_nd((tx if use_chain else x).double(), wr, None, 1).backward(dy.double())
        assert rel(res[(1, use_chain)], wr.grad) < 5e-6
        assert rel(res[(0, use_chain)], wr.grad) < 5e-6
        assert rel(res[(1, use_chain)], res[(0, use_chain)]) < 2e-6


@pytest.mark.parametrize("name", ["bn3d", "bn2d"])
def test_bn_golden(golden, ops, name):
    g = golden("ops")[name]
    st = g["state"]
    x, ga, be = G(g["x"], True), G(st["weight"], True), G(st["bias"], True)
    rm, rv = G(st["running_mean"]), G(st["running_var"])
    nbt = torch.tensor(int(st["num_batches_tracked"]), device=DEV)
    y = ops.batch_norm(x, ga, be, rm, rv, nbt)
    close(y, g["y"], 1e-5, 1e-5)
    y.backward(G(g["dy"]))
    close(x.grad, g["dx"], 1e-4, 1e-5)
    close(ga.grad, g["grads"]["weight"], 1e-5, 2e-5)
    close(be.grad, g["grads"]["bias"], 1e-5, 2e-5)
    close(rm, g["state_after"]["running_mean"], 1e-6, 1e-7)
    close(rv, g["state_after"]["running_var"], 1e-6, 1e-7)
    assert int(nbt) == int(g["state_after"]["num_batches_tracked"])


def test_bn_large_offset(ops):
    """Statistics stay accurate when |mean| >> std (double-precision two-stage reduction)."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((1, 3, 16, 32, 32), generator=gen) * 0.01 + torch.tensor([100.0, -50.0, 0.0]).reshape(1, 3, 1, 1, 1)
    ga, be = torch.ones(3), torch.zeros(3)
    yr = O.batch_norm_train(x.double(), ga.double(), be.double())
    y = ops.batch_norm(x.to(DEV), ga.to(DEV), be.to(DEV))
    # |mean|/std = 1e4: the statistics are exact (double), the fused fp32 apply a*x+b rounds at ulp(a*x) ~ 1e-3
    assert rel(y, yr.float()) < 2e-2
    assert abs(float(y.mean())) < 5e-3
    # moderate offsets (|mean|/std = 10, the regime of the network's activations) stay at fp32 round-off
    x = torch.randn((1, 3, 16, 32, 32), generator=gen) + torch.tensor([10.0, -5.0, 0.0]).reshape(1, 3, 1, 1, 1)
    yr = O.batch_norm_train(x.double(), ga.double(), be.double())
    y = ops.batch_norm(x.to(DEV), ga.to(DEV), be.to(DEV))
    assert rel(y, yr.float()) < 5e-6


@pytest.mark.parametrize("C,shape,fork,chain_b", [(25, (12, 16, 24), (12, 25), True), (16, (9, 11, 13), None, False), (6, (16, 16, 32), (4, 6), True),
                                                   (51, (8, 8, 16), (25, 51), True), (13, (5, 7, 9), None, False)])
def test_join_bwd_two_pass_vs_float64_autograd_and_the_four_kernel_sequence(ops, C, shape, fork, chain_b):
    """dpi_join_bwd (ABI 403, round 5): the BatchNorm backward of a residual join — y = BN(act(t)), t = act(BN_a(xa)) + BN_b(T_b(xb)) with, on
    the fork range, one more BatchNorm under T_b (Block3d, reference mulresunet.py:85-96), or t = act(BN_a(xa)) + act(BN_b(xb)) (ResPath3d,
    mulresunet.py:109-112) — in one reduction pass + one apply pass.  Against (1) float64 autograd of the same expressions built from
    torch primitives on the CPU (tolerance of the fp32 kernels: 2e-5 norm-wise on the tensors, 2e-4 on the per-channel gradients, which
    are sums of 1e3..1e4 cancelling terms) and (2) the rounds 1-4 sequence dpi_bn_bwd_reduce -> _apply_fork -> _apply_dual -> _apply
    (same per-element expressions; only the constants come from expanded sums: 5e-6)."""
    torch.manual_seed(7)
    slope = 0.2
    V = int(np.prod(shape))
    f64 = dict(dtype=torch.float64)
    xa = torch.randn((1, C) + shape) * 2.0 + 0.3
    xb = torch.randn((1, C) + shape) * 1.5 - 0.2
    dy = torch.randn((1, C) + shape)
    gam = {k: torch.rand(C) * 2 + 0.5 for k in ("top", "a", "b")}
    bet = {k: torch.randn(C) * 0.3 for k in ("top", "a", "b")}
    lo, hi = fork if fork else (0, 0)
    gf, bf_ = torch.rand(hi - lo) * 2 + 0.5, torch.randn(hi - lo) * 0.3

    def bn(x, g, b):            # train-mode BatchNorm over the spatial axes of a single patch, biased variance, eps 1e-5
        m = x.mean(dim=(2, 3, 4), keepdim=True)
        v = x.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
        return (x - m) / torch.sqrt(v + 1e-5) * g.view(1, -1, 1, 1, 1) + b.view(1, -1, 1, 1, 1), m.flatten(), (1.0 / torch.sqrt(v + 1e-5)).flatten()
    lrelu = lambda x: torch.where(x > 0, x, x * slope)
    # ---- float64 reference through autograd
    xa64 = xa.double().requires_grad_()
    xb64 = xb.double().requires_grad_()
    P = {k: (gam[k].double().requires_grad_(), bet[k].double().requires_grad_()) for k in gam}
    gf64, bf64 = gf.double().requires_grad_(), bf_.double().requires_grad_()
    ua, ma, ra = bn(xa64, *P["a"])
    if chain_b:                 # Block3d: side B = bn1 over c = T_CH(R); T_CH = the conv BatchNorm + activation on the fork range, identity elsewhere here
        cb = xb64.clone()
        if hi > lo:
            zf, mf, rf = bn(xb64[:, lo:hi], gf64, bf64)
            cb = torch.cat([xb64[:, :lo], lrelu(zf), xb64[:, hi:]], dim=1)
        ub, mb, rb = bn(cb, *P["b"])
        t64 = lrelu(ua) + ub
    else:
        ub, mb, rb = bn(xb64, *P["b"])
        t64 = lrelu(ua) + lrelu(ub)
    y64, mt, rt = bn(lrelu(t64), *P["top"])
    (y64 * dy.double()).sum().backward()
    # ---- HIP
    g = lambda x: x.float().contiguous().to(DEV)
    mi = lambda m, r: g(torch.cat([m.detach(), r.detach()]))
    t = g(t64.detach())
    chain = None
    if chain_b:                 # value-only input chain of side B: identity outside the fork range, act(a x + b) of the fork BatchNorm inside
        ch = torch.zeros(C, 5)
        ch[:, 0] = 1.0; ch[:, 2] = 1.0; ch[:, 3] = 1.0
        if hi > lo:
            a_f = (gf.double() * rf.detach()).float()
            ch[lo:hi, 0] = a_f
            ch[lo:hi, 1] = (bf_.double() - mf.detach() * gf.double() * rf.detach()).float()
            ch[lo:hi, 2] = slope
        chain = g(ch)
    side_a = (g(xa), mi(ma, ra), g(gam["a"]), g(bet["a"]), None, slope)
    side_b = (g(xb), mi(mb, rb), g(gam["b"]), g(bet["b"]), chain, 1.0 if chain_b else slope)
    dxf = torch.empty((1, hi - lo) + shape, device=DEV) if hi > lo else None
    fk = (lo, hi, mi(mf, rf), g(gf), g(bf_), slope, dxf) if hi > lo else None
    top = (mi(mt, rt), g(gam["top"]), g(bet["top"]))
    (dxa, dga, dea), (dxb, dgb_, deb), (dgt, det), f = ops._join_backward(g(dy), t, top[0], top[1], top[2], slope, side_a, side_b, fk)
    torch.cuda.synchronize()
    # reference gradients: dL/d(xa); dL/d(T_b(xb)) outside the fork range; dL/d(xb) on it
    assert rel(dxa, xa64.grad.numpy()) < 2e-5
    if chain_b:
        # outside the fork range T_b is the identity, so dL/d(T_b(xb)) = dL/d(xb)
        keep = [c for c in range(C) if not (lo <= c < hi)]
        if keep:
            assert rel(dxb[:, keep], xb64.grad[:, keep].numpy()) < 2e-5
        if hi > lo:
            assert rel(dxf, xb64.grad[:, lo:hi].numpy()) < 2e-5
            assert rel(f[0], gf64.grad.numpy()) < 2e-4 and rel(f[1], bf64.grad.numpy()) < 2e-4
    else:
        assert rel(dxb, xb64.grad.numpy()) < 2e-5
    for got, want in ((dgt, P["top"][0].grad), (det, P["top"][1].grad), (dga, P["a"][0].grad), (dea, P["a"][1].grad),
                      (dgb_, P["b"][0].grad), (deb, P["b"][1].grad)):
        assert rel(got, want.numpy()) < 2e-4, rel(got, want.numpy())
    # ---- t not stored: re-formed from the two sides through the chains of the forward join (dpi_chain_add_apply's operands)
    aa = (gam["a"].double() * ra.detach())
    fwd_a = torch.stack([aa, bet["a"].double() - ma.detach() * aa, torch.full((C,), slope, **f64), torch.ones(C, **f64), torch.zeros(C, **f64)], dim=1)
    ab = (gam["b"].double() * rb.detach())
    sh_b = bet["b"].double() - mb.detach() * ab
    if chain_b:       # bn1 o T_CH: the composition dpi_bn_finalize(in_chain=...) makes
        chd = ch.double()
        fwd_b = torch.stack([chd[:, 0], chd[:, 1], chd[:, 2], ab * chd[:, 3], ab * chd[:, 4] + sh_b], dim=1)
    else:
        fwd_b = torch.stack([ab, sh_b, torch.full((C,), slope, **f64), torch.ones(C, **f64), torch.zeros(C, **f64)], dim=1)
    dxf2 = torch.empty_like(dxf) if dxf is not None else None
    fk2 = fk[:6] + (dxf2,) if fk else None
    (dxa2, dga2, dea2), (dxb2, dgb2, deb2), (dgt2, det2), f2 = ops._join_backward(g(dy), None, top[0], top[1], top[2], slope, side_a, side_b, fk2,
                                                                                  fwd_chains=(g(fwd_a), g(fwd_b)))
    assert rel(dxa2, dxa.cpu().numpy()) < 2e-5 and rel(dgt2, dgt.cpu().numpy()) < 1e-4 and rel(dga2, dga.cpu().numpy()) < 1e-4
    keep = [c for c in range(C) if not (lo <= c < hi)]
    if keep:
        assert rel(dxb2[:, keep], dxb[:, keep].cpu().numpy()) < 2e-5
    if hi > lo:
        assert rel(dxf2, dxf.cpu().numpy()) < 2e-5
    # ---- the rounds 1-4 sequence on the same operands
    dt, dgB, deB, (redA, redB) = ops._bn_backward_fork(g(dy), t, top[0], top[1], top[2], slope, 1.0,
                                                      [(side_a[0], side_a[1], side_a[2], side_a[3], None, slope),
                                                       (side_b[0], side_b[1], side_b[2], side_b[3], chain, side_b[5])])
    (oa, oga, oea), (ob, ogb, oeb), redf = ops._bn_backward_apply_dual(
        dt, side_a + (redA,), side_b + (redB,), fork=(lo, hi, fk[2], fk[3], fk[4], slope) if fk else None)
    assert rel(dxa, oa.cpu().numpy()) < 5e-6
    if hi > lo:
        of, ogf, oef = ops._bn_backward_apply(ob[:, lo:hi], side_b[0][:, lo:hi], fk[2], fk[3], fk[4], 1.0, slope, redf)
        assert rel(dxf, of.cpu().numpy()) < 5e-6
        assert rel(f[0], ogf.cpu().numpy()) < 1e-4 and rel(f[1], oef.cpu().numpy()) < 1e-4      # sums of cancelling terms: both sides approximate
        keep = [c for c in range(C) if not (lo <= c < hi)]
        if keep:
            assert rel(dxb[:, keep], ob[:, keep].cpu().numpy()) < 5e-6
    else:
        assert rel(dxb, ob.cpu().numpy()) < 5e-6
    for got, want in ((dgt, dgB), (det, deB), (dga, oga), (dea, oea), (dgb_, ogb), (deb, oeb)):
        assert rel(got, want.cpu().numpy()) < 1e-4, rel(got, want.cpu().numpy())


def test_lrelu_upsample_concat_golden(golden, ops):
    g = golden("ops")["lrelu"]
    x = G(g["x"], True)
    y = ops.leaky_relu(x, 0.2)
    close(y, g["y"], 0, 0)
    y.backward(G(g["dy"]))
    close(x.grad, g["dx"], 0, 0)
    for name, mode in [("up3d_nearest", "nearest"), ("up3d_trilinear", "trilinear"), ("up3d_trilinear_1", "trilinear"),
                       ("up2d_nearest", "nearest"), ("up2d_bilinear", "bilinear")]:
        g = golden("ops")[name]
        x = G(g["x"], True)
        y = ops.upsample2x(x, mode)
        close(y, g["y"], 1e-6, 1e-6, name)
        y.backward(G(g["dy"]))
        close(x.grad, g["dx"], 1e-5, 1e-6, name)
    for name in ("concat3d_crop", "concat2d_crop"):
        g = golden("ops")[name]
        x = G(g["x"], True)
        deep = ops.upsample2x(ops.conv(x, G(g["state"]["1.0.0.weight"]), G(g["state"]["1.0.0.bias"]), 2), "nearest")
        y = ops.concat_crop([x, deep])
        close(y, g["y"], 1e-5, 1e-5, name)
        y.backward(G(g["dy"]))
        close(x.grad, g["dx"], 1e-5, 2e-5, name)


def test_upsample_crop_vs_oracle(ops):
    """Up-sampling fused with the Concat centre-crop (odd skip sizes): out = up(x)[:Do,:Ho,:Wo]."""
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((1, 3, 4, 5, 6), generator=gen)
    for mode in ("nearest", "trilinear"):
        xr = x.clone().requires_grad_(True)
        yr = O.upsample2x(xr, mode)[:, :, :7, :9, :12]
        dy = torch.randn(yr.shape, generator=gen)
        yr.backward(dy)
        xg = x.to(DEV).requires_grad_(True)
        y = ops.upsample2x(xg, mode, (7, 9, 12))
        assert rel(y, yr) < 1e-6
        y.backward(dy.to(DEV))
        assert rel(xg.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("kind", ["mae", "mse"])
def test_masked_loss_metrics(ops, kind):
    gen = torch.Generator().manual_seed(11)
    shape = (1, 1, 20, 33, 31)
    out = torch.randn(shape, generator=gen)
    img = out + 0.3 * torch.randn(shape, generator=gen) + 0.1
    mask = (torch.rand((1, 1, 1, 33, 31), generator=gen) > 0.5).float().expand(shape).contiguous()
    o_r = out.clone().requires_grad_(True)
    lr_ = O.masked_loss(o_r, img, mask, kind)
    lr_.backward()
    og = out.to(DEV).requires_grad_(True)
    loss, met = ops.masked_loss(og, img.to(DEV), mask.to(DEV), kind)
    loss.backward()
    assert abs(loss.item() - lr_.item()) < 2e-6 * abs(lr_.item())
    assert rel(og.grad, o_r.grad) < 1e-6
    m = met.cpu().numpy()
    assert abs(m[1] - O.snr(out.double(), img.double()).item()) < 1e-6
    assert abs(m[2] - O.pcorr(out.double(), img.double()).item()) < 1e-8


def test_adam_golden(golden):
    from deep_prior_interpolation_amd.optim import FusedAdam
    a = golden("host")["adam"]
    p = torch.nn.Parameter(G(a["p0"]))
    q = torch.nn.Parameter(torch.zeros(5, device=DEV))     # a second tensor in the same launch
    opt = FusedAdam([p, q], lr=1e-3)
    for k in range(5):
        p.grad = G(a["grads"][k])
        q.grad = torch.ones(5, device=DEV) * (k + 1)
        opt.step()
        np.testing.assert_allclose(p.detach().cpu().numpy(), a["traj"][k], rtol=5e-6, atol=1e-8)
    np.testing.assert_allclose(opt._m[0].cpu().numpy(), a["m"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(opt._v[0].cpu().numpy(), a["v"], rtol=1e-5, atol=1e-24)
    assert torch.all(q < 0)


def test_noise_statistics():
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    n = 1 << 22
    z = torch.zeros(n, device=DEV)
    out = torch.empty(n, device=DEV)
    step = torch.tensor([1], dtype=torch.int64, device=DEV)
    _lib.check(L.dpi_noise_add(_lib.ptr(z), n, 0.03, 1234, _lib.ptr(step), _lib.ptr(out), _lib.stream()))
    a = out.double().cpu().numpy()
    assert abs(a.mean()) < 1e-4 and abs(a.std() - 0.03) < 1e-4
    k = ((a / 0.03) ** 4).mean()
    assert abs(k - 3.0) < 0.05                      # normal kurtosis
    out2 = torch.empty(n, device=DEV)
    step += 1
    _lib.check(L.dpi_noise_add(_lib.ptr(z), n, 0.03, 1234, _lib.ptr(step), _lib.ptr(out2), _lib.stream()))
    assert abs(np.corrcoef(a, out2.double().cpu().numpy())[0, 1]) < 5e-3   # fresh draw per step
    out3 = torch.empty(n, device=DEV)
    _lib.check(L.dpi_noise_add(_lib.ptr(z), n, 0.03, 1234, _lib.ptr(step), _lib.ptr(out3), _lib.stream()))
    assert torch.equal(out2, out3)                   # deterministic given (seed, step)


def test_fill_normal_and_noise_distribution_and_stream_independence():
    """z = noise_std * N(0,1) (dpi_fill_normal, reference main.py:62-64 / utils/torch.py:61-73) and the per-iteration perturbation
    (dpi_noise_add): Kolmogorov-Smirnov against the normal CDF, tails out to 4.5 sigma, and independence between the streams the
    drivers actually use — different seeds (one per patch), z vs perturbation (stream id), consecutive patches of one Interpolator."""
    from scipy import stats
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    n = 1 << 22

    def fill(seed, stream_id, mean=0.0, std=0.1):
        out = torch.empty(n, device=DEV)
        _lib.check(L.dpi_fill_normal(_lib.ptr(out), n, mean, std, seed, stream_id, _lib.stream()))
        return out.double().cpu().numpy()

    def noise(seed, step):
        z = torch.zeros(n, device=DEV)
        out = torch.empty(n, device=DEV)
        st = torch.tensor([step], dtype=torch.int64, device=DEV)
        _lib.check(L.dpi_noise_add(_lib.ptr(z), n, 1.0, seed, _lib.ptr(st), _lib.ptr(out), _lib.stream()))
        return out.double().cpu().numpy()
    zid = (0xFFFFFFFF << 32)
    a = fill(0, zid)
    assert abs(a.mean()) < 3e-4 and abs(a.std() - 0.1) < 2e-4
    sub = a[::16] / 0.1                                                  # 262144 samples: KS critical value at 1 % is 0.0032
    assert stats.kstest(sub, "norm").statistic < 0.0032
    for k in (2.0, 3.0, 4.0, 4.5):                                       # two-sided tail mass within 4 binomial sigma
        p = 2 * stats.norm.sf(k)
        got = np.mean(np.abs(a / 0.1) > k)
        assert abs(got - p) < 4 * np.sqrt(p / n) + 1e-9, (k, got, p)
    b = fill(0, zid, mean=2.0, std=0.5)
    np.testing.assert_allclose((b - 2.0) / 0.5, a / 0.1, rtol=0, atol=2e-6)   # same draw, affine map
    pairs = {"another seed": fill(1, zid), "next patch of the same Interpolator": fill(0, zid | 1),
             "perturbation stream, same seed": noise(0, 1) * 0.1, "perturbation, next step": noise(0, 2) * 0.1,
             "perturbation, another seed": noise(1, 1) * 0.1}
    for name, v in pairs.items():
        assert abs(np.corrcoef(a, v)[0, 1]) < 3e-3, name                   # 4 M samples: |r| ~ 5e-4
        assert abs(np.corrcoef(a[:-1], v[1:])[0, 1]) < 3e-3, name          # ... and at lag 1
        assert stats.kstest(v[::16] / 0.1, "norm").statistic < 0.0032, name
    assert abs(np.corrcoef(noise(0, 1), noise(1, 1))[0, 1]) < 3e-3


def test_overlap_add_vs_oracle():
    from deep_prior_interpolation_amd import _lib
    from deep_prior_interpolation_amd.utils import window_origins
    L = _lib.load()
    rng = np.random.RandomState(0)
    shape, dim, stride = (20, 18, 22), (8, 6, 10), (4, 4, 6)
    grid = O.patch_grid(shape, dim, stride)
    pa = rng.randn(*(grid + dim)).astype(np.float32)
    ref = O.reconstruct_nd(pa.astype(np.float64), dim, stride) / 40.0
    cs = ref.shape
    acc = torch.zeros(cs, device=DEV)
    flat = pa.reshape((-1,) + dim)
    for p, org in zip(flat, window_origins(shape, dim, stride)):
        t = torch.from_numpy(p).to(DEV)
        _lib.check(L.dpi_overlap_add(_lib.ptr(t), *dim, *[int(o) for o in org], _lib.ptr(acc), *cs, _lib.stream()))
    _lib.check(L.dpi_overlap_normalize(_lib.ptr(acc), *cs, *dim, *stride, 40.0, _lib.stream()))
    np.testing.assert_allclose(acc.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


# ---- size-independent properties at BASELINE sizes (configs[2]: 64^3 patch, 64-channel input) -------------------
def test_conv_adjoint_and_linearity_at_full_size(ops):
    gen = torch.Generator(device=DEV).manual_seed(0)
    for cin, cout, shape, k, s in [(64, 4, (64, 64, 64), 3, 1), (25, 16, (64, 64, 64), 3, 1), (25, 25, (64, 64, 64), 3, 2),
                                   (67, 25, (64, 64, 64), 1, 1)]:
        x = torch.randn((1, cin) + shape, device=DEV, generator=gen)
        x2 = torch.randn((1, cin) + shape, device=DEV, generator=gen)
        w = (torch.randn((cout, cin, k, k, k), device=DEV, generator=gen) / np.sqrt(cin * k ** 3)).requires_grad_(True)
        xg = x.clone().requires_grad_(True)
        y = ops.conv(xg, w, None, s)
        dy = torch.randn(y.shape, device=DEV, generator=gen)
        y.backward(dy)
        lhs = float((y.detach().double() * dy.double()).sum())
        assert abs(lhs - float((x.double() * xg.grad.double()).sum())) < 1e-5 * abs(lhs) + 1e-3     # <Ax,y> = <x,A^T y>
        assert abs(lhs - float((w.double() * w.grad.double()).sum())) < 1e-5 * abs(lhs) + 1e-3     # bilinear in W
        y2 = ops.conv(x2, w.detach(), None, s)
        y12 = ops.conv(x + x2, w.detach(), None, s)
        assert rel(y12, y.detach() + y2) < 5e-6


def test_input_noise_fir_filters_golden(golden):
    """--filter_noise_with_wavelet / --lowpass_* path: FIR along the time axis (reference ConvolveKernel_1d, utils/processing.py:34-79)."""
    from deep_prior_interpolation_amd import utils as u
    g = golden("host")["fir"]
    for tag in ("nd2", "nd3"):
        c = g[tag]
        y = u.ConvolveKernel_1d(kernel=c["taps"], ndim=c["x"].ndim - 2)(G(c["x"]))
        np.testing.assert_allclose(y.cpu().numpy(), c["y"], rtol=1e-5, atol=1e-6)
    fc, fs, ntaps, order, nfft = g["butter"]["cfg"]
    np.testing.assert_allclose(u.butterworth_fir_taps(fc, fs, int(ntaps), int(order), int(nfft)), g["butter"]["taps"], rtol=1e-10, atol=1e-12)


def test_data_forgetting_and_filters_in_build_input():
    """build_input with --lowpass_* and --data_forgetting_factor: shapes, normalisation and the per-iteration axpy."""
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    a = parse_arguments(["--imgdir", "x", "--datadim", "3d", "--filters", "4", "8", "--skip", "4", "--inputdepth", "6", "--epochs", "4",
                         "--lowpass_fs", "250", "--lowpass_fc", "20", "--data_forgetting_factor", "3", "--gpu", "0"])
    rng = np.random.RandomState(0)
    img = rng.randn(16, 8, 8, 1)
    mask = np.broadcast_to((rng.rand(1, 8, 8, 1) > 0.4).astype(np.float64), img.shape).copy()
    T = Interpolator(a, "/tmp")
    T.load_data({"image": img, "mask": mask, "name": "0"})
    T.build_model()
    T.build_input()
    assert T.input_.shape == (1, 6, 16, 8, 8) and T.add_data_.shape == (1, 6, 16, 8, 8)
    assert abs(float(T.add_data_.std()) - float(T.input_.std())) < 1e-6 * float(T.input_.std()) + 1e-7
    np.testing.assert_allclose(T.add_data_weight, np.logspace(0, -4, 3))
    T.optimize(verbose=False)
    assert len(T.history.loss) == 4 and len(T.input_list) == 3 and T.input_list[0].shape == (6, 16, 8, 8)
    assert np.isfinite(T.history.loss).all()


@pytest.mark.parametrize("cin,cout,shape,k,stride", [(4, 64, (8, 16, 40), 3, 1), (16, 25, (8, 16, 32), 3, 1), (8, 4, (8, 8, 32), 3, 1),
                                                     (25, 64, (8, 16, 32), 1, 1), (25, 25, (16, 16, 32), 3, 2), (13, 8, (5, 9, 33), 3, 1)])
def test_conv_bwd_data_accumulate(ops, cin, cout, shape, k, stride):
    """gradient fan-in: dpi_conv_bwd_data(accumulate=1) adds to the destination (every kernel family).
    Note the roles: d describes the FORWARD conv cin -> cout; backward-data maps dy (cout) to dx (cin)."""
    gen = torch.Generator().manual_seed(cin + 31 * cout)
    x = torch.randn((1, cin) + shape, generator=gen).to(DEV)
    w = (torch.randn((cout, cin, k, k, k), generator=gen) * 0.1).to(DEV)
    d = ops.make_desc(x, w, stride)
    Do, Ho, Wo = ops.desc_out_dims(d)
    dy = torch.randn((1, cout, Do, Ho, Wo), generator=gen).to(DEV)
    base = torch.randn(x.shape, generator=gen).to(DEV)
    plain = torch.empty_like(x)
    ops.raw_conv_bwd_data(d, dy, w, plain)
    acc = base.clone()
    ops.raw_conv_bwd_data(d, dy, w, acc, accumulate=True)
    assert rel(acc, base + plain) < 1e-6


def test_channel_dropout_semantics(ops):
    """Dropout3d(p) in training mode: whole channels zeroed, survivors scaled by 1/(1-p); the gradient sees the same mask."""
    from deep_prior_interpolation_amd import nn as hnn
    torch.manual_seed(3)
    x = torch.randn(1, 64, 5, 6, 7, device=DEV, requires_grad=True)
    y = hnn.Dropout(0.25)(x)
    ratio = (y.detach() / x.detach()).amax(dim=(0, 2, 3, 4)), (y.detach() / x.detach()).amin(dim=(0, 2, 3, 4))
    r = ratio[0].cpu().numpy()
    assert np.allclose(ratio[0].cpu().numpy(), ratio[1].cpu().numpy(), atol=1e-6)          # one factor per channel
    assert np.all((np.abs(r) < 1e-6) | (np.abs(r - 1 / 0.75) < 1e-5)) and 0 < (np.abs(r) < 1e-6).sum() < 40
    y.sum().backward()
    np.testing.assert_allclose(x.grad[0, :, 0, 0, 0].cpu().numpy(), r, atol=1e-6)
    assert hnn.Dropout(0.0)(x) is x


# ---- BASELINE configs[4] mixed precision: bf16 operands, fp32 accumulate (csrc/conv_bf16_mfma.hip) ------------------------------
BF16_CASES = [(25, 16, (12, 9, 40), 3), (64, 4, (8, 12, 36), 3), (4, 8, (9, 17, 33), 3), (8, 13, (8, 8, 32), 3), (25, 1, (8, 16, 32), 3),
              (67, 4, (6, 10, 34), 3), (35, 71, (8, 8, 8), 3), (71, 142, (4, 4, 4), 3), (17, 26, (6, 6, 6), 3), (105, 64, (4, 8, 16), 3),
              (9, 20, (16, 20, 70), 3), (16, 16, (3, 5, 17), 3), (5, 3, (1, 1, 1), 3), (51, 32, (32, 32, 64), 3),
              # >= 512 tiles: the 4x4x32-tile variant the mode uses at full resolution (ragged edges in every axis)
              (25, 16, (64, 64, 64), 3), (8, 13, (62, 66, 70), 3), (4, 40, (64, 60, 64), 3)]


@pytest.mark.parametrize("cin,cout,shape,k", BF16_CASES)
def test_conv_bf16_mode_vs_oracle(ops, cin, cout, shape, k, monkeypatch):
    """precision = 1: operands rounded to bf16 (RNE) on the way into v_mfma_f32_16x16x32_bf16, fp32 accumulate.
    (1) With bf16-REPRESENTABLE inputs the rounding is the identity, so the kernel must match the fp64 oracle as tightly as the
        fp32 path (2e-6): this pins the tiling, the K = 8 channels x 4 taps packing and the fragment layouts.
    (2) With arbitrary fp32 inputs the result must equal the oracle applied to the bf16-rounded operands (2e-6) and sit within
        the bf16 rounding envelope of the exact result (5e-3 norm-wise).  Backward-data runs the same kernel (flipped weights)."""
    monkeypatch.setattr(ops, "PRECISION", 1)
    from deep_prior_interpolation_amd import _lib
    _lib.load().set_option("bf16_debug", 8)          # every 3x3x3 stride-1 conv through the bf16 kernel (by default only where it pays)
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, k, k, k), generator=gen) * (1.0 / np.sqrt(cin * k ** 3))
    b = torch.randn(cout, generator=gen)
    dy = torch.randn((1, cout) + shape, generator=gen)
    for rounded in (True, False):
        xq, wq, dyq = (t.bfloat16().float() for t in (x, w, dy))
        xi, wi, dyi = (xq, wq, dyq) if rounded else (x, w, dy)
        xr, wr = xq.double().requires_grad_(True), wq.double()
        yr = O.conv_nd(xr, wr, b.double(), 1)
        yr.backward(dyq.double())
        xg, wg = xi.to(DEV).requires_grad_(True), wi.to(DEV)
        y = ops.conv(xg, wg, b.to(DEV), 1)
        assert rel(y, yr) < 2e-6, rounded
        d = ops.make_desc(xg, wg, 1)
        dx = torch.empty_like(xg)
        ops.raw_conv_bwd_data(d, dyi.to(DEV), wg, dx)
        assert rel(dx, xr.grad) < 2e-6, rounded
        if not rounded:
            y_exact = O.conv_nd(x.double(), w.double(), b.double(), 1)
            assert rel(y, y_exact) < 5e-3
    # accumulate and the fused chain / statistics epilogue go through the same entry points
    base = torch.randn(x.shape, generator=gen).to(DEV)
    acc = base.clone()
    ops.raw_conv_bwd_data(d, dy.to(DEV), wg, acc, accumulate=True)
    plain = torch.empty_like(acc)
    ops.raw_conv_bwd_data(d, dy.to(DEV), wg, plain)
    _lib.load().set_option("bf16_debug", 0)
    assert rel(acc, base + plain) < 1e-6


def test_conv_bf16_mode_chain_and_stats(ops, monkeypatch):
    import ctypes as C
    from deep_prior_interpolation_amd import _lib
    monkeypatch.setattr(ops, "PRECISION", 1)
    L = _lib.load()
    L.set_option("bf16_debug", 8)
    gen = torch.Generator().manual_seed(5)
    cin, cout, shape = 13, 9, (7, 12, 37)
    x = torch.randn((1, cin) + shape, generator=gen)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.1).bfloat16().float()
    b = torch.randn(cout, generator=gen)
    chain = torch.stack([1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen)], dim=1).contiguous()
    v = chain[:, 0].view(1, -1, 1, 1, 1) * x + chain[:, 1].view(1, -1, 1, 1, 1)
    tx = chain[:, 3].view(1, -1, 1, 1, 1) * torch.where(v > 0, v, 0.2 * v) + chain[:, 4].view(1, -1, 1, 1, 1)
    # the kernel applies the chain in fp32 (fmaf) and then rounds to bf16
    xg, wg, bg, cg = x.to(DEV), w.to(DEV), b.to(DEV), chain.to(DEV)
    d = ops.make_desc(xg, wg, 1)
    y = torch.empty((1, cout) + shape, device=DEV)
    nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
    part = torch.empty(nblk * cout * 2, dtype=torch.float64, device=DEV)
    ops.raw_conv_fwd(d, xg, cg, wg, bg, y, part)
    txg = torch.empty_like(xg)
    ops.raw_chain_apply(xg, cg, cin, xg.numel() // cin, txg)
    yr = O.conv_nd(txg.cpu().bfloat16().double(), w.double(), b.double(), 1)
    assert rel(y, yr) < 2e-6
    assert rel(txg, tx) < 1e-6
    p = part.view(nblk, cout, 2).sum(0).cpu().numpy()
    yn = y.double().cpu().numpy()[0]
    np.testing.assert_allclose(p[:, 0], yn.reshape(cout, -1).sum(1), rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(p[:, 1], (yn.reshape(cout, -1) ** 2).sum(1), rtol=1e-9)
    L.set_option("bf16_debug", 0)


SPLIT_CASES = [(25, 16, (12, 9, 40), 3), (64, 4, (8, 12, 36), 3), (8, 13, (8, 8, 32), 3), (67, 4, (6, 10, 34), 3), (35, 71, (8, 8, 8), 3),
               (17, 26, (6, 6, 6), 3), (9, 20, (16, 20, 70), 3), (25, 16, (64, 64, 64), 3), (8, 13, (62, 66, 70), 3), (4, 40, (64, 60, 64), 3)]


@pytest.mark.parametrize("cin,cout,shape,k", SPLIT_CASES)
def test_conv_split_mode_has_fp32_accuracy(ops, cin, cout, shape, k, monkeypatch):
    """precision = 2: operands split exactly into three bf16 terms, six partial products accumulated in fp32.  ARBITRARY fp32
    inputs, the fp32 path's own tolerance against the fp64 oracle (2e-6 norm-wise), forward and backward-data; and the error
    must be of the same class as the fp32 kernel's on the same inputs (within 4x)."""
    from deep_prior_interpolation_amd import _lib
    gen = torch.Generator().manual_seed(cin * 1000 + cout + 1)
    x = torch.randn((1, cin) + shape, generator=gen) * torch.logspace(-2, 2, cin).view(1, -1, 1, 1, 1)     # 4 decades of channel scale
    w = torch.randn((cout, cin, k, k, k), generator=gen) * (1.0 / np.sqrt(cin * k ** 3))
    b = torch.randn(cout, generator=gen)
    dy = torch.randn((1, cout) + shape, generator=gen)
    xr, wr = x.double().requires_grad_(True), w.double()
    yr = O.conv_nd(xr, wr, b.double(), 1)
    yr.backward(dy.double())
    errs = {}
    for prec in (2, 0):
        monkeypatch.setattr(ops, "PRECISION", prec)
        _lib.load().set_option("bf16_debug", 8 if prec else 0)
        try:
            xg, wg = x.to(DEV), w.to(DEV)
            y = ops.conv(xg, wg, b.to(DEV), 1)
            d = ops.make_desc(xg, wg, 1)
            dx = torch.empty_like(xg)
            ops.raw_conv_bwd_data(d, dy.to(DEV), wg, dx)
        finally:
            _lib.load().set_option("bf16_debug", 0)
        errs[prec] = (rel(y, yr), rel(dx, xr.grad))
    assert errs[2][0] < 2e-6 and errs[2][1] < 2e-6, errs
    assert errs[2][0] < 4 * errs[0][0] + 1e-7 and errs[2][1] < 4 * errs[0][1] + 1e-7, errs


@pytest.mark.parametrize("cin,cout,shape", [(25, 1, (32, 32, 40)), (25, 8, (16, 32, 64)), (64, 4, (16, 48, 64)), (51, 17, (12, 16, 40)), (13, 4, (33, 31, 37)),
                                            (9, 20, (8, 16, 32))])
def test_conv_bwd_weight_with_input_chain(ops, cin, cout, shape):
    """dW when the conv input is T(x) — a raw conv output with a pending BatchNorm + LeakyReLU chain — for every kernel family,
    including the swapped MFMA orientation (few output channels), where the chain is applied to the A-operand rows."""
    gen = torch.Generator().manual_seed(cin * 77 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    dy = torch.randn((1, cout) + shape, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.1
    chain = torch.stack([1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen)], dim=1).contiguous()
    xg, cg = x.to(DEV), chain.to(DEV)
    tx = torch.empty_like(xg)
    ops.raw_chain_apply(xg, cg, cin, xg.numel() // cin, tx)
    wr = w.double().requires_grad_(True)
    O.conv_nd(tx.cpu().double(), wr, None, 1).backward(dy.double())
    d = ops.make_desc(xg, w.to(DEV), 1)
    dw = torch.empty_like(w, device=DEV)
    ops.raw_conv_bwd_weight(d, xg, cg, dy.to(DEV), dw)
    assert rel(dw, wr.grad) < 5e-6
    dw2 = torch.empty_like(dw)
    ops.raw_conv_bwd_weight(d, tx, None, dy.to(DEV), dw2)                 # same thing with the chain materialised first
    assert rel(dw, dw2) < 2e-6


# shapes with W a multiple of 4 (the kernel's 16-byte staging loads; others keep the fp32 kernels): ragged bands in h, ragged
# 32-column runs, several depth chunks per band, channel counts below / at / above the 16-row blocks, one slice, one row
BF16_BWW_CASES = [(25, 16, (12, 9, 40)), (64, 4, (8, 12, 36)), (4, 8, (9, 17, 32)), (8, 13, (8, 8, 32)), (25, 1, (8, 16, 32)),
                  (67, 4, (6, 10, 36)), (35, 71, (8, 8, 8)), (17, 26, (6, 6, 4)), (9, 20, (16, 20, 72)), (16, 16, (3, 5, 20)),
                  (5, 3, (1, 1, 4)), (51, 32, (32, 32, 64)), (25, 16, (64, 64, 64)), (8, 13, (62, 66, 68)), (137, 8, (20, 24, 64))]


@pytest.mark.parametrize("cin,cout,shape", BF16_BWW_CASES)
def test_conv_bwd_weight_bf16_mode_vs_oracle(ops, cin, cout, shape, monkeypatch):
    """precision = 1, weight gradient (csrc/conv_bf16_bww.hip): X (after its chain) and dY rounded to bf16 (RNE) while staged,
    exact products, fp32 accumulate.  (1) bf16-representable operands: the rounding is the identity and the kernel must match the
    fp64 oracle like the fp32 kernels do (5e-6) — pins the band / ring / shifted-copy layout and the tap pairing.  (2) arbitrary
    fp32 operands: equal to the oracle on the rounded operands (5e-6), and within the bf16 envelope of the exact gradient.
    (3) with a pending chain on X: chain applied in fp32, then rounded."""
    from deep_prior_interpolation_amd import _lib
    monkeypatch.setattr(ops, "PRECISION", 1)
    L = _lib.load()
    L.set_option("bf16_debug", 8)
    try:
        gen = torch.Generator().manual_seed(cin * 131 + cout)
        x = torch.randn((1, cin) + shape, generator=gen)
        dy = torch.randn((1, cout) + shape, generator=gen)
        w = torch.zeros((cout, cin, 3, 3, 3))
        xq, dyq = x.bfloat16().float(), dy.bfloat16().float()
        wr = w.double().requires_grad_(True)
        O.conv_nd(xq.double(), wr, None, 1).backward(dyq.double())
        d = ops.make_desc(x.to(DEV), w.to(DEV), 1)
        assert d.precision == 1
        for xi, dyi in ((xq, dyq), (x, dy)):
            dw = torch.full_like(w, float("nan"), device=DEV)
            ops.raw_conv_bwd_weight(d, xi.to(DEV), None, dyi.to(DEV), dw)
            assert rel(dw, wr.grad) < 5e-6
        we = w.double().requires_grad_(True)
        O.conv_nd(x.double(), we, None, 1).backward(dy.double())
        assert rel(dw, we.grad) < 8e-3
        # pending chain
        chain = torch.stack([1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                             1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen)], dim=1).contiguous()
        xg, cg = x.to(DEV), chain.to(DEV)
        tx = torch.empty_like(xg)
        ops.raw_chain_apply(xg, cg, cin, xg.numel() // cin, tx)
        wc = w.double().requires_grad_(True)
        O.conv_nd(tx.cpu().bfloat16().double(), wc, None, 1).backward(dyq.double())
        dwc = torch.empty_like(w, device=DEV)
        ops.raw_conv_bwd_weight(d, xg, cg, dyq.to(DEV), dwc)
        assert rel(dwc, wc.grad) < 5e-6
    finally:
        L.set_option("bf16_debug", 0)


@pytest.mark.parametrize("cin,cout,shape", BF16_BWW_CASES)
def test_conv_bwd_weight_split_mode_has_fp32_accuracy(ops, cin, cout, shape, monkeypatch):
    """precision = 2, weight gradient: X (after its chain) and dY each split exactly into three bf16 terms, six partial products
    in fp32.  ARBITRARY fp32 operands with 4 decades of channel scale, the fp32 kernels' own tolerance against the fp64 oracle
    (5e-6), and an error of the same class as the fp32 kernel's on the same inputs (within 4x)."""
    from deep_prior_interpolation_amd import _lib
    gen = torch.Generator().manual_seed(cin * 17 + cout)
    x = torch.randn((1, cin) + shape, generator=gen) * torch.logspace(-2, 2, cin).view(1, -1, 1, 1, 1)
    dy = torch.randn((1, cout) + shape, generator=gen) * torch.logspace(-1, 1, cout).view(1, -1, 1, 1, 1)
    w = torch.zeros((cout, cin, 3, 3, 3))
    chain = torch.stack([1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         1.0 + 0.3 * torch.randn(cin, generator=gen), 0.2 * torch.randn(cin, generator=gen)], dim=1).contiguous()
    xg, cg, dyg = x.to(DEV), chain.to(DEV), dy.to(DEV)
    tx = torch.empty_like(xg)
    ops.raw_chain_apply(xg, cg, cin, xg.numel() // cin, tx)
    wr = w.double().requires_grad_(True)
    O.conv_nd(tx.cpu().double(), wr, None, 1).backward(dy.double())
    errs = {}
    for prec in (2, 0):
        monkeypatch.setattr(ops, "PRECISION", prec)
        _lib.load().set_option("bf16_debug", 8 if prec else 0)      # every layer through the split kernel (by default only where it pays)
        try:
            d = ops.make_desc(xg, w.to(DEV), 1)
            assert d.precision == prec
            dw = torch.full_like(w, float("nan"), device=DEV)
            ops.raw_conv_bwd_weight(d, xg, cg, dyg, dw)
        finally:
            _lib.load().set_option("bf16_debug", 0)
        errs[prec] = rel(dw, wr.grad)
    assert errs[2] < 5e-6, errs
    assert errs[2] < 4 * errs[0] + 1e-7, errs


def _integration_block():
    """The ctypes binding printed in INTEGRATION.md §2, verbatim."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "INTEGRATION.md")) as fp:
        text = fp.read()
    sec = text[text.index("## 2. Calling the C ABI directly"):text.index("## 2b.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 2 and "class ConvDesc" in blocks[0] and "dpi_conv_bwd_data_dual" in blocks[1]
    return root, "\n".join(blocks)             # the ABI-300 binding, then the ABI-301 additions (they use `lib` and `ConvDesc` of the first)


def test_integration_md_binding_runs_verbatim(ops, monkeypatch):
    """Row (b) of SURVEY §8: the reference-side binding a maintainer would add, as documented, against the built library — the
    document's ConvDesc must be the header's dpi_conv_desc (round 2 shipped a document with the 8-int layout of round 1)."""
    root, code = _integration_block()
    monkeypatch.chdir(root)
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    g = torch.Generator().manual_seed(5)
    for (cin, cout, shape, k, stride) in [(5, 9, (6, 10, 12), 3, 1), (7, 3, (8, 8, 40), 3, 2), (6, 11, (4, 6, 8), 1, 1)]:
        x = torch.randn((1, cin) + shape, generator=g).to(DEV)
        w = (0.2 * torch.randn((cout, cin, k, k, k), generator=g)).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        y = ns["conv3d_forward"](x, w, b, stride)
        ref = ops.conv(x, w, b, stride)
        assert y.shape == ref.shape and torch.equal(y, ref)
        yo = O.conv_nd(x.cpu().double(), w.cpu().double(), b.cpu().double(), stride)
        assert rel(y, yo) < 5e-6
    # the ABI-301 entry points through the argtypes the document declares
    import ctypes as C
    lib, ConvDesc = ns["lib"], ns["ConvDesc"]
    cin, c3, c1, shape = 21, 8, 6, (6, 10, 20)
    w3 = (0.2 * torch.randn((c3, cin, 3, 3, 3), generator=g)).to(DEV)
    w1 = (0.2 * torch.randn((c1, cin, 1, 1, 1), generator=g)).to(DEV)
    dy3, dy1 = torch.randn((1, c3) + shape, generator=g).to(DEV), torch.randn((1, c1) + shape, generator=g).to(DEV)
    d3 = ConvDesc(C.sizeof(ConvDesc), cin, c3, *shape, 3, 3, 1, 0)
    d1 = ConvDesc(C.sizeof(ConvDesc), cin, c1, *shape, 1, 1, 1, 0)
    dx = torch.empty((1, cin) + shape, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.dpi_conv_bwd_data_dual(C.byref(d3), dy3.data_ptr(), w3.data_ptr(), C.byref(d1), dy1.data_ptr(), w1.data_ptr(), dx.data_ptr(), 0, None, 0, st) == 0
    xr = torch.zeros((1, cin) + shape, dtype=torch.float64, requires_grad=True)
    ((O.conv_nd(xr, w3.cpu().double(), None, 1) * dy3.cpu().double()).sum() + (O.conv_nd(xr, w1.cpu().double(), None, 1) * dy1.cpu().double()).sum()).backward()
    assert rel(dx, xr.grad) < 2e-6
    n = lib.dpi_conv_fwd_ws_floats(C.byref(d3))
    ws = torch.empty(max(int(n), 1), device=DEV)
    x = torch.randn((1, cin) + shape, generator=g).to(DEV)
    y = torch.empty((1, c3) + shape, device=DEV)
    assert lib.dpi_conv_fwd_ws(C.byref(d3), x.data_ptr(), None, w3.data_ptr(), None, y.data_ptr(), None, ws.data_ptr() if n else None, n, st) == 0
    assert rel(y, O.conv_nd(x.cpu().double(), w3.cpu().double(), None, 1)) < 2e-6


def test_stale_descriptor_layouts_are_rejected():
    """A binding built against another dpi_conv_desc (the 8-int layout of round 1, or the 9-int one of round 2 without the
    leading size) must get DPI_E_ARG and a message, not a convolution with `precision` read from adjacent memory."""
    import ctypes as C
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    assert L.dpi_conv_desc_size() == C.sizeof(_lib.ConvDesc) == 44

    class Old8(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("Cin", "Cout", "D", "H", "W", "k", "kd", "stride")]

    class Old9(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("Cin", "Cout", "D", "H", "W", "k", "kd", "stride", "precision")]

    class Abi301(C.Structure):                            # rounds 3: size first, no `io` yet
        _fields_ = [(n, C.c_int) for n in ("size", "Cin", "Cout", "D", "H", "W", "k", "kd", "stride", "precision")]
    x = torch.randn(1, 4, 4, 8, 8, device=DEV)
    w = torch.randn(8, 4, 3, 3, 3, device=DEV)
    y = torch.full((1, 8, 4, 8, 8), 7.0, device=DEV)
    fwd = L.dpi_conv_fwd
    saved = fwd.argtypes
    fwd.argtypes = [C.c_void_p] * 8
    try:
        for d in (Old8(4, 8, 4, 8, 8, 3, 3, 1), Old9(4, 8, 4, 8, 8, 3, 3, 1, 0), Abi301(40, 4, 8, 4, 8, 8, 3, 3, 1, 0)):
            buf = (C.c_char * 64)()                       # the short struct followed by zeros, as it would sit in a caller's frame
            C.memmove(buf, C.byref(d), C.sizeof(d))
            rc = fwd(C.addressof(buf), x.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            assert rc == -1 and b"size" in L.dpi_last_error()
        good = _lib.ConvDesc(4, 8, 4, 8, 8, 3, 3, 1, 0)
        good.size = 36
        assert fwd(C.addressof(good), x.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), None, torch.cuda.current_stream().cuda_stream) == -1
        assert L.dpi_conv_bwd_weight_ws_floats(C.byref(good)) == 0
    finally:
        fwd.argtypes = saved
    torch.cuda.synchronize()
    assert bool((y == 7.0).all())                         # nothing was launched


Q4_CASES = [  # (cin, cout, shape): ragged depth / height / width tiles, 1..8 output channels, channel counts off the chunk size
    (64, 4, (8, 16, 128)), (67, 4, (6, 10, 68)), (25, 1, (9, 17, 64)), (4, 8, (5, 8, 100)), (3, 5, (4, 9, 36)), (9, 7, (13, 7, 132)),
    (1, 2, (3, 3, 4)), (5, 3, (1, 1, 8)), (137, 8, (4, 8, 64)), (2, 6, (7, 25, 60)), (7, 1, (5, 9, 36)), (30, 1, (12, 16, 128)), (1, 1, (2, 3, 8)),
    # few input channels, many outputs (not this kernel's shapes: the dispatcher must route them elsewhere, also when it is forced)
    (3, 20, (5, 9, 36)), (4, 67, (6, 10, 68)), (1, 9, (4, 4, 8)),
]


@pytest.fixture
def q4_forced():
    from deep_prior_interpolation_amd import _lib
    L = _lib.load()
    L.set_option("q4", 2), L.set_option("q4_ck", 2)
    yield L
    L.set_option("q4", 1), L.set_option("q4_ck", 0)


@pytest.mark.parametrize("ck", [2, 4])
@pytest.mark.parametrize("cin,cout,shape", Q4_CASES)
def test_conv_q4_kernel_vs_oracle(ops, q4_forced, cin, cout, shape, ck):
    """csrc/conv_q4_mfma.hip (4x4x1 MFMA, <= 8 output channels) forced onto every shape it can run: forward with bias, and — through
    a convolution whose INPUT has <= 8 channels — backward-data, against the fp64 oracle; then bit-for-bit agreement of the default
    dispatch at these small sizes with what it was before (the kernel must not change results where it is not selected)."""
    q4_forced.set_option("q4", 2), q4_forced.set_option("q4_ck", ck)
    gen = torch.Generator().manual_seed(cin * 131 + cout)
    x = torch.randn((1, cin) + shape, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * (1.0 / np.sqrt(cin * 27))
    b = torch.randn(cout, generator=gen)
    yr = O.conv_nd(x.double(), w.double(), b.double(), 1)
    y = ops.conv(x.to(DEV), w.to(DEV), b.to(DEV), 1)
    assert rel(y, yr) < 2e-6
    # backward-data through the flipped kernel: forward conv cout -> cin has `cout` (<= 8) input channels
    x2 = torch.randn((1, cout) + shape, generator=gen)
    w2 = torch.randn((cin, cout, 3, 3, 3), generator=gen) * (1.0 / np.sqrt(cout * 27))
    xr = x2.double().requires_grad_(True)
    y2 = O.conv_nd(xr, w2.double(), None, 1)
    dy = torch.randn(y2.shape, generator=gen)
    y2.backward(dy.double())
    d = ops.make_desc(x2.to(DEV), w2.to(DEV), 1)
    dx = torch.empty(x2.shape, device=DEV)
    ops.raw_conv_bwd_data(d, dy.to(DEV), w2.to(DEV), dx)
    assert rel(dx, xr.grad) < 2e-6
    base = torch.randn(x2.shape, generator=gen).to(DEV)
    acc = base.clone()
    ops.raw_conv_bwd_data(d, dy.to(DEV), w2.to(DEV), acc, accumulate=True)
    assert rel(acc, base + dx) < 1e-6
    # backward-data of the forward convolution itself: dy has `cout` channels, dx `cin`
    xr = x.double().requires_grad_(True)
    y3 = O.conv_nd(xr, w.double(), None, 1)
    dy3 = torch.randn(y3.shape, generator=gen)
    y3.backward(dy3.double())
    d3 = ops.make_desc(x.to(DEV), w.to(DEV), 1)
    dx3 = torch.empty(x.shape, device=DEV)
    ops.raw_conv_bwd_data(d3, dy3.to(DEV), w.to(DEV), dx3)
    assert rel(dx3, xr.grad) < 2e-6
    base3 = torch.randn(x.shape, generator=gen).to(DEV)
    acc3 = base3.clone()
    ops.raw_conv_bwd_data(d3, dy3.to(DEV), w.to(DEV), acc3, accumulate=True)
    assert rel(acc3, base3 + dx3) < 1e-6


@pytest.mark.parametrize("cin,shape", [(25, (9, 17, 64)), (7, (5, 9, 36)), (30, (12, 16, 128)), (25, (6, 20, 200))])
def test_conv_q4_single_output_channel_kh_packed(ops, q4_forced, cin, shape):
    """The 25 -> 1 output layer's forward kernel (conv_q4i_mfma_kernel<..., KHP>: the three kh taps in the MFMA rows, accumulators per
    INPUT row, shifted sum in the epilogue): chain on load, bias, statistics, aligned and unaligned rows, ragged tile edges."""
    import ctypes as C
    L = q4_forced
    L.set_option("q4", 2), L.set_option("q4_ck", 4)
    gen = torch.Generator().manual_seed(5 + cin)
    x = torch.randn((1, cin) + shape, generator=gen)
    chain = torch.stack([torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen)], dim=1).contiguous()
    w = torch.randn((1, cin, 3, 3, 3), generator=gen) * 0.1
    b = torch.randn(1, generator=gen)
    bc = lambda v: v.reshape(1, -1, 1, 1, 1)
    tx = bc(chain[:, 3]) * O.activation("LeakyReLU", bc(chain[:, 0]) * x + bc(chain[:, 1])) + bc(chain[:, 4])
    wg, bg, cg = w.to(DEV), b.to(DEV), chain.to(DEV)
    for use_chain in (False, True):
        yr = O.conv_nd((tx if use_chain else x).double(), w.double(), b.double(), 1)
        for misalign in (0, 1):
            buf = torch.zeros(x.numel() + 4, device=DEV)
            xg = buf[misalign:misalign + x.numel()].view(x.shape)
            xg.copy_(x.to(DEV))
            d = ops.make_desc(xg, wg, 1)
            nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
            part = torch.zeros(nblk * 2, dtype=torch.float64, device=DEV)
            y = torch.full(yr.shape, float("nan"), device=DEV)
            ops.raw_conv_fwd(d, xg, cg if use_chain else None, wg, bg, y, part)
            assert rel(y, yr) < 2e-6
            p = part.view(nblk, 1, 2).sum(0).cpu()
            np.testing.assert_allclose(p[:, 0].numpy(), yr.sum((0, 2, 3, 4)).numpy(), rtol=1e-5, atol=2e-2)
            np.testing.assert_allclose(p[:, 1].numpy(), (yr ** 2).sum((0, 2, 3, 4)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("cin,cout,shape", [(20, 4, (9, 10, 68)), (9, 3, (8, 8, 64)), (6, 8, (5, 9, 40)), (64, 4, (4, 8, 128)), (4, 19, (5, 9, 72))])
def test_conv_q4_chain_stats_and_unaligned_rows(ops, q4_forced, cin, cout, shape):
    """The same kernel with the producer's BatchNorm + LeakyReLU chain applied on load, the {sum, sum^2} epilogue for the next
    BatchNorm, and an input whose rows are NOT 16-byte aligned (a view one float into a buffer: dword staging instead of dwordx4)."""
    import ctypes as C
    gen = torch.Generator().manual_seed(11 + cin)
    x = torch.randn((1, cin) + shape, generator=gen)
    chain = torch.stack([torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen), torch.full((cin,), 0.2),
                         torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen)], dim=1).contiguous()
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.1
    b = torch.randn(cout, generator=gen)
    bc = lambda v: v.reshape(1, -1, 1, 1, 1)
    tx = bc(chain[:, 3]) * O.activation("LeakyReLU", bc(chain[:, 0]) * x + bc(chain[:, 1])) + bc(chain[:, 4])
    yr = O.conv_nd(tx.double(), w.double(), b.double(), 1)
    L = q4_forced
    wg, bg, cg = w.to(DEV), b.to(DEV), chain.to(DEV)
    for misalign in (0, 1):
        buf = torch.zeros(x.numel() + 4, device=DEV)
        xg = buf[misalign:misalign + x.numel()].view(x.shape)
        xg.copy_(x.to(DEV))
        assert xg.data_ptr() % 16 == 4 * misalign
        d = ops.make_desc(xg, wg, 1)
        nblk = L.dpi_conv_fwd_stat_blocks(C.byref(d))
        part = torch.zeros(nblk * cout * 2, dtype=torch.float64, device=DEV)
        y = torch.empty(yr.shape, device=DEV)
        ops.raw_conv_fwd(d, xg, cg, wg, bg, y, part)
        assert rel(y, yr) < 2e-6
        p = part.view(nblk, cout, 2).sum(0).cpu()
        np.testing.assert_allclose(p[:, 0].numpy(), yr.sum((0, 2, 3, 4)).numpy(), rtol=1e-5, atol=2e-2)
        np.testing.assert_allclose(p[:, 1].numpy(), (yr ** 2).sum((0, 2, 3, 4)).numpy(), rtol=1e-5)


def test_noise_add_with_regenerated_z_is_bit_identical(monkeypatch):
    """dpi_noise_add_regen_io re-draws the fixed input z from its Philox stream instead of reading it (main.py:62-64 draws z once, main.py:148-150
    adds the perturbation every iteration): same bits as dpi_noise_add on the stored z, fp32 and bf16 output, ragged length; and the Interpolator
    takes that path (opt-in, DPI_Z_REGEN=1) exactly while input_ is the untouched fill."""
    from deep_prior_interpolation_amd._lib import check, load, ptr, stream
    L = load()
    for n in (4 * 4096, 4099):
        z = torch.empty(n, device=DEV)
        check(L.dpi_fill_normal(ptr(z), n, 0.0, 0.1, 11, (0xFFFFFFFF << 32) | 3, stream()))
        step = torch.tensor([5], dtype=torch.int64, device=DEV)
        for io, dt in ((0, torch.float32), (1, torch.bfloat16)):
            a, b = torch.empty(n, dtype=dt, device=DEV), torch.empty(n, dtype=dt, device=DEV)
            check(L.dpi_noise_add_io(ptr(z), n, 0.03, 11, ptr(step), ptr(a), io, stream()))
            check(L.dpi_noise_add_regen_io(n, 0.1, 11, (0xFFFFFFFF << 32) | 3, 0.03, 11, ptr(step), ptr(b), io, stream()))
            assert torch.equal(a.view(torch.int16 if io else torch.int32), b.view(torch.int16 if io else torch.int32))
    from deep_prior_interpolation_amd import utils as u
    from deep_prior_interpolation_amd.main import Interpolator
    from deep_prior_interpolation_amd.parameter import parse_arguments
    args = parse_arguments(["--imgdir", "x", "--datadim", "3d", "--filters", "4", "8", "--skip", "4", "--inputdepth", "6", "--epochs", "2", "--gpu", "0"])
    u.set_seed(0)
    T = Interpolator(args, "/tmp")
    rng = np.random.RandomState(0)
    T.load_data({"image": rng.randn(8, 8, 12, 1), "mask": np.ones((8, 8, 12, 1)), "name": "0"})
    T.build_input()
    assert T._z_philox is None                                  # off by default (measured slower: the pass becomes ALU-bound)
    monkeypatch.setenv("DPI_Z_REGEN", "1")
    T.build_input()
    assert T._z_philox is not None and T._z_philox[0] is T.input_
    got = T.perturbed_input()
    T._z_philox = None
    T._noise_step -= 1
    ref = T.perturbed_input()
    assert torch.equal(got, ref)
