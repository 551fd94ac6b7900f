"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    cd /tmp && python /root/repo/oracle/make_golden.py

Imports polimi-ispl/deep_prior_interpolation from /root/reference through oracle/ref_shim.py and
records inputs + outputs of the hot-path pieces listed in SURVEY.md §8(c).  The committed .npz files
are data only (inputs / expected outputs / key lists); no reference source travels.
All runs use torch CPU fp32 with a FIXED thread count (recorded in each file as `meta/threads`).
"""
import argparse
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

THREADS = 4
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npy(t):
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.array(t)


def save(name, d):
    flat = {}

    def rec(prefix, v):
        if isinstance(v, dict):
            for k, x in v.items():
                rec(prefix + "/" + str(k) if prefix else str(k), x)
        else:
            flat[prefix] = npy(v)

    rec("", d)
    flat["meta/threads"] = np.int64(THREADS)
    flat["meta/torch"] = np.array(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print("wrote %s (%.1f kB, %d arrays)" % (path, os.path.getsize(path) / 1e3, len(flat)))


def sd_np(module):
    return {k: npy(v) for k, v in module.state_dict().items()}


def grads_np(module):
    return {k: npy(p.grad) for k, p in module.named_parameters()}


def randomize(module, gen):
    """Non-trivial parameters and BN affine so that every term of the backward is exercised."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            if p.ndim > 1:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
            elif "bias" in k:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            else:
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gen))


def fwd_bwd(module, x, gen):
    x = x.clone().requires_grad_(True)
    y = module(x)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    return {"x": x, "y": y, "dy": dy, "dx": x.grad, "grads": grads_np(module), "state_after": sd_np(module)}


# ------------------------------------------------------------------------------------------
def gen_ops():
    import architectures.base as B
    g = torch.Generator().manual_seed(1234)
    out = {}
    cases = {
        "conv3d_k3s1": (lambda: B.conv3d(5, 4, 3, stride=1), (1, 5, 7, 6, 9)),
        "conv3d_k3s1_wide": (lambda: B.conv3d(3, 19, 3, stride=1), (1, 3, 4, 5, 34)),
        "conv3d_k3s2_odd": (lambda: B.conv3d(3, 3, 3, stride=2), (1, 3, 7, 9, 11)),
        "conv3d_k3s2_even": (lambda: B.conv3d(6, 6, 3, stride=2), (1, 6, 8, 6, 12)),
        "conv3d_k1": (lambda: B.conv3d(7, 5, 1), (1, 7, 4, 6, 10)),
        "conv2d_k3s1": (lambda: B.conv(5, 4, 3, stride=1), (1, 5, 11, 13)),
        "conv2d_k3s2": (lambda: B.conv(4, 4, 3, stride=2), (1, 4, 11, 14)),
        "conv2d_k1": (lambda: B.conv(6, 3, 1), (1, 6, 9, 10)),
        "bn3d": (lambda: torch.nn.BatchNorm3d(6), (1, 6, 5, 4, 7)),
        "bn2d": (lambda: torch.nn.BatchNorm2d(5), (1, 5, 9, 7)),
        "lrelu": (lambda: B.get_activation("LeakyReLU"), (1, 3, 4, 5, 6)),
        "up3d_nearest": (lambda: torch.nn.Upsample(scale_factor=2, mode="nearest"), (1, 3, 3, 4, 5)),
        "up3d_trilinear": (lambda: torch.nn.Upsample(scale_factor=2, mode="trilinear"), (1, 3, 3, 4, 5)),
        "up3d_trilinear_1": (lambda: torch.nn.Upsample(scale_factor=2, mode="trilinear"), (1, 2, 1, 2, 6)),
        "up2d_nearest": (lambda: torch.nn.Upsample(scale_factor=2, mode="nearest"), (1, 3, 4, 5)),
        "up2d_bilinear": (lambda: torch.nn.Upsample(scale_factor=2, mode="bilinear"), (1, 3, 4, 5)),
        "conv3dbn": (lambda: B.conv3dbn(4, 6, 3, 1), (1, 4, 5, 6, 8)),
        "conv2dbn": (lambda: B.conv2dbn(4, 6, 3, 1), (1, 4, 9, 8)),
    }
    for name, (mk, shape) in cases.items():
        m = mk()
        randomize(m, g)
        d = {"state": sd_np(m)}
        x = torch.randn(shape, generator=g)
        if name == "lrelu":
            m = torch.nn.LeakyReLU(0.2)  # reference uses inplace=True (base.py:102); keep x intact here
        d.update(fwd_bwd(m, x, g))
        out[name] = d
    # Concat3D centre-crop at odd sizes (base.py:342-357)
    cat = B.Concat3D(1, torch.nn.Sequential(),
                     torch.nn.Sequential(B.conv3d(3, 2, 3, stride=2), torch.nn.Upsample(scale_factor=2, mode="nearest")))
    randomize(cat, g)
    d = {"state": sd_np(cat)}
    d.update(fwd_bwd(cat, torch.randn((1, 3, 7, 5, 9), generator=g), g))
    out["concat3d_crop"] = d
    cat2 = B.Concat(1, torch.nn.Sequential(),
                    torch.nn.Sequential(B.conv(3, 2, 3, stride=2), torch.nn.Upsample(scale_factor=2, mode="nearest")))
    randomize(cat2, g)
    d = {"state": sd_np(cat2)}
    d.update(fwd_bwd(cat2, torch.randn((1, 3, 7, 9), generator=g), g))
    out["concat2d_crop"] = d
    save("ops", out)


def gen_blocks():
    import architectures.mulresunet as M
    g = torch.Generator().manual_seed(4321)
    out = {}
    cases = {
        "block3d": (lambda: M.Block3d(8, 5), (1, 5, 6, 8, 10)),
        "block3d_u16": (lambda: M.Block3d(16, 11), (1, 11, 4, 6, 8)),
        "respath3d": (lambda: M.ResPath3d(7, 4), (1, 7, 6, 5, 8)),
        "block2d": (lambda: M.Block2d(8, 5), (1, 5, 12, 10)),
        "respath2d": (lambda: M.ResPath2d(7, 4, 1), (1, 7, 9, 8)),
    }
    for name, (mk, shape) in cases.items():
        m = mk()
        randomize(m, g)
        d = {"state": sd_np(m)}
        d.update(fwd_bwd(m, torch.randn(shape, generator=g), g))
        out[name] = d
    save("blocks", out)


# ------------------------------------------------------------------------------------------
def hyperbolic_volume(shape, seed=0, nev=4):
    """Small synthetic seismic-like cube (t,x,y): hyperbolic events, Ricker wavelet (ours, not the
    absent hyperbolic3d dataset)."""
    rng = np.random.RandomState(seed)
    nt, nx, ny = shape
    t = np.arange(nt)[:, None, None]
    x = np.arange(nx)[None, :, None] / max(nx, 1)
    y = np.arange(ny)[None, None, :] / max(ny, 1)
    vol = np.zeros(shape)
    for _ in range(nev):
        t0 = rng.uniform(0.1, 0.6) * nt
        v = rng.uniform(0.6, 1.6)
        amp = rng.uniform(0.5, 1.0) * rng.choice([-1, 1])
        tt = np.sqrt(t0 ** 2 + ((x ** 2 + y ** 2) * (nt / v) ** 2))
        a = (np.pi * 0.25 * (t - tt)) ** 2
        vol += amp * (1 - 2 * a) * np.exp(-a)
    return vol / np.abs(vol).max()


def trace_mask(shape, rate, seed=0):
    rng = np.random.RandomState(seed)
    nt = shape[0]
    ntr = int(np.prod(shape[1:]))
    keep = np.ones(ntr)
    keep[rng.choice(ntr, int(ntr * rate), replace=False)] = 0
    return np.broadcast_to(keep.reshape((1,) + tuple(shape[1:])), shape).copy()


def run_reference_interpolator(argv, image, mask, epochs, tag):
    """Drive the reference Interpolator exactly as proof_of_concept_3D.ipynb does and capture
    init state, per-iteration net inputs, history, final state and out_best."""
    main = ref_shim.load_main()
    import utils as u
    args = ref_shim.parse_args(argv + ["--epochs", str(epochs)])
    args.param_noise = False            # notebooks run with param_noise=False
    u.set_seed(0)
    tmp = tempfile.mkdtemp()
    T = main.Interpolator(args, tmp)
    patch = {"image": image, "mask": mask, "name": "0"}
    with redirect_stdout(io.StringIO()):
        std = T.load_data(patch)
        T.build_model()
        T.build_input()
    init = sd_np(T.net)
    z = npy(T.input_)
    inputs = []
    h = T.net.register_forward_pre_hook(lambda mod, inp: inputs.append(npy(inp[0]).copy()))
    with redirect_stdout(io.StringIO()):
        T.optimize()
    h.remove()
    d = {
        "argv": np.array(json.dumps(argv)),
        "args": np.array(json.dumps({k: v for k, v in vars(args).items()})),
        "image": image, "mask": mask, "std": np.float64(std),
        "init_state": init, "z": z, "net_inputs": np.stack(inputs),
        "loss": np.array(T.history.loss), "snr": np.array(T.history.snr),
        "pcorr": np.array(T.history.pcorr), "lr": np.array(T.history.lr),
        "final_state": sd_np(T.net), "out_best": T.out_best, "loss_min": np.float64(T.loss_min),
        "num_params": np.int64(T.num_params),
    }
    print(tag, "loss", T.history.loss)
    return d


def gen_nets():
    K = 6
    # tiny MulResUnet3D: filters 4 8 16, skip 4 8, inputdepth 8, 16^3, trilinear, MAE
    vol = hyperbolic_volume((16, 16, 16), seed=3) * 40 / 40.0
    msk = trace_mask((16, 16, 16), 0.5, seed=4)
    img4 = (vol * 2.0)[..., None]
    msk4 = msk[..., None]
    base = ["--imgdir", "/nonexistent", "--datadim", "3d", "--filters", "4", "8", "16", "--skip", "4", "8",
            "--inputdepth", "8"]
    save("net_mulresunet3d_tiny_trilinear_mae",
         run_reference_interpolator(base + ["--upsample", "linear", "--loss", "mae"], img4, msk4, K, "3d tri mae"))
    save("net_mulresunet3d_tiny_nearest_mse",
         run_reference_interpolator(base + ["--upsample", "nearest", "--loss", "mse"], img4, msk4, K, "3d nn mse"))
    # odd-sized patch -> Concat3D crop path + ceil strides
    volo = hyperbolic_volume((12, 10, 14), seed=5)
    msko = trace_mask((12, 10, 14), 0.4, seed=6)
    save("net_mulresunet3d_tiny_odd",
         run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "3d", "--filters", "4", "8", "--skip", "4",
                                     "--inputdepth", "6", "--upsample", "linear"],
                                    (volo * 2.0)[..., None], msko[..., None], 3, "3d odd"))
    # tiny Skip3D — reachable only via args.net='skip' (not in --net choices)
    main = ref_shim.load_main()  # noqa
    import parameter
    _orig = parameter.parse_arguments

    def run_skip():
        argv = ["--imgdir", "/nonexistent", "--datadim", "3d", "--filters", "4", "8", "--skip", "2", "2",
                "--inputdepth", "6", "--upsample", "linear"]
        _pa = ref_shim.parse_args

        def patched(a):
            ar = _pa(a)
            ar.net = "skip"
            return ar
        ref_shim.parse_args = patched
        try:
            return run_reference_interpolator(argv, img4, msk4, 4, "skip3d")
        finally:
            ref_shim.parse_args = _pa
    save("net_skip3d_tiny", run_skip())
    # tiny 2-D MulResUnet on a (T,X,1) image
    v2 = hyperbolic_volume((24, 20, 1), seed=7)
    m2 = trace_mask((24, 20, 1), 0.5, seed=8)
    save("net_mulresunet2d_tiny",
         run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2d", "--filters", "4", "8", "16",
                                     "--skip", "4", "8", "--inputdepth", "8", "--upsample", "linear", "--gain", "1"],
                                    v2 * 1.0, m2, K, "2d"))
    # 2.5-D: slab of 3 slices as channels (image (H,W,C=3)), outchannel = imgchannel
    v25 = hyperbolic_volume((16, 12, 3), seed=9)
    m25 = trace_mask((16, 12, 3), 0.5, seed=10)
    save("net_mulresunet25d_tiny",
         run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2.5d", "--imgchannel", "3",
                                     "--filters", "4", "8", "--skip", "4", "--inputdepth", "8",
                                     "--upsample", "nearest"], v25 * 1.0, m25, 3, "2.5d"))


def gen_acts():
    """The non-default activations of get_activation (base.py:97-114) through the reference's own Interpolator:
    ELU everywhere; Tanh inside + Sigmoid as last activation."""
    vol = hyperbolic_volume((10, 12, 14), seed=13)
    msk = trace_mask((10, 12, 14), 0.5, seed=14)
    base = ["--imgdir", "/nonexistent", "--datadim", "3d", "--filters", "4", "8", "--skip", "4", "--inputdepth", "6",
            "--upsample", "linear", "--gain", "1"]
    save("net_mulresunet3d_tiny_elu",
         run_reference_interpolator(base + ["--activation", "ELU"], (vol * 1.0)[..., None], msk[..., None], 3, "3d elu"))
    save("net_mulresunet3d_tiny_tanh_sigmoid",
         run_reference_interpolator(base + ["--activation", "Tanh", "--last_activation", "Sigmoid"],
                                    (np.abs(vol) * 0.5)[..., None], msk[..., None], 2, "3d tanh/sigmoid"))
    # (2 iterations only: Tanh saturates behind the BatchNorm weights ~ N(10, 0.2) of init_weights, and from the third
    #  iteration on the reference's own trajectory depends on summation order at the 1e-2 level)


def gen_unet():
    """Reference UNet class (architectures/unet.py) driven directly — its own get_net cannot reach it."""
    import architectures
    g = torch.Generator().manual_seed(99)
    out = {}
    for mode, shape in (("deconv", (1, 6, 32, 32)), ("bilinear", (1, 6, 32, 48)), ("nearest", (1, 6, 34, 38))):
        m = architectures.UNet(num_input_channels=6, num_output_channels=2, filters=[2, 4, 8, 16, 32], upsample_mode=mode,
                               act_fun="LeakyReLU")
        randomize(m, g)
        d = {"state": sd_np(m), "keys": np.array(json.dumps([[k, list(v.shape)] for k, v in m.state_dict().items()]))}
        r = fwd_bwd(m, torch.randn(shape, generator=g), g)
        r.pop("state_after")                      # no buffers in this net: identical to `state`
        d.update(r)
        out[mode] = d
    # leaf ops
    mp = torch.nn.MaxPool2d(2, 2)
    out["op_maxpool"] = fwd_bwd(mp, torch.randn((1, 3, 9, 12), generator=g), g)
    dc = torch.nn.ConvTranspose2d(5, 3, 4, stride=2, padding=1)
    randomize(dc, g)
    d = {"state": sd_np(dc)}
    d.update(fwd_bwd(dc, torch.randn((1, 5, 6, 7), generator=g), g))
    out["op_deconv"] = d
    inn = torch.nn.InstanceNorm2d(4)
    out["op_instnorm"] = fwd_bwd(inn, torch.randn((1, 4, 7, 9), generator=g), g)
    save("unet", out)


def gen_skip2d():
    """Reference 2-D Skip class (architectures/skip.py:5-48) driven directly — `get_net` never returns it, so no Interpolator run
    reaches it; forward + backward of a tiny hourglass (two scales, odd width -> Concat crop) in both up-sampling modes."""
    from architectures.skip import Skip
    g = torch.Generator().manual_seed(123)
    out = {}
    for mode, shape in (("nearest", (1, 5, 16, 20)), ("bilinear", (1, 5, 18, 14))):
        m = Skip(num_input_channels=5, num_output_channels=2, num_channels_down=[4, 6], num_channels_up=[4, 6], num_channels_skip=[2, 3],
                 upsample_mode=mode, act_fun="LeakyReLU")
        randomize(m, g)
        d = {"state": sd_np(m), "keys": np.array(json.dumps([[k, list(v.shape)] for k, v in m.state_dict().items()]))}
        d.update(fwd_bwd(m, torch.randn(shape, generator=g), g))
        out[mode] = d
    save("skip2d", out)


def gen_structure():
    """state_dict key/shape tables and parameter counts of the full-size nets."""
    import architectures
    out = {}

    def table(tag, argv, outch, net=None):
        args = ref_shim.parse_args(argv)
        if net:
            args.net = net
        n = architectures.get_net(args, outch)
        sd = n.state_dict()
        out[tag] = {"keys": np.array(json.dumps([[k, list(v.shape)] for k, v in sd.items()])),
                    "num_params": np.int64(sum(p.numel() for p in n.parameters()))}
        print(tag, out[tag]["num_params"], len(sd))

    table("mulresunet3d_default", ["--imgdir", "x", "--datadim", "3d"], 1)
    table("mulresunet2d_default", ["--imgdir", "x", "--datadim", "2d"], 1)
    table("mulresunet25d_c8", ["--imgdir", "x", "--datadim", "2.5d", "--imgchannel", "8"], 8)
    table("skip3d_a12", ["--imgdir", "x", "--datadim", "3d", "--filters", "16", "32", "64", "128", "128",
                         "--skip", "4", "4", "4", "4", "4"], 1, net="skip")
    table("mulresunet3d_noskip", ["--imgdir", "x", "--datadim", "3d", "--filters", "4", "8", "16",
                                  "--skip", "0", "4"], 1)
    save("structure", out)


def gen_host():
    """Host-side pieces: patches, data.extract_patches, metrics, schedulers, Adam, utils, lines."""
    ref_shim.install()
    import utils as u
    import data as D
    out = {}
    rng = np.random.RandomState(0)
    # -- PatchExtractor on small volumes (stride==dim, overlapping, non-divisible) ------------
    pe_cases = {
        "blocks_3d": ((9, 8, 7), (4, 4, 3), (4, 4, 3)),
        "overlap_3d": ((10, 9, 8), (4, 5, 4), (2, 2, 3)),
        "overlap_2d": ((11, 7), (4, 3), (3, 2)),
        "full_3d": ((5, 4, 3), (5, 4, 3), (5, 4, 3)),
    }
    for tag, (shape, dim, stride) in pe_cases.items():
        vol = rng.randn(*shape)
        pe = u.PatchExtractor(dim=dim, stride=stride)
        pa = pe.extract(vol)
        rec = pe.reconstruct(pa)
        out["pe/" + tag] = {"vol": vol, "dim": np.array(dim), "stride": np.array(stride), "patches": pa,
                            "recon": rec, "cropped_shape": np.array(pe.in_content_cropped_shape),
                            "count": np.int64(u.count_patches(shape, dim, stride)),
                            "array_shape": np.array(u.patch_array_shape(shape, dim, stride))}
        # reconstruction of perturbed patches (not the trivial round trip)
        pa2 = pa + rng.randn(*pa.shape)
        out["pe/" + tag]["patches2"] = pa2
        out["pe/" + tag]["recon2"] = pe.reconstruct(pa2)
    # -- data.extract_patches through files (3d, 2.5d slices, 2d), NaN-decimated + binary masks --
    tmp = tempfile.mkdtemp()
    vol = rng.randn(12, 10, 8)
    msk = trace_mask((12, 10, 8), 0.5, seed=1)
    nanvol = vol.copy()
    nanvol[msk == 0] = np.nan
    np.save(os.path.join(tmp, "orig.npy"), vol)
    np.save(os.path.join(tmp, "mask.npy"), msk)
    np.save(os.path.join(tmp, "nan.npy"), nanvol)
    dcases = {
        "3d_bin": ["--datadim", "3d", "--maskname", "mask.npy", "--patch_shape", "8", "6", "8", "--patch_stride", "4", "4", "8", "--gain", "3"],
        "3d_nan": ["--datadim", "3d", "--maskname", "nan.npy", "--patch_shape", "8", "6", "8", "--patch_stride", "4", "4", "8", "--gain", "3"],
        "3d_full": ["--datadim", "3d", "--maskname", "mask.npy", "--gain", "2"],
        "25d_xy": ["--datadim", "2.5d", "--slice", "xy", "--imgchannel", "4", "--maskname", "mask.npy", "--patch_shape", "-1", "-1", "-1", "--gain", "1"],
        "25d_tx": ["--datadim", "2.5d", "--slice", "tx", "--imgchannel", "4", "--maskname", "mask.npy", "--patch_shape", "-1", "-1", "-1", "--gain", "1"],
        "25d_ty": ["--datadim", "2.5d", "--slice", "ty", "--imgchannel", "2", "--maskname", "mask.npy", "--patch_shape", "6", "-1", "-1", "--patch_stride", "3", "-1", "-1", "--gain", "1"],
    }
    out["data/vol"] = vol
    out["data/mask"] = msk
    for tag, argv in dcases.items():
        args = ref_shim.parse_args(["--imgdir", tmp, "--imgname", "orig.npy"] + argv)
        ps = D.extract_patches(args)
        out["data/" + tag] = {"argv": np.array(json.dumps(argv)),
                              "images": np.stack([p["image"] for p in ps]),
                              "masks": np.stack([p["mask"] for p in ps]),
                              "names": np.array(json.dumps([p["name"] for p in ps]))}
    # -- data.reconstruct_patches through result files (3d overlap + 2.5d xy), cwd-relative ./results
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for tag in ("3d_bin", "25d_xy", "25d_ty"):
            argv = dcases[tag]
            args = ref_shim.parse_args(["--imgdir", tmp, "--imgname", "orig.npy", "--outdir", "r_" + tag] + argv)
            ps = D.extract_patches(args)
            os.makedirs(os.path.join("results", "r_" + tag), exist_ok=True)
            outs = []
            for p in ps:
                img = p["image"]
                # what Interpolator stores: 3d -> (T,X,Y) ; 2.5d -> (H,W,C)  (main.py:175-176)
                o = img[..., 0] if args.datadim == "3d" else img
                o = o + 0.1 * rng.randn(*o.shape)
                outs.append(o)
                np.save(os.path.join("results", "r_" + tag, p["name"] + "_run.npy"),
                        {"output": o, "elapsed": "0h:0m:1s", "history": None, "device": "CPU"})
            # glob order is directory order (data.py:99); single-digit zero-padded names sort the same
            import glob as _g
            order = [os.path.basename(q).split("_")[0] for q in _g.glob(os.path.join("results", "r_" + tag) + "/*.npy")]
            rec = D.reconstruct_patches(args)
            out["data/" + tag]["outputs"] = np.stack(outs)
            out["data/" + tag]["glob_order"] = np.array(json.dumps(order))
            out["data/" + tag]["recon"] = rec
    finally:
        os.chdir(cwd)
    # -- bool2bin, masks ------------------------------------------------------------------------
    out["bool2bin/in"] = nanvol
    out["bool2bin/out"] = u.bool2bin(nanvol)
    np.random.seed(0)
    out["build_mask/rand3d"] = u.build_mask(np.ones((6, 5, 4)), 0.5, regular=False)
    np.random.seed(0)
    out["build_mask/rand2d"] = u.build_mask(np.ones((6, 10)), 0.3, regular=False)
    out["build_mask/reg_hi"] = u.build_mask(np.ones((4, 12)), 0.66, regular=True)
    out["build_mask/reg_lo"] = u.build_mask(np.ones((4, 12)), 0.25, regular=True)
    np.random.seed(1)
    out["add_rand_mask/out3d"] = u.add_rand_mask(msk, 0.3)
    # -- metrics ---------------------------------------------------------------------------------
    a = torch.randn(3, 4, 5, generator=torch.Generator().manual_seed(5))
    b = a + 0.3 * torch.randn(3, 4, 5, generator=torch.Generator().manual_seed(6))
    out["metrics"] = {"out": a, "tgt": b, "snr": u.snr(a, b), "pcorr": u.pcorr(a, b),
                      "snr_np": np.float64(u.snr(npy(a).astype(np.float64), npy(b).astype(np.float64))),
                      "pcorr_np": np.float64(u.pcorr(npy(a).astype(np.float64), npy(b).astype(np.float64)))}
    H = u.History(3000)
    H.append((0.0123, 4.5, 0.87))
    H.lr.append(1e-3)
    out["history/msg"] = np.array(H.log_message(0))
    out["history/zfill"] = np.int64(H.zfill)
    # -- generic ---------------------------------------------------------------------------------
    out["generic"] = {"ten_digit": np.array([u.ten_digit(n) for n in (1, 9, 10, 343, 2001, 99999)]),
                      "sec2time": np.array(json.dumps([u.sec2time(s) for s in (0, 59.9, 61, 3600, 6739.4)])),
                      "time2sec": np.array([u.time2sec(s) for s in ("0h:0m:59s", "1h:52m:19s")]),
                      "nextpow2": np.array([u.nextpow2(n) for n in (1, 2, 3, 64, 65, 1000)])}
    # -- EarlyStopping + ReduceLROnPlateau traces on a synthetic loss curve ------------------------
    losses = np.concatenate([np.linspace(1, 0.5, 30), 0.5 + 0.001 * rng.rand(60), np.linspace(0.5, 0.45, 10),
                             0.45 + 0.0001 * rng.rand(80)])
    st = u.EarlyStopping(patience=40, min_delta=1.0, percentage=True)
    out["earlystop/losses"] = losses
    out["earlystop/stop"] = np.array([bool(st.step(torch.tensor(l))) for l in losses])
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.9, threshold=1e-5, patience=10)
    lrs = []
    for l in losses:
        sch.step(torch.tensor(l))
        lrs.append(opt.param_groups[0]["lr"])
    out["plateau/lr"] = np.array(lrs)
    out["plateau/cfg"] = np.array([1e-3, 0.9, 1e-5, 10])
    # -- Adam (torch.optim.Adam defaults, main.py:200) ----------------------------------------------
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(37, generator=g) * 1e-3
    grads = torch.randn(5, 37, generator=g) * torch.logspace(-9, 0, 37)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-3)
    traj = []
    for k in range(5):
        p.grad = grads[k].clone()
        opt.step()
        traj.append(npy(p).copy())
    s = opt.state[p]
    out["adam"] = {"p0": p0, "grads": grads, "traj": np.stack(traj), "m": s["exp_avg"], "v": s["exp_avg_sq"]}
    # -- datasets/lines known answers (proof_of_concept_2D.ipynb:308) -------------------------------
    orig = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/original.npy"))
    m66 = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/random66.npy"))
    t = torch.from_numpy((orig * m66).astype(np.float32))
    out["lines"] = {"original": orig.astype(np.float32), "mask": m66.astype(np.uint8),
                    "std": np.float64(torch.std(t).item()), "kept_traces": np.int64(m66[0, :, 0].sum())}
    # -- input-noise FIR filters (utils/processing.py:34-79; 'next' row f.1) -------------------------
    gz = torch.Generator().manual_seed(21)
    taps = np.array([0.1, -0.2, 0.5, 0.9, 0.4, -0.1, 0.05])
    for nd, shape in ((2, (1, 3, 16, 5)), (3, (1, 3, 16, 4, 5))):
        x = torch.randn(shape, generator=gz)
        W = u.ConvolveKernel_1d(kernel=taps, ndim=nd, dtype=torch.FloatTensor)
        out["fir/nd%d" % nd] = {"x": x, "taps": taps, "y": W(x)}
    L = u.LowPassButterworth(fc=20.0, ndim=3, fs=250.0, ntaps=7, order=4, nfft=2 ** u.nextpow2(16), dtype=torch.FloatTensor)
    out["fir/butter"] = {"taps": L.taps, "cfg": np.array([20.0, 250.0, 7, 4, 2 ** u.nextpow2(16)])}
    save("host", out)


def gen_lines():
    """BASELINE configs[3] data: the shipped 2-D section datasets/lines (170,100,1) with its random66 mask, driven through the
    reference Interpolator as --datadim 2d, and tiled to a (170,100,8) volume (the section shifted one trace per slice) cut into
    2.5-D slabs of 4 slices (--datadim 2.5d --imgchannel 4 --slice tx: the slice axis becomes the channel axis)."""
    orig = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/original.npy")).astype(np.float64)
    m66 = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/random66.npy")).astype(np.float64)
    tiny = ["--filters", "4", "8", "16", "--skip", "4", "8", "--inputdepth", "8", "--upsample", "linear", "--gain", "1"]
    save("net_lines2d_tiny", run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2d"] + tiny, orig, m66, 2, "lines 2d"))
    slab = np.stack([np.roll(orig[..., 0], k, axis=1) for k in range(4)], axis=-1)
    mslab = np.stack([np.roll(m66[..., 0], 3 * k, axis=1) for k in range(4)], axis=-1)
    save("net_lines25d_tiny", run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2.5d", "--imgchannel", "4", "--slice", "tx"] + tiny,
                                                         slab, mslab, 2, "lines 2.5d"))


def gen_lines_skip():
    """BASELINE configs[3] as written: the 2.5-D SKIP net on the datasets/lines slabs.  The reference's get_net sends every 2d / 2.5d
    `--net` that is not unet / attmultiunet / part to MulResUnet (architectures/__init__.py:41-53) and its parser has no `skip` choice,
    so the 2-D `Skip` class (architectures/skip.py:5-48) is unreachable upstream.  Here the reference's OWN class runs inside the
    reference's OWN Interpolator: `architectures.get_net` is wrapped (before main.py binds it, main.py:10) to return
    Skip(...) with the argument mapping get_net uses for Skip3D (architectures/__init__.py:62-72) when args.net == 'skip'."""
    import architectures
    from architectures.skip import Skip
    orig_get_net = architectures.get_net

    def get_net(args, outchannel=1):
        if args.datadim in ("2d", "2.5d") and args.net == "skip":
            return Skip(num_input_channels=args.inputdepth, num_output_channels=outchannel, num_channels_down=args.filters,
                        num_channels_up=args.filters, num_channels_skip=args.skip, upsample_mode=args.upsample, need_bias=True,
                        act_fun=args.activation, last_act_fun=args.last_activation, dropout=args.dropout)
        return orig_get_net(args, outchannel)
    _pa = ref_shim.parse_args

    def parse_skip(a):
        ar = _pa(a)
        ar.net = "skip"
        return ar
    orig = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/original.npy")).astype(np.float64)
    m66 = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/random66.npy")).astype(np.float64)
    slab = np.stack([np.roll(orig[..., 0], k, axis=1) for k in range(4)], axis=-1)
    mslab = np.stack([np.roll(m66[..., 0], 3 * k, axis=1) for k in range(4)], axis=-1)
    tiny = ["--filters", "4", "8", "16", "--skip", "2", "2", "2", "--inputdepth", "8", "--upsample", "linear", "--gain", "1"]
    architectures.get_net, ref_shim.parse_args = get_net, parse_skip
    try:
        save("net_lines25d_skip_tiny", run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2.5d", "--imgchannel", "4", "--slice", "tx"] + tiny,
                                                                  slab, mslab, 3, "lines 2.5d skip"))
        save("net_lines2d_skip_tiny", run_reference_interpolator(["--imgdir", "/nonexistent", "--datadim", "2d"] + tiny, orig, m66, 3, "lines 2d skip"))
    finally:
        architectures.get_net, ref_shim.parse_args = orig_get_net, _pa


def gen_checkpoint():
    """A checkpoint PRODUCED BY THE REFERENCE (main.py:238-240 torch.save(state_dict) + utils/generic.py:46 write_args) for the
    transfer-learning flow --netdir (main.py:101-110): tests/golden/ckpt_ref/{args.txt, 0_model.pth} + an input/output pair."""
    main = ref_shim.load_main()
    import utils as u
    argv = ["--imgdir", "/nonexistent", "--datadim", "3d", "--filters", "4", "8", "16", "--skip", "4", "8", "--inputdepth", "8",
            "--upsample", "linear", "--epochs", "2", "--savemodel", "--outdir", "ckpt_ref"]
    args = ref_shim.parse_args(argv)
    args.param_noise = False
    vol = hyperbolic_volume((16, 16, 16), seed=3)
    msk = trace_mask((16, 16, 16), 0.5, seed=4)
    u.set_seed(0)
    d = os.path.join(OUT, "ckpt_ref")
    os.makedirs(d, exist_ok=True)
    u.write_args(os.path.join(d, "args.txt"), args)
    T = main.Interpolator(args, d)
    with redirect_stdout(io.StringIO()):
        T.load_data({"image": (vol * 2.0)[..., None], "mask": msk[..., None], "name": "0"})
        T.build_model()
        T.build_input()
        T.optimize()
        T.save_result()                      # writes 0_run.npy and 0_model.pth
    os.remove(os.path.join(d, "0_run.npy"))  # only the checkpoint travels
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 8, 16, 16, 16), generator=g)
    sd = {k: v.clone() for k, v in T.net.state_dict().items()}
    y = T.net(x)
    save("ckpt_ref_io", {"x": x, "y": y, "state": {k: npy(v) for k, v in sd.items()}})


def gen_operators():
    """Anti-aliasing add-on operators and the POCS regulariser: reference operators/derivative.py,
    operators/signal.py, operators/base.py, utils/slopes.py, utils/processing.py:88-181, utils/pocs.py."""
    import operators as OP
    import utils as u
    from utils import slopes as SL, processing as PR, pocs as PC
    g = torch.Generator().manual_seed(77)
    out = {}
    x = torch.randn((1, 2, 9, 7), generator=g)
    yv = torch.randn((1, 2, 9, 7), generator=g)
    V = OP.VerticalGrad()
    out["vgrad"] = {"x": x, "y": V.forward(x), "r": yv, "adj": V.adjoint(yv)}
    # Chain / Hessian over VerticalGrad
    Ch = OP.Chain([V, V])
    out["chain"] = {"x": x, "y": Ch.forward(x), "adj": Ch.adjoint(yv), "hess": OP.Hessian(V).forward(x)}
    # derivatives: every stencil on every axis of a 4-D tensor, non-unit spacing
    d = {}
    xd = torch.randn((2, 3, 6, 5), generator=g)
    for ax in range(4):
        for st in ("forward", "backward", "centered"):
            d["first/ax%d/%s" % (ax, st)] = PR.first_derivative(xd, spacing=0.7, axis=ax, stencil=st)
        d["second/ax%d" % ax] = PR.second_derivative(xd, spacing=0.7, axis=ax)
    d["x"] = xd
    out["deriv"] = d
    # Hale2D / directional_laplacian
    xh = torch.randn((1, 1, 12, 10), generator=g)
    th = (torch.rand((1, 1, 12, 10), generator=g) - 0.5) * 3.0
    H = SL.Hale2D(th)
    out["hale"] = {"x": xh, "theta": th, "y": H(xh), "dl": SL.directional_laplacian(xh, th)}
    xh2 = torch.randn((1, 3, 8, 9), generator=g)
    th2 = (torch.rand((1, 3, 8, 9), generator=g) - 0.5) * 3.0
    out["hale_c3"] = {"x": xh2, "theta": th2, "y": SL.Hale2D(th2)(xh2)}
    # structure tensor dips: plain, with smoothing (channels = 1: the reference's GaussianFilter weight is (1,1,K,K)), and on the
    # shipped 2-D section datasets/lines/original.npy (BASELINE configs[3] data)
    xs = torch.randn((1, 1, 14, 10), generator=g)
    p0, a0 = SL.structure_tensor_dips(xs, dv=1.0, dh=1.0, smooth=0.0)
    p1, a1 = SL.structure_tensor_dips(xs, dv=0.5, dh=2.0, smooth=1.5)
    out["dips"] = {"x": xs, "phi0": p0, "aniso0": a0, "phi1": p1, "aniso1": a1}
    lines = np.load(os.path.join(ref_shim.REFERENCE_ROOT, "datasets/lines/original.npy"))[..., 0].astype(np.float32)
    xl = torch.from_numpy(lines)[None, None]
    pl, al = SL.structure_tensor_dips(xl, smooth=2.0)
    Hl = SL.Hale2D(pl)
    out["lines"] = {"phi": pl, "aniso": al, "hale_of_data": Hl(xl)}
    # Gaussian filter 1-D / 2-D
    xg1 = torch.randn((1, 1, 17), generator=g)
    xg2 = torch.randn((1, 1, 11, 13), generator=g)
    out["gauss"] = {"x1": xg1, "y1": PR.GaussianFilter(1, 7, 1, 1.3)(xg1).detach(), "x2": xg2, "y2": PR.GaussianFilter(1, 9, 2, 2.0)(xg2).detach(),
                    "kernel": PR._gaussian_kernel(9, 2.0)}
    # VerticalConv
    wav = PR.ricker_wavelet(9, 2.0).numpy().astype(np.float64) + 0.05 * np.arange(9)      # asymmetric on purpose
    VC = OP.VerticalConv(wav)
    xc = torch.randn((1, 3, 15, 6), generator=g)
    out["vconv"] = {"wavelet": wav, "x": xc, "y": VC.forward(xc), "adj": VC.adjoint(xc)}
    # POCS: threshold / compute_threshold / POCS.forward with the removed torch.rfft / irfft (onesided=False) emulated by their
    # documented semantics; the thresholding and weighting are the reference's code
    def rfft_full(t, nd):
        return torch.view_as_real(torch.fft.fftn(t, dim=tuple(range(-nd, 0))))

    def irfft_full(T, nd):
        return torch.fft.ifftn(torch.view_as_complex(T.contiguous()), dim=tuple(range(-nd, 0))).real
    for tag, shape in (("2d", (1, 1, 16, 12)), ("3d", (1, 1, 8, 6, 10))):
        nd = len(shape) - 2
        data = torch.randn(shape, generator=g)
        mask = (torch.rand(shape, generator=g) > 0.5).float()
        xo = torch.randn(shape, generator=g)
        P = PC.POCS(data=data * mask, mask=mask, weight=0.1, thresh_perc=5.0,
                    forward_fn=lambda t, nd=nd: rfft_full(t, nd), adjoint_fn=lambda T, nd=nd: irfft_full(T, nd))
        spec = rfft_full(xo, nd)
        out["pocs_" + tag] = {"data": data * mask, "mask": mask, "x": xo, "y": P(xo), "spec": spec,
                              "thresh": np.float64(PC.compute_threshold(spec, 5.0)), "thresholded": PC.threshold(spec, PC.compute_threshold(spec, 5.0))}
    save("operators", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    torch.set_num_threads(THREADS)
    os.makedirs(OUT, exist_ok=True)
    ref_shim.install()
    todo = {"ops": gen_ops, "blocks": gen_blocks, "nets": gen_nets, "structure": gen_structure, "host": gen_host,
            "unet": gen_unet, "skip2d": gen_skip2d, "acts": gen_acts, "operators": gen_operators, "lines": gen_lines, "lines_skip": gen_lines_skip, "checkpoint": gen_checkpoint}
    for k, fn in todo.items():
        if a.only is None or k in a.only:
            fn()
