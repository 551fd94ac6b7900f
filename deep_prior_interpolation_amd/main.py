"""Per-patch deep-prior optimiser — drop-in for reference main.py (Interpolator life-cycle
load_data -> build_model -> build_input -> optimize -> save_result -> clean, and the `main()` CLI).

Everything inside the iteration runs on the GPU through libdpi_hip.so: input perturbation (Philox),
network forward/backward (HIP convs / BN / up-sampling), fused masked loss + SNR/PCORR, fused Adam.
"""
import os
import warnings
from time import time

import numpy as np
import torch

from . import _lib, ops
from . import utils as u
from .architectures import get_net
from .data import extract_patches
from .optim import DevicePlateau, FusedAdam
from .parameter import net_args_are_same, parse_arguments

warnings.filterwarnings("ignore")


class Interpolator:
    def __init__(self, args, outpath, device=None, seed=0):
        self.args = args
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.DpiError("no HIP device: deep_prior_interpolation_amd has no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.outpath = outpath
        self.loss_kind = "mse" if args.loss == "mse" else "mae"
        self.elapsed = None
        self.iiter = 0
        self.iter_to_be_saved = list(range(0, args.epochs, int(args.save_every))) if args.save_every is not None else [0]
        self.loss_min = None
        self.outchannel = args.imgchannel
        self.history = u.History(args.epochs)
        self.image_name = None
        self.img = self.img_ = self.mask = self.mask_ = None
        self.out_best = None
        self._out_best_dev = None
        self.zfill = u.ten_digit(args.epochs)
        self.input_ = None
        self.add_data_ = None
        self.add_data_weight = None
        self.input_list = []
        self.net = None
        self.num_params = None
        self.optimizer = None
        self.noise_seed = int(seed)
        self._noise_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._patches_seen = 0          # mixed into the Philox stream id of z: every patch gets its own z (main.py:62-64 draws a fresh one)

    # ------------------------------------------------------------------------------------------
    def begin_patch(self, index, base_seed=0):
        """Seed everything patch `index` draws — initial weights (torch's CPU generator, consumed by init_weights), z and the
        per-iteration perturbation (Philox key) — from the patch index alone, so its result does not depend on which process,
        rank or concurrency slot optimises it.  Patch 0 with base_seed 0 reproduces `u.set_seed(0)` + a fresh Interpolator
        (bit-identical initial weights to the reference's first patch, tests/test_host.py)."""
        seed = int(base_seed) + int(index)
        torch.manual_seed(seed)
        self.noise_seed = seed
        self._patches_seen = 0
        self._noise_step.zero_()

    def load_data(self, data):
        """(T,X,Y,C) numpy patch -> (1,C,T,X,Y) fp32 device tensors; returns std of the masked data (main.py:118-139)."""
        self.image_name = data["name"]
        self.img = data["image"]
        self.mask = data["mask"]
        if self.mask.shape != self.img.shape:
            raise ValueError("The loaded mask shape has to be", self.img.shape)
        perm = (self.img.ndim - 1,) + tuple(range(self.img.ndim - 1))
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.transpose(a, perm))).unsqueeze(0).float().to(self.device)
        self.img_ = to_dev(self.img)
        self.mask_ = to_dev(self.mask)
        return torch.std(self.img_ * self.mask_).item()

    def release_packed_weights(self):
        """Hand the packed-weight scratch slots of this Interpolator's network back to the library (dpi_pack_forget, ABI 402): the bf16
        arithmetic modes keep one slot per (weight tensor, shape) and every patch builds a new network (reference main.py:286), so a
        343-patch job would otherwise hold 343 networks' worth of slots.  Called when the network is replaced; the patch that used it has
        been synchronised by then (optimize() / graph_finish() read the best output back)."""
        if self.net is None or getattr(self.args, "precision", "fp32") == "fp32":
            return 0
        L = _lib.load()
        n = 0
        for p in self.net.parameters():
            if p.ndim >= 4 and p.is_cuda:
                n += L.dpi_pack_forget(p.data_ptr())
        self._graph = None            # a captured iteration of the old network holds the slot addresses: it must not be replayed again
        return n

    def build_model(self, netpath=None):
        self.release_packed_weights()
        if self.outchannel is None:
            self.outchannel = self.img_.shape[1]
        if len(self.args.netdir) != 0:
            saved = u.read_args(os.path.join("./results", *netpath.split("/")[:-1], "args.txt"))
            assert net_args_are_same(self.args, saved)
            self.net = get_net(saved, self.outchannel)
            self.net.load_state_dict(torch.load(os.path.join("./results", netpath), map_location="cpu"))
        else:
            self.net = get_net(self.args, self.outchannel)
            u.init_weights(self.net, self.args.inittype, self.args.initgain)
        self.net = self.net.float().to(self.device)
        self._storage_ok = None
        self.num_params = sum(int(np.prod(list(p.size()))) for p in self.net.parameters())

    def build_input(self):
        """z = noise_std * N(0,1) of shape (1, inputdepth, *patch), optionally FIR-filtered along t (wavelet / Butterworth)
        and prepared for the data-forgetting term (main.py:59-97)."""
        a = self.args
        philox_z = (None,)
        self._z_cpu = None
        if self.noise_source() == "torch_cpu":
            # parity mode: the reference's own draw (main.py:61-64: get_noise(...).type(dtype); input_ *= noise_std) from torch's CPU generator
            z = u.get_noise((1, a.inputdepth) + self.img.shape[:-1], a.noise_dist).float()
            z *= a.noise_std
            self._z_cpu = z
            z = z.to(self.device)
        elif a.noise_dist != "n":
            z = u.get_noise((1, a.inputdepth) + self.img.shape[:-1], a.noise_dist).to(self.device) * a.noise_std
        else:
            z = torch.empty((1, a.inputdepth) + self.img.shape[:-1], dtype=torch.float32, device=self.device)
            # stream id: high word marks "z" (the per-iteration perturbation uses the iteration counter as stream id), low word =
            # running patch count of this Interpolator, so consecutive patches do not share z
            zstream = (0xFFFFFFFF << 32) | (self._patches_seen & 0xFFFFFFFF)
            _lib.check(_lib.load().dpi_fill_normal(_lib.ptr(z), z.numel(), 0.0, float(a.noise_std), self.noise_seed, zstream, _lib.stream()),
                       "dpi_fill_normal")
            philox_z = (z, float(a.noise_std), int(self.noise_seed), int(zstream))      # dropped below if anything rewrites z
        self._patches_seen += 1
        self._z_philox = None
        if a.filter_noise_with_wavelet:                         # main.py:66-72
            z = u.ConvolveKernel_1d(kernel=np.load(os.path.join(a.imgdir, "wavelet.npy")), ndim=z.ndim - 2)(z)
        if a.lowpass_fs and a.lowpass_fc:                       # main.py:74-84: 4th-order Butterworth as an FIR along t
            z = u.LowPassButterworth(fc=a.lowpass_fc, ndim=z.ndim - 2, fs=a.lowpass_fs, ntaps=a.lowpass_ntaps, order=4,
                                     nfft=2 ** u.nextpow2(z.shape[2]))(z)
        if a.data_forgetting_factor != 0:                       # main.py:86-97
            data_ = self.img_ * self.mask_
            rep = int(np.ceil(z.shape[1] / data_.shape[1]))
            data_ = data_.repeat([1, rep] + [1] * (z.ndim - 2))[:, :a.inputdepth].contiguous()
            data_ = data_ * (torch.std(z) / torch.std(data_))
            self.add_data_ = data_
            self.add_data_weight = np.logspace(0, -4, a.data_forgetting_factor)
        self.input_ = z
        if self._z_cpu is not None and (a.filter_noise_with_wavelet or (a.lowpass_fs and a.lowpass_fc)):
            self._z_cpu = z.float().cpu()                       # the perturbation below is added to the FILTERED z (main.py:148-150)
        # z is still the plain Philox fill: the per-iteration perturbation COULD re-draw it instead of reading it (dpi_noise_add_regen_io,
        # DPI_Z_REGEN=1).  Off by default — measured slower: a second Philox4x32-10 + Box-Muller per four elements makes the pass ALU-bound
        # (0.45 ms at 256x128x128 x 64 channels against 0.36-0.41 ms for reading z: 134 M normal deviates per iteration either way)
        if a.noise_dist == "n" and z is philox_z[0] and os.environ.get("DPI_Z_REGEN") == "1":
            self._z_philox = philox_z

    # ------------------------------------------------------------------------------------------
    def perturbed_input(self):
        """input = z + reg_noise_std * N(0,1), fresh every iteration (main.py:148-150)."""
        if self.noise_source() == "torch_cpu":
            return self._perturbed_input_torch_cpu()
        if self.args.reg_noise_std <= 0:
            return self.input_
        # (bf16 storage: the perturbed input is the first activation the net reads — 64 channels at full resolution — and is written as
        #  bf16 straight away; z itself stays fp32)
        bf = ops.storage_bf16() and self.input_.ndim == 5
        out = torch.empty_like(self.input_, dtype=torch.bfloat16) if bf else torch.empty_like(self.input_)
        self._noise_step += 1
        zp = getattr(self, "_z_philox", None)
        if zp is not None and zp[0] is self.input_:
            _lib.check(_lib.load().dpi_noise_add_regen_io(out.numel(), zp[1], zp[2], zp[3], float(self.args.reg_noise_std), self.noise_seed,
                                                          _lib.ptr(self._noise_step), _lib.ptr(out), _lib.STORE_FWD_BF16 if bf else 0,
                                                          _lib.stream()), "dpi_noise_add_regen")
            return out
        _lib.check(_lib.load().dpi_noise_add_io(_lib.ptr(self.input_), out.numel(), float(self.args.reg_noise_std),
                                                self.noise_seed, _lib.ptr(self._noise_step), _lib.ptr(out),
                                                _lib.STORE_FWD_BF16 if bf else 0, _lib.stream()), "dpi_noise_add")
        return out

    def noise_source(self):
        return getattr(self.args, "noise_source", "philox")

    def _perturbed_input_torch_cpu(self):
        """--noise_source torch_cpu (parity mode): the reference's draws of one iteration in the reference's order, from torch's CPU generator —
        main.py:143-145: with --param_noise (ON by default in the CLI) one normal_() per 4-D / 5-D parameter, whose result the
        reference discards (it rebinds a local) but which moves the generator; main.py:148-150: input = z + reg_noise_std * z.clone().normal_()
        in fp32 on the host — then uploaded.  ~3 s of host time per iteration at 256x128x128 x 64 channels: a parity tool, not the fast path."""
        a = self.args
        if a.param_noise:
            for p in self.net.parameters():
                if p.ndim in (4, 5):
                    torch.empty(p.shape, dtype=torch.float32).normal_()
        inp = self._z_cpu.clone()
        if a.reg_noise_std > 0:
            inp += a.reg_noise_std * inp.clone().normal_()
        inp = inp.to(self.device)
        return inp.to(torch.bfloat16) if (ops.storage_bf16() and inp.ndim == 5) else inp

    def _to_numpy_out(self, out_):
        """(1,1,T,X,Y) -> (T,X,Y) ; (1,C,H,W) -> (H,W,C)  (main.py:175-176)."""
        return u.torch_to_np(out_, True) if out_.ndim > 4 else u.torch_to_np(out_, False)[0].transpose((1, 2, 0))

    def precision_scope(self):
        """--precision of THIS Interpolator as an ops.mode_scope: the arithmetic mode of the convolutions and the storage type of the activations of
        everything built inside it, on this host thread only (round 6; rounds 2-5 flipped module globals of `ops` at the top of every iteration).
        Every iteration, eager or captured, runs inside one — so the mode never depends on which Interpolator ran before in the process, and two
        Interpolators of different --precision cannot disturb each other whichever way they are interleaved.  DPI_PRECISION / DPI_STORAGE in the
        environment are tools-only overrides of the fp32 default: an fp32 Interpolator leaves that half to the process default.
        bf16 STORAGE of the activations (BASELINE configs[4]) where every node of the net is a fused 3-D node that takes it (storage_bf16_ok)."""
        prec = getattr(self.args, "precision", "fp32")
        p = prec if (prec != "fp32" or "DPI_PRECISION" not in os.environ) else None
        st = ("bf16" if (prec == "bf16" and self.storage_bf16_ok()) else "fp32") if (prec != "fp32" or "DPI_STORAGE" not in os.environ) else None
        return ops.mode_scope(p, st)

    def wants_weight_grad_overlap(self):
        """Weight gradients on side streams next to the backward-data chain (and with them the once-per-step join and the ResPath branch
        stream, ops.py): patches of >= 2^20 voxels, eager.  fp32: 32.3 -> 29.6 ms at 256x128x128 (round 5).  bf16 storage: round 4 measured
        20.2 ms with the side stream against 18.9 without (per-node joins) and kept it off; with the round-5 schedule the order is
        reversed — 15.8-16.1 ms eager with it, 16.2 ms replayed from a graph without, 17.1-17.3 eager without; configs[4] (512x256x256
        patches) 8.4-8.8 against 8.2 it/s — so it is on for every storage type."""
        big = int(np.prod(self.img.shape[:-1])) >= (1 << 20)
        if os.environ.get("DPI_FORCE_OVERLAP") in ("0", "1"):       # A/B knob
            return big and os.environ["DPI_FORCE_OVERLAP"] == "1"
        return big

    def storage_bf16_ok(self):
        """Whether this net runs with bf16 activations: the 3-D MultiRes-UNet whose blocks all execute as the fused nodes
        (LeakyReLU, no dropout) — Block3dFn / SkipJoinFn / ConvBnActFn allocate the tensors and every kernel under them takes the
        storage type.  Other nets keep fp32 storage under --precision bf16 (operand rounding only, as --precision bf16mm)."""
        ok = getattr(self, "_storage_ok", None)
        if ok is None:
            from .architectures import mulresunet as M
            a = self.args
            # (the data-forgetting term adds the fp32 data to the network input on the host side of the node boundary: fp32 storage)
            ok = bool(self.net is not None and a.datadim == "3d" and getattr(a, "net", "multiunet") in ("multiunet", "load") and M.FUSE_BLOCKS
                      and a.data_forgetting_factor == 0)
            if ok:
                blocks = [m for m in self.net.modules() if isinstance(m, (M.MultiResBlock, M.ResPath))]
                ok = bool(blocks) and all(m.nd == 3 and m._fusable() for m in blocks)
            if self.net is not None:
                self._storage_ok = ok
        return ok

    def optimization_loop(self, net_input=None):
        try:
            with self.precision_scope():
                return self._optimization_loop(net_input)
        except BaseException:
            ops.abort_iteration()          # streams joined, per-iteration state of ops forgotten: a later bare backward() joins per node again
            raise

    def _optimization_loop(self, net_input=None):
        ops.begin_iteration()
        input_ = self.perturbed_input() if net_input is None else net_input
        if self.iiter < self.args.data_forgetting_factor:       # main.py:153-155
            if input_ is self.input_ or net_input is not None:
                input_ = input_.clone()
            _lib.check(_lib.load().dpi_axpy(float(self.add_data_weight[self.iiter]), _lib.ptr(self.add_data_), input_.numel(),
                                            _lib.ptr(input_), _lib.stream()), "dpi_axpy")
            self.input_list.append(u.torch_to_np(input_, True))
        out_ = self.net(input_)
        total_loss, metrics = ops.masked_loss(out_, self.img_, self.mask_, self.loss_kind)
        reg = self.regularization(out_, total_loss)          # None, or (weight tensor / float, reg loss) of a subclass / add-on
        if reg is None:
            total_loss.backward()
            ops.finish_backward(self._grad_params())
            l, s, p = metrics[:3].tolist()          # one read-back for loss, snr, pcorr
            self.history.append((l, s, p))
        else:
            eps, reg_loss = reg
            total = total_loss + eps * reg_loss
            total.backward()
            ops.finish_backward(self._grad_params())
            main_l, s, p = metrics[:3].tolist()
            l, r = float(total.item()), float(reg_loss.item())
            self.history.append((l, main_l, r, s, p))           # HistoryReg layout (main_pocs.py:198-202)
        self.history.lr.append(self.optimizer.param_groups[0]["lr"])
        if self.iiter == 0 or l <= self.loss_min:
            self.loss_min = l
            self._out_best_dev = out_.detach()   # stays on the GPU; copied to the host once, after the loop
        if self.iiter in self.iter_to_be_saved and self.iiter != 0:
            np.save(os.path.join(self.outpath, self.image_name.split(".")[0]
                                 + "_output%s.npy" % str(self.iiter).zfill(self.zfill)), self._to_numpy_out(out_))
        self.iiter += 1
        return l

    def _grad_params(self):
        """The parameters ops.finish_backward() checks the deferred gradients against (only when a side / branch stream ran: big patches)."""
        return self.optimizer._params if (ops._deferred and self.optimizer is not None) else None

    def regularization(self, out_, main_loss):
        """Extra loss term hook.  The base path has none.  With --aa_weight > 0 (2-D / 2.5-D sections) the anti-aliasing add-on
        is active: aa_weight * loss_fn(Hale2D(dips)(out), 0) — the directional Laplacian of utils/slopes.py along dips estimated
        once per patch from the Gaussian-smoothed structure tensor of the decimated data (build_regularizer)."""
        if getattr(self, "_aa_op", None) is None:
            return None
        lap = self._aa_op(out_)
        reg_loss, _ = ops.masked_loss(lap, self._aa_zero, self._aa_one, self.loss_kind)
        return float(self.args.aa_weight), reg_loss

    def build_regularizer(self):
        """Anti-aliasing add-on (BASELINE configs[3]; the reference ships the operators — utils/slopes.py, operators/ — but no
        caller): dips from the structure tensor of the available traces, smoothed with --aa_smooth."""
        self._aa_op = None
        a = self.args
        if not getattr(a, "aa_weight", 0.0):
            return
        if self.img_.ndim != 4:
            raise NotImplementedError("the anti-aliasing add-on works on 2-D / 2.5-D sections (BCHW)")
        if a.aa_dips is not None:
            dips = torch.from_numpy(np.load(a.aa_dips).astype(np.float32)).to(self.device).reshape(self.img_.shape)
        else:
            dips, _ = u.structure_tensor_dips(self.img_ * self.mask_, smooth=float(a.aa_smooth))
        self._aa_op = u.Hale2D(dips)
        self._aa_zero = torch.zeros_like(self.img_)
        self._aa_one = torch.ones_like(self.img_)
        self.history = u.HistoryReg(a.epochs)

    def optimize(self, net_inputs=None, verbose=True, mode="auto", check_every=64):
        """Adam loop with optional ReduceLROnPlateau and EarlyStopping (main.py:195-220).

        mode "eager": the reference's control flow, one host read-back of {loss, snr, pcorr} per iteration.
        mode "graph": iteration 0 runs eagerly, iteration 1 is captured into a hipGraph and replayed; history, best-output
        tracking, plateau LR and early stopping live on the device (dpi_loop_control / dpi_copy_if), the host only polls
        the `active` flag every `check_every` replays.  Same arithmetic, same stopping iteration.
        "auto" picks "graph" unless per-iteration host work was requested (net_inputs, --save_every) or the patch has >= 2^20 voxels
        (eager with the weight-gradient and branch streams)."""
        a = self.args
        big = self.wants_weight_grad_overlap()
        if mode == "auto":
            # big fp32 patches are GPU-bound either way and gain from overlapping the weight gradients (eager only);
            # small ones are launch-bound without a graph
            # big patches run eager with the weight-gradient / branch streams (`big`).  With those switched off (DPI_FORCE_OVERLAP=0) a
            # big patch is captured only for long runs: the capture allocates a memory pool of its own — as much again as one
            # iteration's activations, 20 GB at 512x256x256 — and costs 0.1-3 s of hipMalloc / instantiation depending on the box (measured
            # on configs[4], 30 iterations per patch: 160-210 ms per iteration with the capture, 113 ms eager)
            large = int(np.prod(self.img.shape[:-1])) >= (1 << 20)
            mode = "eager" if (net_inputs is not None or a.save_every is not None or a.epochs < 3 or self.has_regularizer()
                               or a.data_forgetting_factor != 0 or self.noise_source() != "philox" or big or (large and a.epochs < 1000)) else "graph"
        self.optimizer = FusedAdam(self.net.parameters(), lr=a.lr)
        ops.set_weight_grad_overlap(big, in_graph=(mode == "graph" and big))
        start = time()
        if mode == "graph" and self.noise_source() != "philox":
            raise ValueError("--noise_source torch_cpu draws on the host every iteration: it cannot be captured (mode='eager')")
        if mode == "graph":
            self._optimize_graph(verbose, check_every)
        else:
            sched = DevicePlateau(self.optimizer, a.lr_factor, a.lr_thresh, a.lr_patience) if a.reduce_lr else None
            stopper = u.EarlyStopping(patience=a.earlystop_patience, min_delta=a.earlystop_min_delta, percentage=True)
            for j in range(a.epochs):
                self.optimizer.zero_grad()
                loss = self.optimization_loop(None if net_inputs is None else net_inputs[j])
                self.optimizer.step()
                if sched is not None:
                    sched.step(loss)
                if verbose:
                    print(self.history.log_message(self.iiter - 1), "\r", end="")
                if stopper.step(loss):
                    break
            torch.cuda.synchronize(self.device)
            self.out_best = self._to_numpy_out(self._out_best_dev)
        self.elapsed = time() - start
        if verbose:
            print("\n" + u.sec2time(self.elapsed))

    # ---- hipGraph path -------------------------------------------------------------------------------------
    def graph_prepare(self, quiet_device=True):
        """Allocate the device-resident loop state, run iteration 0 eagerly and capture iteration 1.
        Returns the captured graph; `graph_finish()` reads history / best output back.
        quiet_device=False (optimize_concurrently, round 5): other streams keep replaying the graphs of other patches meanwhile — only THIS
        stream is synchronised before the capture and the capture is opened with CUDAGraph.capture_begin() directly (torch.cuda.graph()
        would synchronise the whole device, collect garbage and empty the allocator's cache first)."""
        a = self.args
        if self.noise_source() != "philox":
            raise ValueError("--noise_source torch_cpu draws on the host every iteration: it cannot be captured into a graph")
        L = _lib.load()
        dev = self.device
        if self.optimizer is None:
            self.optimizer = FusedAdam(self.net.parameters(), lr=a.lr)
            # prepared directly (optimize_concurrently, bench.py): the weight-gradient side streams follow THIS patch's size, not
            # whatever an earlier optimize() in the process left behind; optimize() / bench.py set it before they
            # call graph_prepare with an optimiser of their own
            big = self.wants_weight_grad_overlap()
            ops.set_weight_grad_overlap(big, in_graph=big)
        opt = self.optimizer
        self._g_state = torch.zeros(8, dtype=torch.float64, device=dev)
        self._g_state[2] = float("inf")
        self._g_hist = torch.zeros(a.epochs * 4, dtype=torch.float64, device=dev)
        self._g_improved = torch.zeros(1, dtype=torch.int32, device=dev)
        self._g_best = None
        kind = self.loss_kind

        def one_iteration():
            with self.precision_scope():
                _one_iteration()

        def _one_iteration():
            ops.begin_iteration()
            opt.zero_grad()
            out_ = self.net(self.perturbed_input())
            loss, metrics = ops.masked_loss(out_, self.img_, self.mask_, kind)
            loss.backward()
            ops.finish_backward(self._grad_params())
            opt.step()                                   # skipped on the device once `active` is 0
            if self._g_best is None:
                self._g_best = torch.empty_like(out_)
            _lib.check(L.dpi_loop_control(_lib.ptr(metrics), _lib.ptr(self._g_state), _lib.ptr(self._g_hist), a.epochs,
                                          _lib.ptr(opt.step_lr), _lib.ptr(opt.active), _lib.ptr(self._g_improved),
                                          int(bool(a.reduce_lr)), float(a.lr_factor), float(a.lr_thresh), int(a.lr_patience), 0.0, 1e-8,
                                          int(a.earlystop_patience), float(a.earlystop_min_delta), _lib.stream()), "dpi_loop_control")
            _lib.check(L.dpi_copy_if(_lib.ptr(self._g_improved), _lib.ptr(out_), _lib.ptr(self._g_best), out_.numel(),
                                     _lib.stream()), "dpi_copy_if")

        one_iteration()                                  # iteration 0, eager (also warms every lazy cache)
        if quiet_device:
            torch.cuda.synchronize(dev)
        else:
            torch.cuda.current_stream(dev).synchronize()
        opt.prepare_capture()
        graph = torch.cuda.CUDAGraph()
        if quiet_device:
            with torch.cuda.graph(graph):
                one_iteration()                          # recorded, not executed
        else:
            graph.capture_begin(capture_error_mode="thread_local")     # on the current (non-default) stream, a private memory pool of its own;
                                                                       # thread_local: another host thread may keep launching its patches' replays
            try:
                one_iteration()
            finally:
                graph.capture_end()
        self._graph = graph
        return graph

    def graph_finish(self, quiet_device=True):
        """quiet_device=False: wait for the CURRENT stream only (other patches keep running on theirs)."""
        if quiet_device:
            torch.cuda.synchronize(self.device)
        else:
            torch.cuda.current_stream(self.device).synchronize()
        n = int(self._g_state[0].item())
        h = self._g_hist[:4 * n].view(n, 4).cpu().numpy()
        self.history = u.History(self.args.epochs)
        self.history.loss, self.history.snr, self.history.pcorr, self.history.lr = (h[:, i].tolist() for i in range(4))
        self.loss_min = float(self._g_state[1].item())
        self.iiter = n
        self._out_best_dev = self._g_best
        self.out_best = self._to_numpy_out(self._g_best)

    def _optimize_graph(self, verbose, check_every):
        graph = self.graph_prepare()
        for j in range(1, self.args.epochs):
            graph.replay()
            if j % check_every == 0:
                if int(self.optimizer.active.item()) == 0:     # early stop / NaN decided on the device
                    break
                if verbose:
                    n = int(self._g_state[0].item())
                    l, s_, p_ = self._g_hist[4 * (n - 1):4 * (n - 1) + 3].tolist()
                    print("Iter %d, Loss = %+.2e, SNR = %+2.2f dB, PCORR = %+.2f %%" % (n, l, s_, p_ * 100), "\r", end="")
        self.graph_finish()

    def has_regularizer(self):
        return getattr(self, "_aa_op", None) is not None

    def graph_capable(self):
        """True when optimize(mode='auto') would take the hipGraph path for the loaded patch."""
        a = self.args
        return not (a.save_every is not None or a.epochs < 3 or a.data_forgetting_factor != 0 or self.has_regularizer()
                    or self.noise_source() != "philox"
                    or int(np.prod(self.img.shape[:-1])) >= (1 << 20))

    # ------------------------------------------------------------------------------------------
    def save_result(self):
        np.save(os.path.join(self.outpath, self.image_name + "_run.npy"), {
            "device": u.get_gpu_name(), "elapsed": u.sec2time(self.elapsed), "outpath": self.outpath,
            "history": self.history, "mask": self.mask, "image": self.img, "output": self.out_best,
            "noise": self.input_list,
        })
        if self.args.savemodel:
            torch.save(self.net.state_dict(), os.path.join(self.outpath, self.image_name + "_model.pth"))

    def clean(self):
        self.iiter = 0
        self.loss_min = None
        self._out_best_dev = None
        self.history = self._new_history()

    def _new_history(self):
        return u.HistoryReg(self.args.epochs) if self.has_regularizer() else u.History(self.args.epochs)


def optimize_concurrently(Ts, check_every=64, prepare=None, timings=None):
    """Optimise several Interpolators at the same time on ONE GPU: each iteration is a captured hipGraph, the graphs are replayed
    round-robin on one stream per patch.  A 64^3 patch keeps only a fraction of an MI355X busy (its coarse levels are a few workgroups
    wide), six of them side by side give ~1.8x the patch-iterations per second of one (tools/c3_probe.py).  Same arithmetic per patch as
    optimize().

    prepare(T) (round 5): the host-side set-up of a patch — weights, z (Interpolator.build_model / build_input; ~30 ms of host time) — is
    done HERE, patch by patch, and between two patches the graphs that are already captured get a burst of replays: the device works on
    the first patches while the host prepares the next ones, instead of idling through the set-up of the whole group (6 x ~55 ms per
    group of 100-iteration patches).  Without `prepare` the Interpolators come prepared (data, model and input built).
    timings (dict): `prepare_s` = host seconds spent in prepare + capture (they overlap device work, so they are not additive to the loop)."""
    if not Ts:
        return
    start = time()
    epochs = Ts[0].args.epochs
    streams = [torch.cuda.Stream(device=T.device) for T in Ts]
    graphs, left = [], []
    t_prep = 0.0

    def burst(n):
        """n rounds of replays over the patches captured so far (asynchronous: ~0.2 ms of host time per replay)."""
        for _ in range(n):
            for k, g in enumerate(graphs):
                if left[k] > 0:
                    with torch.cuda.stream(streams[k]):
                        g.replay()
                    left[k] -= 1
    for T, st in zip(Ts, streams):
        t0 = time()
        if prepare is not None:
            prepare(T)
        st.wait_stream(torch.cuda.current_stream(T.device))      # z / data / weights were produced on the caller's stream
        with torch.cuda.stream(st):
            T.optimizer = None
            graphs.append(T.graph_prepare(quiet_device=prepare is None))
        left.append(epochs - 1)
        t_prep += time() - t0
        if prepare is not None and len(graphs) < len(Ts):
            burst(min(check_every, 24))       # what the device chews on while the next patch is being prepared (bounded: the early-stop poll below)
    alive = [True] * len(Ts)
    j = 0
    while any(a and n > 0 for a, n in zip(alive, left)):
        for k, (g, st) in enumerate(zip(graphs, streams)):
            if alive[k] and left[k] > 0:
                with torch.cuda.stream(st):
                    g.replay()
                left[k] -= 1
        j += 1
        if j % check_every == 0:
            # poll every patch on ITS stream: the read-back waits for the replays queued so far, so the host never runs more than
            # check_every replays ahead of the device and stops replaying a patch whose loop ended on the device (early stop / NaN)
            for k, (T, st) in enumerate(zip(Ts, streams)):
                if alive[k]:
                    with torch.cuda.stream(st):
                        alive[k] = int(T.optimizer.active.item()) != 0
    for T, st in zip(Ts, streams):
        with torch.cuda.stream(st):
            T.graph_finish()
        torch.cuda.current_stream(T.device).wait_stream(st)      # consumers (overlap-add, save_result) run on the caller's stream
        T.elapsed = time() - start
    if timings is not None:
        timings["prepare_s"] = timings.get("prepare_s", 0.0) + t_prep


def main(argv=None):
    args = parse_arguments(argv)
    u.set_gpu(args.gpu if args.gpu is not None else -1)
    u.set_seed(0)
    outpath = os.path.join("./results/", args.outdir if args.outdir is not None else u.random_code())
    os.makedirs(outpath, exist_ok=True)
    print("Saving to %s" % outpath)
    u.write_args(os.path.join(outpath, "args.txt"), args)
    patches = extract_patches(args)
    print("Processing %d patches" % len(patches))
    T = Interpolator(args, outpath)
    for i, patch in enumerate(patches):
        std = T.load_data(patch)
        print("\nThe data shape is %s, the std of coarse data is %.2e" % (str(patch["image"].shape), std))
        if np.isclose(std, 0.0, atol=1e-12):
            print("skipping...")
            T.out_best = T.img * T.mask
            T.elapsed = 0.0
        else:
            T.begin_patch(i)
            if T.net is None or not args.start_from_prev:
                T.build_model(netpath=args.netdir[i]) if len(args.netdir) != 0 else T.build_model()
            T.build_input()
            T.build_regularizer()
            T.optimize()
        T.save_result()
        T.clean()
    print("Interpolation done! Saved to %s" % outpath)


if __name__ == "__main__":
    main()
