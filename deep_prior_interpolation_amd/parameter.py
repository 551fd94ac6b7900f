"""Command-line flags — drop-in for reference parameter.py (same flag names, defaults and post-processing,
parameter.py:4-130), table-driven.  Differences, both fixing upstream crashes:
  * --netdir defaults to [] (upstream: None -> len(None) TypeError at main.py:105,287);
  * 'skip' is an accepted --net choice (upstream builds Skip3D for it but the parser rejects it).
"""
from argparse import ArgumentParser, Namespace

_ACT = ["LeakyReLU", "ReLU", "ELU", "Tanh", "Sigmoid"]

# (flags, kwargs)
_FLAGS = [
    # dataset
    (("--imgdir",), dict(type=str, required=True, default="./datasets/", help="Directory containing the processed data")),
    (("--outdir",), dict(type=str, required=False, help="Subfolder in ./results/ for saving.")),
    (("--imgname",), dict(type=str, help="The name of original images")),
    (("--maskname",), dict(type=str, help="The name of corrupted images")),
    (("--gain",), dict(type=float, required=False, default=2e3, help="gain for the input")),
    (("--datadim",), dict(type=str, required=False, default="2d", choices=["2d", "2.5d", "3d"],
                          help="The dimensionality of the data")),
    (("--slice",), dict(type=str, required=False, default="xy", choices=["tx", "ty", "xy"],
                        help="The type of slice of 3D data when datadim=2.5d")),
    (("--imgchannel",), dict(type=int, required=False, help="Number of 2.5d patches to be stacked in the channel dimension.")),
    (("--adirandel",), dict(type=float, required=False, default=0.0, help="The percent of addictive random deleting samples")),
    (("--padwidth",), dict(type=int, required=False, default=0, help="The padding width to the process data using edge mode")),
    (("--patch_shape",), dict(nargs="+", type=int, required=False, help="Patch shape to be processed (2D, 2.5D, 3D)")),
    (("--patch_stride",), dict(nargs="+", type=int, required=False, help="Patch stride for the extraction (2D, 2.5D, 3D)")),
    # network design
    (("--net",), dict(type=str, required=False, default="multiunet",
                      choices=["multiunet", "attmultiunet", "part", "unet", "load", "skip"], help="The network architecture")),
    (("--gpu",), dict(type=int, required=False, help="GPU to use (default lowest memory usage)")),
    (("--activation",), dict(type=str, default="LeakyReLU", required=False, choices=_ACT,
                             help="Activation function to be used in the convolution block")),
    (("--last_activation",), dict(type=str, required=False, choices=_ACT, help="Activation function to the network output")),
    (("--dropout",), dict(type=float, default=0.0, required=False, help="Dropout rate to be applied in each convolution")),
    (("--filters",), dict(nargs="+", type=int, required=False, default=[16, 32, 64, 128, 256],
                          help="Numbers of channels in every layer of encoder and decoder")),
    (("--skip",), dict(nargs="+", type=int, required=False, default=[16, 32, 64, 128],
                       help="Number of channels for skip-connection")),
    (("--inputdepth",), dict(type=int, required=False, default=64, help="Depth of the input noise tensor")),
    (("--upsample",), dict(type=str, required=False, default="nearest", choices=["nearest", "linear"],
                           help="Network's upgoing deconvolution strategy")),
    (("--inittype",), dict(type=str, required=False, default="xavier",
                           choices=["xavier", "normal", "default", "kaiming", "orthogonal"],
                           help="Initialization strategy for the network weights")),
    (("--initgain",), dict(type=float, required=False, default=0.02,
                           help="Initialization scaling factor for normal, xavier and orthogonal.")),
    (("--savemodel",), dict(action="store_true", default=False, help="Save the optimized model to disk")),
    (("--netdir",), dict(type=str, nargs="+", required=False, default=[], help="Path for loading the optimized network")),
    # input noise
    (("--param_noise",), dict(action="store_false", help="Add normal noise to the parameters every epoch")),
    (("--reg_noise_std",), dict(type=float, required=False, default=0.03,
                                help="Standard deviation of the normal noise to be added to the input every epoch")),
    (("--noise_dist",), dict(type=str, default="n", required=False, choices=["n", "u", "c"],
                             help="Type of noise for the input tensor [(n)ormal, (u)niform, (c)auchy]")),
    (("--noise_std",), dict(type=float, default=0.1, required=False, help="Standard deviation of the noise for the input tensor")),
    (("--noise_source",), dict(type=str, default="philox", required=False, choices=["philox", "torch_cpu"],
                             help="ours: philox = z and the per-iteration perturbation drawn on the GPU (counter-based, the default); torch_cpu = parity "
                                  "mode: both drawn by torch's CPU generator in the reference's order (main.py:59-64,143-150, including the "
                                  "--param_noise draws that only shift the stream) and uploaded: same seed, same input stream as the reference")),
    (("--data_forgetting_factor",), dict(type=int, default=0, required=False,
                                         help="Duration of additional decimated data to the input noise tensor")),
    (("--filter_noise_with_wavelet",), dict(action="store_true", default=False,
                                            help="Filter input noise tensor with the wavelet bandwidth")),
    (("--lowpass_fs",), dict(type=float, required=False, help="Butterworth LPF on the input noise: sampling frequency")),
    (("--lowpass_fc",), dict(type=float, required=False, help="Butterworth LPF on the input noise: cutoff frequency")),
    (("--lowpass_ntaps",), dict(type=int, required=False, default=7, help="Low pass filter lenght")),
    # training
    (("--loss",), dict(type=str, required=False, choices=["mae", "mse"], default="mae", help="Loss function to be used.")),
    (("--epochs", "-e", "--iter"), dict(type=int, required=False, default=2001, help="Number of optimization iterations")),
    (("--lr",), dict(type=float, default=1e-3, required=False, help="Learning Rate for Adam optimizer")),
    (("--lr_factor",), dict(type=float, default=0.9, required=False, help="LR reduction for Plateau scheduler.")),
    (("--lr_thresh",), dict(type=float, default=1e-5, required=False, help="LR threshold for Plateau scheduler.")),
    (("--lr_patience",), dict(type=int, default=100, required=False, help="LR patience for Plateau scheduler.")),
    (("--save_every",), dict(type=int, required=False, help="Number of epochs every which to save the results")),
    (("--start_from_prev",), dict(action="store_true", default=False, help="Start training from previous patch")),
    (("--reduce_lr",), dict(action="store_true", default=False, help="Use ReduceLROnPlateau scheduler")),
    (("--earlystop_patience",), dict(type=int, required=False, help="Early stopping patience")),
    (("--earlystop_min_delta",), dict(type=float, required=False, default=1.0, help="Early stopping min percentage delta")),
    # mixed precision (ours; BASELINE configs[4]): bf16 operands / fp32 accumulate in the 3x3(x3) convolutions, everything else fp32
    (("--precision",), dict(type=str, required=False, default="fp32", choices=["fp32", "bf16", "bf16mm", "split"],
                          help="fp32 (reference); bf16 = BASELINE configs[4]: activations and their gradients STORED as bf16 (3-D MultiRes-UNet), "
                               "bf16 operands in the 3x3x3 convolutions, fp32 accumulation / master weights / BatchNorm statistics / Adam; "
                               "bf16mm = bf16 operands only, fp32 storage; split = fp32-accurate three-term bf16 split.  bf16 STORAGE applies to the 3-D MultiRes-UNet "
                               "with LeakyReLU and no dropout (the fused nodes); 2-D / 2.5-D nets, Skip3D, UNet, ELU / dropout nets and runs with "
                               "--data_forgetting_factor keep fp32 tensors under --precision bf16 (operand rounding only, as bf16mm)")),
    # anti-aliasing add-on (ours: the reference ships operators/ + utils/slopes.py without a caller)
    (("--aa_weight",), dict(type=float, required=False, default=0.0, help="Weight of the directional-Laplacian regulariser (0 = off)")),
    (("--aa_smooth",), dict(type=float, required=False, default=2.0, help="Gaussian smoothing (std, samples) of the structure tensor")),
    (("--aa_dips",), dict(type=str, required=False, help="Optional .npy with a precomputed dip field (same shape as a section)")),
    # POCS regulariser (main_pocs.py)
    (("--pocs_alpha",), dict(type=float, required=False, default=0.1, help="POCS data weighting.")),
    (("--pocs_thresh",), dict(type=float, required=False, default=5.0, help="POCS thresholding percentage")),
    (("--pocs_weight",), dict(type=float, required=False, help="POCS regularization weight")),
]

_MUST_MATCH = ["datadim", "slice", "imgchannel", "patch_shape", "inputdepth", "loss", "lr", "lr_factor", "lr_thresh",
               "lr_patience", "reduce_lr"]
_OVERRIDDEN = ["net", "activation", "last_activation", "dropout", "filters", "skip", "upsample", "inittype", "initgain"]


def build_parser() -> ArgumentParser:
    parser = ArgumentParser()
    for flags, kw in _FLAGS:
        parser.add_argument(*flags, **kw)
    return parser


def postprocess(args: Namespace) -> Namespace:
    """Derived defaults of parameter.py:113-125."""
    if args.upsample == "linear":
        args.upsample = "trilinear" if args.datadim == "3d" else "bilinear"
    if args.patch_shape is None:
        args.patch_shape = [-1, -1] if args.datadim == "2d" else [-1, -1, -1]
    if args.patch_stride is None:
        args.patch_stride = args.patch_shape
    if args.earlystop_patience is None:
        args.earlystop_patience = args.epochs
    if args.netdir is None:
        args.netdir = []
    return args


def parse_arguments(argv=None) -> Namespace:
    return postprocess(build_parser().parse_args(argv))


def net_args_are_same(args1: Namespace, args2: Namespace) -> bool:
    """Compatibility of a saved args.txt with the current run before loading its weights (parameter.py:133-173)."""
    a, b = vars(args1), vars(args2)
    errors = [k for k in _MUST_MATCH if a[k] != b[k]]
    warns = [k for k in _OVERRIDDEN if a[k] != b[k]]
    if errors:
        print("The following arguments keys have to be the same:\n\t")
        print(", ".join(errors))
        return False
    if warns:
        print("\nThe following arguments are different, but they are overridden by the network loading:")
        print("\t", ", ".join(warns))
    return True
