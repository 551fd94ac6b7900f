"""Network factory — drop-in for reference architectures/__init__.py:10-86.

`get_net(args, outchannel)` consumes the same Namespace fields (datadim, net, inputdepth, filters, skip,
upsample, activation, last_activation, dropout) and returns an nn.Module whose state_dict keys equal the
reference's, built from HIP-backed leaf modules.
"""
from .base import *            # noqa: F401,F403
from .mulresunet import *      # noqa: F401,F403
from .skip import *            # noqa: F401,F403
from .unet import *            # noqa: F401,F403
from .unet import UNet
from .mulresunet import MulResUnet, MulResUnet3D
from .skip import Skip, Skip3D

_OUT_OF_SCOPE = {
    "attmultiunet": "AttMulResUnet2D (attention.py) is outside the hot-path scope (SURVEY §2 row 4f)",
    "part": "PartialUNet.forward(x, mask) cannot be called by Interpolator (SURVEY §2 row 4g)",
}


def get_net(args, outchannel=1):
    net_name = getattr(args, "net", "multiunet")
    if net_name in _OUT_OF_SCOPE:
        raise NotImplementedError("--net %s: %s" % (net_name, _OUT_OF_SCOPE[net_name]))
    common = dict(num_input_channels=args.inputdepth, num_output_channels=outchannel, upsample_mode=args.upsample,
                  need_bias=True, act_fun=args.activation, last_act_fun=args.last_activation, dropout=args.dropout)
    if args.datadim in ("2d", "2.5d"):
        if net_name == "unet":
            # upstream names an undefined `UNetMod` here (architectures/__init__.py:13); the intended class is unet.UNet
            return UNet(num_input_channels=args.inputdepth, num_output_channels=outchannel, filters=args.filters,
                        upsample_mode=args.upsample, need_bias=True, act_fun=args.activation,
                        last_act_fun=args.last_activation, dropout=args.dropout)
        if net_name == "skip":
            # BASELINE configs[3] names the 2.5-D skip net.  Upstream cannot reach its 2-D `Skip` (architectures/skip.py:5-48): `--net` has
            # no such choice and get_net falls through to MulResUnet (architectures/__init__.py:41-53).  Documented deviation, like
            # `--net skip` in 3-D: same argument mapping as the Skip3D branch (architectures/__init__.py:62-72), which needs one skip
            # width per scale (len(--skip) == len(--filters))
            if len(args.skip) != len(args.filters):
                raise ValueError("--net skip needs one --skip width per --filters scale (got %d and %d)" % (len(args.skip), len(args.filters)))
            return Skip(num_channels_down=args.filters, num_channels_up=args.filters, num_channels_skip=args.skip, **common)
        return MulResUnet(num_channels_down=args.filters, num_channels_up=args.filters, num_channels_skip=args.skip, **common)
    if net_name == "skip":
        return Skip3D(num_channels_down=args.filters, num_channels_up=args.filters, num_channels_skip=args.skip, **common)
    return MulResUnet3D(num_channels_down=args.filters, num_channels_up=args.filters, num_channels_skip=args.skip, **common)
