"""Import shim for the upstream reference (TEST INFRASTRUCTURE, build container only).

The reference at /root/reference is pure Python but does not import on this image
unmodified: it needs `np.float` (utils/metrics.py:8,22; utils/pocs.py:13-14) and the
absent packages GPUtil / termcolor (utils/torch.py:3-4), skimage (utils/patch_extractor.py:8),
cv2 (utils/mask.py:3, utils/plotting.py:6), imageio (utils/plotting.py:5) and torchvision
(architectures/convgru.py:5).  This module registers minimal stand-ins for those THIRD-PARTY
packages in `sys.modules` and puts the reference on `sys.path`, so that `oracle/make_golden.py`
can run the reference itself and record golden vectors under tests/golden/.

Nothing here is shipped or used by the product path; /root/reference does not exist on the
GPU box, so only make_golden.py (run here, by hand) imports this file.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("DPI_REFERENCE_ROOT", "/root/reference")


def _windows(arr, window_shape, step):
    """skimage.util.view_as_windows by its documented definition: window w starts at w*step."""
    window_shape = tuple(int(s) for s in window_shape)
    if isinstance(step, int):
        step = (step,) * arr.ndim
    v = np.lib.stride_tricks.sliding_window_view(arr, window_shape)
    return v[tuple(slice(None, None, int(s)) for s in step)]


def _blocks(arr, block_shape):
    return _windows(arr, block_shape, tuple(block_shape))


def install():
    """Idempotently install the stubs and make the reference importable."""
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if not hasattr(np, "float"):
        np.float = float  # noqa: annotation-only use in the reference

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("GPUtil", getFirstAvailable=lambda *a, **k: [0], getGPUs=lambda: [])
    mod("termcolor", colored=lambda s, *a, **k: s)
    mod("cv2", dilate=None, resize=None)
    mod("imageio", mimsave=None)
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    sk = mod("skimage")
    sk.util = mod("skimage.util", view_as_windows=_windows, view_as_blocks=_blocks)
    sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def load_main():
    """Load reference main.py as a module (it calls u.set_seed() on import, main.py:15)."""
    install()
    spec = importlib.util.spec_from_file_location("ref_main", os.path.join(REFERENCE_ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def parse_args(argv):
    """Run the reference's own parser (parameter.py:4) on argv, then fix netdir (main.py:105)."""
    install()
    import parameter  # reference module
    old = sys.argv
    sys.argv = ["main.py"] + list(argv)
    try:
        args = parameter.parse_arguments()
    finally:
        sys.argv = old
    if args.netdir is None:
        args.netdir = []
    return args
