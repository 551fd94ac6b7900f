"""Patch I/O — drop-in for reference data.py: npy loading, NaN -> binary mask, patch extraction with the
2.5-D slab transposes, and overlap-add reassembly from the per-patch result files.

`reconstruct_patches` sorts the result files by patch name (the reference relies on directory order of an
unsorted glob, data.py:99 — SURVEY App. B.7) and can run the overlap-add on the GPU (`device=`)."""
import os
from glob import glob
from typing import List

import numpy as np

from . import utils as u

__all__ = ["extract_patches", "reconstruct_patches", "patch_extractor_for", "transpose_patches_25d"]


def patch_extractor_for(in_shape, patch_shape, patch_stride, datadim, imgchannel=None) -> u.PatchExtractor:
    """-1 in --patch_shape means 'whole axis'; in 2.5-D the last patch axis is the channel stack (data.py:8-17)."""
    ndim = len(in_shape)
    shape = [patch_shape[d] if patch_shape[d] != -1 else in_shape[d] for d in range(ndim)]
    if datadim == "2.5d" and imgchannel is not None:
        shape[-1] = imgchannel
    stride = [patch_stride[d] if patch_stride[d] != -1 else shape[d] for d in range(len(shape))]
    return u.PatchExtractor(dim=tuple(shape), stride=tuple(stride))


_get_patch_extractor = patch_extractor_for   # reference name

_TO_SLAB = {"xy": (0, 2, 3, 1), "ty": (0, 1, 3, 2)}   # B,T,X,Y -> B,X,Y,T  /  B,T,Y,X
_FROM_SLAB = {"xy": (0, 3, 1, 2), "ty": (0, 1, 3, 2)}


def transpose_patches_25d(in_content, slice="XY", adj=False):
    """Bring the axis that becomes the conv channel to the end (adj=True undoes it); 'tx' is the identity."""
    s = {"xt": "tx", "yt": "ty"}.get(slice.lower(), slice.lower())
    table = _FROM_SLAB if adj else _TO_SLAB
    return in_content.transpose(table[s]) if s in table else in_content


_transpose_patches_25d = transpose_patches_25d


def extract_patches(args) -> List[dict]:
    """List of {'image': patch*gain, 'mask': binary patch, 'name': zero-padded index} (data.py:44-84)."""
    original = np.load(os.path.join(args.imgdir, args.imgname), allow_pickle=True)
    corrupted = np.load(os.path.join(args.imgdir, args.maskname), allow_pickle=True)
    assert original.shape == corrupted.shape, "Original and Corrupted data must have the same dimension"
    assert original.ndim in [2, 3], "Data volumes have to be 2D or 3D"
    if np.isnan(corrupted).any():
        corrupted = u.bool2bin(corrupted)
    pe = patch_extractor_for(original.shape, args.patch_shape, args.patch_stride, args.datadim, args.imgchannel)
    if args.datadim == "2.5d" or (args.datadim == "2d" and pe.ndim == 3):
        final_shape = (-1,) + pe.dim              # last axis = channels
    else:
        final_shape = (-1,) + pe.dim + (1,)
    img = pe.extract(original).reshape(final_shape)
    msk = pe.extract(corrupted).reshape(final_shape)
    if args.datadim == "2.5d":
        img = transpose_patches_25d(img, args.slice)
        msk = transpose_patches_25d(msk, args.slice)
    width = u.ten_digit(img.shape[0])
    out = []
    for p in range(img.shape[0]):
        m = msk[p]
        if args.adirandel > 0:
            m = u.add_rand_mask(m, args.adirandel)
        out.append({"image": img[p] * args.gain, "mask": m, "name": str(p).zfill(width)})
    return out


def reconstruct_patches(args, return_history=False, verbose=False, results_root="./results"):
    """Re-assemble the volume from ./results/<outdir>/<name>_run.npy (data.py:87-130)."""
    inputs = np.load(os.path.join(args.imgdir, args.imgname), allow_pickle=True)
    pe = patch_extractor_for(inputs.shape, args.patch_shape, args.patch_stride, args.datadim, args.imgchannel)
    pe.extract(inputs)
    pa_shape = u.patch_array_shape(inputs.shape, pe.dim, pe.stride)
    files = [p for p in glob(os.path.join(results_root, args.outdir) + "/*.npy")
             if "output" not in os.path.basename(p)]
    files.sort(key=lambda p: os.path.basename(p))
    outs, elapsed, history, dev = [], [], [], "?"
    for path in files:
        rec = np.load(path, allow_pickle=True).item()
        outs.append(np.asarray(rec["output"]))
        elapsed.append(rec.get("elapsed", rec.get("elapsed time")))
        history.append(rec["history"])
        dev = rec.get("device", dev)
    # skipped (all-masked) patches are stored with a trailing singleton channel (main.py:283)
    outs = [o[..., 0] if (o.ndim == len(pe.dim) + 1 and o.shape[-1] == 1 and args.datadim != "2.5d") else o for o in outs]
    patches_out = np.asarray(outs)
    if args.datadim == "2.5d":
        patches_out = transpose_patches_25d(patches_out, args.slice, adj=True)
    outputs = pe.reconstruct(patches_out.reshape(pa_shape)) / args.gain
    if verbose:
        print("\n%d patches; total elapsed time on %s: %s"
              % (len(history), dev, u.sec2time(sum(u.time2sec(e) for e in elapsed if e))))
    return (outputs, history) if return_history else outputs
