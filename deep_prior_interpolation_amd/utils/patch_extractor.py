"""N-D sliding-window patch extraction and overlap-add reassembly
(drop-in for the part of reference utils/patch_extractor.py that data.py uses: rectangular taper,
C-order over the window grid, no scoring / shuffling / tapering options).

Window w = (w_0..w_{n-1}) starts at w_k * stride_k; the grid has (in_k - dim_k)//stride_k + 1 windows per
axis; reassembly averages overlapping samples (hit-count normalisation)."""
import numpy as np

__all__ = ["PatchExtractor", "count_patches", "patch_array_shape", "in_content_cropped_shape", "window_origins"]


def _grid(in_size, patch_size, patch_stride):
    return tuple((int(n) - int(d)) // int(s) + 1 for n, d, s in zip(in_size, patch_size, patch_stride))


def count_patches(in_size, patch_size, patch_stride):
    return int(np.prod(_grid(in_size, patch_size, patch_stride)))


def patch_array_shape(in_size, patch_size, patch_stride):
    return _grid(in_size, patch_size, patch_stride) + tuple(patch_size)


def in_content_cropped_shape(in_size, patch_size, patch_stride):
    assert len(in_size) == len(patch_size) == len(patch_stride)
    return tuple((g - 1) * s + d for g, s, d in zip(_grid(in_size, patch_size, patch_stride), patch_stride, patch_size))


def window_origins(in_size, patch_size, patch_stride):
    """Origins of all windows in C order of the window grid: array (num_patches, ndim)."""
    grid = _grid(in_size, patch_size, patch_stride)
    idx = np.stack(np.meshgrid(*[np.arange(g) for g in grid], indexing="ij"), axis=-1).reshape(-1, len(grid))
    return idx * np.asarray(patch_stride)[None, :]


class PatchExtractor:
    def __init__(self, dim, offset=None, stride=None, tapering="rect", padding=None, **unsupported):
        if not isinstance(dim, tuple):
            raise ValueError("dim must be a tuple")
        for k, v in unsupported.items():
            if v is not None:
                raise NotImplementedError("PatchExtractor option %s is outside the hot-path scope" % k)
        if tapering != "rect" or padding is not None:
            raise NotImplementedError("only rectangular tapering without padding is supported")
        self.dim = dim
        self.ndim = len(dim)
        self.offset = tuple([0] * self.ndim) if offset is None else offset
        self.stride = dim if stride is None else stride
        if not isinstance(self.stride, tuple) or len(self.stride) != self.ndim:
            raise ValueError("stride must a tuple of length {:d}".format(self.ndim))
        if not isinstance(self.offset, tuple) or len(self.offset) != self.ndim:
            raise ValueError("offset must a tuple of length {:d}".format(self.ndim))
        self.tapering = "rect"
        self.in_content_original_shape = None
        self.in_content_cropped_shape = None
        self.patch_array_shape = None

    def extract(self, in_content):
        if not isinstance(in_content, np.ndarray):
            raise ValueError("in_content must be of type: " + str(np.ndarray))
        if in_content.ndim != self.ndim:
            raise ValueError("in_content shape must a tuple of length {:d}".format(self.ndim))
        self.in_content_original_shape = in_content.shape
        in_content = in_content[tuple(slice(o, None) for o in self.offset)]
        view = np.lib.stride_tricks.sliding_window_view(in_content, self.dim)
        view = view[tuple(slice(None, None, s) for s in self.stride)]
        patch_array = np.ascontiguousarray(view)
        self.in_content_cropped_shape = tuple((g - 1) * s + d for g, s, d in
                                              zip(patch_array.shape[:self.ndim], self.stride, self.dim))
        self.patch_array_shape = patch_array.shape
        return patch_array

    def reconstruct(self, patch_array):
        if not isinstance(patch_array, np.ndarray):
            raise ValueError("patch_array must be of type: " + str(np.ndarray))
        ndim = patch_array.ndim // 2
        grid = patch_array.shape[:ndim]
        image_shape = tuple((g - 1) * s + d for g, s, d in zip(grid, self.stride, self.dim))
        if self.in_content_cropped_shape is not None and image_shape != tuple(self.in_content_cropped_shape):
            raise ValueError("There is something wrong with the dimensions!")
        recon = np.zeros(image_shape)
        hits = np.zeros(image_shape)
        for idx in np.ndindex(*grid):
            sl = tuple(slice(i * s, i * s + d) for i, s, d in zip(idx, self.stride, self.dim))
            recon[sl] += patch_array[idx]
            hits[sl] += 1
        recon /= hits
        return recon.astype(patch_array.dtype)
