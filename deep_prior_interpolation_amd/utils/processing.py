"""Data pre-processing helpers on the hot path's data side (drop-in subset of reference utils/processing.py)."""
import numpy as np

__all__ = ["bool2bin", "ConvolveKernel_1d", "LowPassButterworth", "butterworth_fir_taps", "GaussianFilter", "first_derivative",
           "second_derivative", "gaussian_kernel"]


def bool2bin(in_content, logic=True):
    """NaN-decimated copy -> binary mask: finite samples -> 1 (0 if not logic), NaN -> 0 (1)."""
    nan = np.isnan(in_content)
    return np.where(nan, 0.0 if logic else 1.0, 1.0 if logic else 0.0).astype(in_content.dtype)


class ConvolveKernel_1d:
    """Filter a (1,C,T,...) tensor along its time axis with a 1-D FIR kernel (reference utils/processing.py:34-67:
    grouped conv_transposeNd, i.e. a true convolution centred on len(kernel)//2).  Runs dpi_fir_axis0 on the GPU."""

    def __init__(self, kernel, ndim=2, dtype=None):
        kernel = np.asarray(kernel, dtype=np.float64)
        assert kernel.ndim == 1
        self.taps = kernel
        self.pad = kernel.size // 2
        self.ndim = ndim

    def __call__(self, x):
        import torch
        from .. import _lib
        if not x.is_cuda:
            raise _lib.DpiError("ConvolveKernel_1d: tensor must live on the GPU (no CPU path)")
        x = x.contiguous().float()
        taps = torch.from_numpy(self.taps.astype(np.float32)).to(x.device)
        y = torch.empty_like(x)
        C_, T_ = x.shape[1], x.shape[2]
        S = x.numel() // (C_ * T_)
        _lib.check(_lib.load().dpi_fir_axis0(_lib.ptr(x), _lib.ptr(taps), int(self.taps.size), C_, T_, S, _lib.ptr(y), _lib.stream()),
                   "dpi_fir_axis0")
        return y

    forward = __call__


def butterworth_fir_taps(fc, fs, ntaps=101, order=2, nfft=1024):
    """FIR approximation of a Butterworth low-pass: butter -> freqz -> least-squares FIR (utils/processing.py:74-77)."""
    from scipy.signal import butter, firls, freqz
    b, a = butter(order, fc, fs=fs, btype="low", analog=False)
    w_iir, h_iir = freqz(b, a, worN=nfft, fs=fs)
    return firls(ntaps, w_iir, abs(h_iir), fs=fs)


class LowPassButterworth(ConvolveKernel_1d):
    def __init__(self, fc, ndim=2, fs=None, ntaps=101, order=2, nfft=1024, dtype=None):
        super().__init__(butterworth_fir_taps(fc, fs, ntaps, order, nfft), ndim=ndim)


def gaussian_kernel(M, std, sym=True):
    """exp(-n^2 / 2 std^2), n centred, un-normalised (utils/processing.py:88-98)."""
    assert M > 1
    odd = M % 2
    if not sym and not odd:
        M = M + 1
    n = np.arange(0, M) - (M - 1.0) / 2.0
    w = np.exp(-n ** 2 / (2 * std * std))
    if not sym and not odd:
        w = w[:-1]
    return w


class GaussianFilter:
    """Isotropic Gaussian blur of a (B,C,...) tensor, 'same' size with zero padding (utils/processing.py:112-136: a
    ConvTransposeNd whose weight is the outer product of the 1-D bell).  The kernel is separable, so it runs as one
    dpi_fir_axis0 pass per spatial axis; symmetric taps make it self-adjoint.  Like the reference's weight of shape
    (1,1,K,..) the filter acts per channel (the reference only works for channels = 1)."""

    def __init__(self, channels, kernel_size, ndim, std):
        if ndim not in (1, 2, 3):
            raise ValueError
        assert kernel_size % 2 == 1, "odd kernel size expected ('same' output)"
        self.taps = gaussian_kernel(kernel_size, std, sym=True).astype(np.float32)
        self.ndim, self.channels = ndim, channels
        self._dev = {}

    def __call__(self, x):
        import torch
        from .. import _lib
        if not x.is_cuda:
            raise _lib.DpiError("GaussianFilter: tensor must live on the GPU (no CPU path)")
        if x.ndim != self.ndim + 2:
            raise _lib.DpiError("GaussianFilter(ndim=%d) expects a %d-D tensor" % (self.ndim, self.ndim + 2))
        taps = self._dev.get(str(x.device))
        if taps is None:
            taps = torch.from_numpy(self.taps).to(x.device)
            self._dev[str(x.device)] = taps
        L = _lib.load()
        cur = x.contiguous().float()
        for ax in range(2, x.ndim):
            outer = int(np.prod(cur.shape[:ax]))
            n = int(cur.shape[ax])
            inner = cur.numel() // (outer * n)
            y = torch.empty_like(cur)
            _lib.check(L.dpi_fir_axis0(_lib.ptr(cur), _lib.ptr(taps), int(taps.numel()), outer, n, inner, _lib.ptr(y), _lib.stream()),
                       "dpi_fir_axis0")
            cur = y
        return cur

    forward = __call__


def _axis_diff(in_content, axis, stencil, spacing):
    from ..operators.derivative import AxisDerivative
    return AxisDerivative(axis, stencil, spacing)(in_content)


def first_derivative(in_content, spacing=1., axis=0, stencil="forward"):
    """First derivative with a first-order stencil along `axis` (utils/processing.py:139-162); differentiable."""
    return _axis_diff(in_content, axis, stencil, spacing)


def second_derivative(in_content, spacing=1., axis=0):
    """Second derivative, centred first-order stencil (utils/processing.py:165-181)."""
    return _axis_diff(in_content, axis, "second", spacing)
