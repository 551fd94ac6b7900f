"""Data pre-processing helpers on the hot path's data side (drop-in subset of reference utils/processing.py)."""
import numpy as np

__all__ = ["bool2bin", "ConvolveKernel_1d", "LowPassButterworth", "butterworth_fir_taps"]


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
