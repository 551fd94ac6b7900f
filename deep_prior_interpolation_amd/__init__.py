"""deep_prior_interpolation_amd — MI355X-native deep-prior seismic interpolation engine.

Hot path (HIP, gfx950): csrc/ -> libdpi_hip.so (C ABI: include/dpi_hip.h) <- _lib (ctypes) <- ops / engine.
Drop-in host surface: parameter, data, architectures.get_net, utils, main.Interpolator.
"""
__version__ = "0.1.0"
