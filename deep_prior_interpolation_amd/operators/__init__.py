"""Linear operators with forward / adjoint (drop-in for the reference's `operators` package: base.py, derivative.py,
signal.py).  Every operator runs HIP kernels of libdpi_hip.so; `forward` is differentiable (its backward is the adjoint
kernel), so an operator can sit inside a loss term of the deep-prior loop (the anti-aliasing add-on, BASELINE configs[3])."""
from .base import *        # noqa: F401,F403
from .derivative import *  # noqa: F401,F403
from .signal import *      # noqa: F401,F403
