"""Data pre-processing helpers on the hot path's data side (drop-in subset of reference utils/processing.py)."""
import numpy as np

__all__ = ["bool2bin"]


def bool2bin(in_content, logic=True):
    """NaN-decimated copy -> binary mask: finite samples -> 1 (0 if not logic), NaN -> 0 (1)."""
    nan = np.isnan(in_content)
    return np.where(nan, 0.0 if logic else 1.0, 1.0 if logic else 0.0).astype(in_content.dtype)
