"""Small filters used by the pilot receiver (behaviour of qampy/core/filter.py:215-237)."""
import numpy as np


def moving_average(sig, N=3):
    """Length ``len - N + 1`` running mean along the last axis, formed from a cumulative sum like the reference."""
    s2 = np.atleast_2d(sig)
    acc = np.cumsum(np.insert(s2, 0, 0, axis=-1), dtype=sig.dtype, axis=-1)
    out = (acc[:, N:] - acc[:, :-N]) / N
    return out.flatten() if sig.ndim == 1 else out
