"""
Sequence alignment helpers of the pilot receiver / SER harness, behaviour of ``qampy.core.ber_functions``
(qampy/core/ber_functions.py:33-106): delay of one sequence inside another from the peak of their full cross-correlation,
trying the four quarter-turn rotations for complex data.  Host-side numpy/scipy (a few thousand samples per call).
"""
import numpy as np
from scipy.signal import fftconvolve


def find_sequence_offset(x, y, show_cc=False):
    """Index by which ``y`` has to be shifted to line up with ``x`` (peak of the full cross-correlation, :33-69)."""
    X = 1. * np.asarray(x)
    Y = 1. * np.asarray(y)
    rev = Y.conj()[::-1] if np.iscomplexobj(Y) else Y[::-1]
    cc = fftconvolve(X, rev, "full")
    idx = abs(cc).argmax() - (Y.shape[0] - 1)
    return (idx, cc) if show_cc else idx


def find_sequence_offset_complex(x, y):
    """``(offset, rotated y, quarter turns, peak)`` over the rotations ``1j**i`` of the received sequence ``y`` (:71-106)."""
    if not np.iscomplexobj(x) and not np.iscomplexobj(y):
        idx, cc = find_sequence_offset(x, y, show_cc=True)
        return idx, y, 0, cc
    best, best_i, best_idx = 0., 0, 0
    for i in range(4):
        idx, cc = find_sequence_offset(x, y * 1.j ** i, show_cc=True)
        peak = cc.real.max()
        if peak > best:
            best, best_i, best_idx = peak, i, idx
    return best_idx, y * 1.j ** best_i, best_i, best
