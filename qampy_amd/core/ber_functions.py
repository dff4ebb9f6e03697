"""
Sequence alignment helpers of the pilot receiver / SER harness, behaviour of ``qampy.core.ber_functions``
(qampy/core/ber_functions.py:33-106): delay of one sequence inside another from the peak of their full cross-correlation,
trying the four quarter-turn rotations for complex data.  Host-side numpy/scipy (a few thousand samples per call).  :func:`cal_ser_dev` is the device-resident counterpart of
``SignalQAM.cal_ser`` (qampy/core/signals.py:295-335) for captures that stay in HBM.
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
    # turning y by 1j**i turns the whole cross-correlation by (-1j)**i: one correlation serves the four hypotheses, and its
    # magnitude peak - the offset - is the same for all of them
    idx, cc = find_sequence_offset(x, y, show_cc=True)
    peaks = np.array([cc.real.max(), cc.imag.max(), (-cc.real).max(), (-cc.imag).max()])      # max Re((-1j)**i cc), i = 0..3
    if not peaks.max() > 0:
        return 0, y, 0, 0.
    turns = int(np.argmax(peaks))
    return idx, y * 1.j ** turns, turns, float(peaks[turns])


def tx_indices_dev(symbols_tx, alphabet):
    """Decided indices of the transmitted symbols ``(nmodes, Nsym)`` as an int32 DeviceArray (once per capture)."""
    from .. import _lib
    from .equalisation import hip_equalisation as hk
    tx = symbols_tx if isinstance(symbols_tx, _lib.DeviceArray) else _lib.DeviceArray.from_host(np.ascontiguousarray(symbols_tx))
    alpha = alphabet if isinstance(alphabet, _lib.DeviceArray) else _lib.DeviceArray.from_host(
        np.ascontiguousarray(alphabet, dtype=tx.dtype))
    idx = _lib.DeviceArray(tx.shape, np.int32)
    hk.make_decision_dev(tx, alpha, None, None, idx)
    return idx


def cal_ser_dev(out, idx_tx, alphabet, maxlag=256, window=4096, trim=0):
    """
    Symbol error rate of recovered rows that live in HBM, without moving them to the host.

    What ``cal_ser`` does on the host - synchronise with the transmitted sequence over the four quarter-turn rotations
    (``sync_and_adjust`` / ``find_sequence_offset_complex``, ber_functions.py:33-160), decide, compare - as a bounded-lag
    search on decided indices plus one counting pass (``qh_ser_*_dev``).

    Parameters
    ----------
    out : DeviceArray (nrows, N) complex
    idx_tx : DeviceArray (nmodes, Nsym) int32 from :func:`tx_indices_dev`
    alphabet : DeviceArray (M,) complex, same dtype as ``out``
    maxlag : largest |lag| (symbols) searched;  window : symbols used for the search;  trim : symbols ignored at both ends

    Returns
    -------
    list of dicts per row: ``errors, compared, ser, tx_mode, rotation, lag, window_matches, window``
    """
    from .. import _lib
    suf = "c64" if np.dtype(out.dtype) == np.complex64 else "c128"
    nrows, N = out.shape
    nmodes, ntx = idx_tx.shape
    res = []
    for r in range(nrows):
        h = np.zeros(7, np.int64)
        _lib.call("qh_ser_%s_dev" % suf, out.row(r).ptr, N, idx_tx.ptr, nmodes, ntx, alphabet.ptr, int(np.prod(alphabet.shape)),
                  int(maxlag), int(window), int(trim), _lib.ptr(h))
        res.append(dict(errors=int(h[0]), compared=int(h[1]), ser=float(h[0]) / max(int(h[1]), 1), tx_mode=int(h[2]),
                        rotation=int(h[3]), lag=int(h[4]), window_matches=int(h[5]), window=int(h[6])))
    return res
