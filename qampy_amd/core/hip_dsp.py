"""
Drop-in for the BPS entry points of the compiled module ``qampy.core.pythran_dsp`` on MI355X.

``bps`` and ``select_angles`` keep the signatures of the ``#pythran export`` lines qampy/core/pythran_dsp.py:45-46 and
:133-136; the kernels live in qampy_amd/csrc/bps.hip.
"""
import numpy as np

from .. import _lib


def bps(E, testangles, symbols, N):
    """
    Blind phase search: int32 index of the best test angle for every symbol of ONE mode (pythran_dsp.py:45-85 with
    select_angle_index :26-42).  ``testangles`` is ``(1, A)`` (one grid) or ``(L, A)`` (per-symbol grid).
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    if E.ndim != 1 or not np.iscomplexobj(E):
        raise TypeError("bps works on a 1-d complex array")
    E = np.ascontiguousarray(E)
    if testangles.ndim != 2 or testangles.dtype != rt:
        raise TypeError("testangles must be a 2-d %s array" % np.dtype(rt).name)
    symbols = np.ascontiguousarray(symbols)
    if symbols.dtype != ct:
        raise TypeError("symbols must be %s" % np.dtype(ct).name)
    testangles = np.ascontiguousarray(testangles)
    p, A = testangles.shape
    L = E.shape[0]
    if not (p == 1 or p == L):
        raise ValueError("p must be either 1 or the length of the input signal")
    idx = np.zeros(L, dtype=np.int32)
    _lib.call("qh_bps_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), L, _lib.ptr(testangles), p, A, _lib.ptr(symbols),
              symbols.size, int(N), _lib.ptr(idx))
    return idx


def select_angles(angles, idx):
    """``angles[0, idx[i]]`` (one grid) or ``angles[i, idx[i]]`` (per-symbol grid), pythran_dsp.py:133-153."""
    suf, rt, ct = _lib.suffix(angles.dtype)
    angles = np.ascontiguousarray(angles)
    if angles.ndim != 2:
        raise TypeError("angles must be 2-d")
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    p, A = angles.shape
    L = idx.shape[0] if p <= 1 else p
    if idx.size and (idx.max() >= A or idx.min() < 0):
        raise ValueError("angle index out of range")
    out = np.zeros(L, dtype=rt)
    _lib.call("qh_select_angles_f" + suf, _lib.ptr(angles), p, A, _lib.ptr(idx), L, _lib.ptr(out))
    return out


def bps_recover_dev(E, Mtestangles, symbols, N, idx, ph, Eout):
    """
    Device-resident carrier recovery of all modes at once: BPS index, linspace grid look-up, unwrap of the interior
    and de-rotation (host layer qampy/core/phaserecovery.py:145-159) without leaving HBM.  All arguments except the
    integers are DeviceArrays: E, Eout (nmodes, L) complex; ph (nmodes, L) real; idx (nmodes, L) int32.
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    nm, L = E.shape
    _lib.call("qh_bps_recover_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nm, L, int(Mtestangles), symbols.ptr,
              int(np.prod(symbols.shape)), int(N), idx.ptr, ph.ptr, Eout.ptr)
