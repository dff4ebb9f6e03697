"""
Host layer of blind-phase-search carrier recovery, mirror of ``qampy.core.phaserecovery.bps``
(qampy/core/phaserecovery.py:93-159): test-angle grid, per-mode index search on the GPU, ``np.unwrap`` of the interior
and de-rotation.  The index search and the angle gather are the HIP kernels of :mod:`.hip_dsp`.
"""
import numpy as np

from .hip_dsp import bps as _bps_idx_hip
from .hip_dsp import select_angles


def bps(E, Mtestangles, symbols, N, method="pyt", **kwargs):
    """
    Blind phase search (Pfau et al. 2009), same contract as phaserecovery.py:93-159.

    E : 1-d (one mode) or 2-d (modes x symbols) complex array at 1 sample/symbol
    Mtestangles : number of test angles on [-pi/4, pi/4)
    symbols : alphabet
    N : half width of the averaging window (2N symbols are averaged)
    method : kept for signature compatibility; "pyt" and "hip" both run the HIP kernel

    Returns ``(Eout, ph)``: de-rotated signal and the applied (unwrapped) phase.
    """
    if method.lower() not in ("pyt", "hip"):
        raise ValueError("Method needs to be 'pyt' or 'hip' (the py/pyx/af back-ends of the reference are not provided)")
    dtype = np.float32 if E.dtype is np.dtype(np.complex64) else np.float64
    angles = np.linspace(-np.pi / 4, np.pi / 4, Mtestangles, endpoint=False, dtype=dtype).reshape(1, -1)
    Ew = np.atleast_2d(E).astype(E.dtype)
    symbols = np.asarray(symbols).astype(E.dtype, copy=False)
    ph = []
    for i in range(Ew.shape[0]):
        idx = _bps_idx_hip(np.ascontiguousarray(np.asarray(Ew[i])), angles, symbols, N)
        ph.append(select_angles(np.copy(angles), idx.astype(int)))
    ph = np.asarray(ph, dtype=dtype)
    # only the interior is unwrapped; the first and last N symbols keep angles[0] (phaserecovery.py:155)
    ph[:, N:-N] = np.unwrap(ph[:, N:-N] * 4) / 4
    if E.ndim == 1:
        return (Ew * np.exp(1.j * ph)).flatten(), ph.flatten()
    return Ew * np.exp(1.j * ph), ph
