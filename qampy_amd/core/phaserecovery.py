"""
Host layer of blind-phase-search carrier recovery, mirror of ``qampy.core.phaserecovery.bps`` (qampy/core/phaserecovery.py:
93-159) and ``bps_twostage`` (:222-288): test-angle grid, per-mode index search on the GPU, ``np.unwrap`` and de-rotation.
The index search (one grid or a per-symbol grid) and the angle gather are the HIP kernels of :mod:`.hip_dsp`.
"""
import numpy as np

from . import hip_dsp as _dsp
from .hip_dsp import bps as _bps_idx_hip
from .hip_dsp import select_angles


def bps(E, Mtestangles, symbols, N, method="pyt", **kwargs):
    """
    Blind phase search (Pfau et al. 2009), contract of qampy/core/phaserecovery.py:93-159: ``(Eout, ph)`` - the de-rotated
    signal and the applied phase, unwrapped on ``[N, -N)`` (the first and last ``N`` symbols keep the first grid angle).

    E : 1-d (one mode) or 2-d (modes x symbols) complex array at 1 sample/symbol;  Mtestangles : angles on [-pi/4, pi/4);
    symbols : alphabet;  N : half width of the averaging window (2N symbols);  method : "pyt" / "hip" (same kernels).

    All modes go through one fused device pass (index search, grid look-up, unwrap, de-rotation: ``hip_dsp.bps_recover``).
    """
    if method.lower() not in ("pyt", "hip"):
        raise ValueError("Method needs to be 'pyt' or 'hip' (the py/pyx/af back-ends of the reference are not provided)")
    rows = np.atleast_2d(E)
    out, ph = _dsp.bps_recover(np.ascontiguousarray(rows), Mtestangles, np.asarray(symbols).astype(rows.dtype, copy=False), N)
    if E.ndim == 1:
        return out.reshape(-1), ph.reshape(-1)
    if type(E) is not np.ndarray and hasattr(E, "recreate_from_np_array"):        # keep the signal subclass like E * exp(..) does
        out = E.recreate_from_np_array(out)
    return out, ph


def bps_twostage(E, Mtestangles, symbols, N, B=4, method="pyt", **kwargs):
    """
    Two-stage blind phase search (Zhuge et al., OFC 2011), same contract as qampy/core/phaserecovery.py:222-288: a
    coarse search over ``Mtestangles`` angles, then ``B`` angles around each symbol's coarse estimate (a per-symbol
    ``(L, B)`` grid, the ``p == L`` branch of the kernel).  Returns ``(Eout, ph)``; the whole phase track is unwrapped.
    """
    if method.lower() not in ("pyt", "hip"):
        raise ValueError("Method needs to be 'pyt' or 'hip' (the py/pyx/af back-ends of the reference are not provided)")
    rdt = E.real.dtype
    rows = np.atleast_2d(E)
    alphabet = np.asarray(symbols).astype(E.dtype, copy=False)
    coarse = np.linspace(-np.pi / 4, np.pi / 4, Mtestangles, endpoint=False, dtype=rdt).reshape(1, -1)
    steps = np.linspace(-B / 2, B / 2, B)                  # offsets of the fine grid, in units of one coarse step / B

    def track(row):
        row = np.ascontiguousarray(np.asarray(row))
        first = select_angles(np.copy(coarse), _bps_idx_hip(row, coarse, alphabet, N))
        fine = (first[:, np.newaxis] + steps[np.newaxis, :] / (B * Mtestangles) * np.pi / 2).astype(rdt)       # (L, B): a grid per symbol
        second = select_angles(np.copy(fine), _bps_idx_hip(row, fine, alphabet, N))
        return np.unwrap(second * 4, discont=np.pi) / 4

    ph = np.asarray([track(r) for r in rows], dtype=rdt)
    out = rows * np.exp(1.j * ph)
    return (out.flatten(), ph.flatten()) if E.ndim == 1 else (out, ph)


def find_freq_offset(sig, os=1, average_over_modes=True, fft_size=2 ** 16):
    """
    Blind frequency-offset estimate from the spectral peak of the signal raised to the 4th power, in units of the symbol
    rate (qampy/core/phaserecovery.py:385-433).  Host-side FFT (the pilot receiver calls it on a few thousand symbols).
    """
    if not ((np.log2(fft_size) % 2 == 0) | (np.log2(fft_size) % 2 == 1)):
        fft_size = 2 ** (int(np.ceil(np.log2(fft_size))))
    sig = np.atleast_2d(sig)
    npols = sig.shape[0]
    fvec = np.fft.fftfreq(fft_size, 1 / os) / 4
    off = np.zeros([npols, 1])
    for k in range(npols):
        spec = np.abs(np.fft.fft(sig[k, :] ** 4, fft_size)) ** 2
        off[k, 0] = fvec[np.argmax(np.abs(spec))]
    if average_over_modes:
        off = np.mean(off) * np.ones(off.shape)
    return off


def comp_freq_offset(sig, freq_offset, os=1):
    """Remove a frequency offset given in units of the symbol rate, per mode (contract of qampy/core/phaserecovery.py:435-473:
    ``sig[k] * exp(-2j pi t freq_offset[k] / os)`` with ``t = 1 .. L``); one elementwise device pass over all modes."""
    rows = np.atleast_2d(sig)
    out = _dsp.comp_freq_offset(np.ascontiguousarray(rows), np.asarray(freq_offset, dtype=np.float64).reshape(-1)[:rows.shape[0]], os)
    return out.reshape(-1) if np.ndim(sig) == 1 else out
