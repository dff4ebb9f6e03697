"""
Basic carrier-recovery API, mirror of ``qampy.phaserec.bps`` (qampy/phaserec.py:62-92) and ``bps_twostage`` (:24-60): the
alphabet is taken from the signal object and the signal subclass is preserved through the de-rotation.  Pilot receiver:
``find_freq_offset`` / ``comp_freq_offset`` (:94-136) and ``pilot_cpe`` (:156-192).
"""
import numpy as np

from . import core


def bps(E, Mtestangles, N, **kwargs):
    """Blind phase search on a signal object: ``(Eout, ph)``; see :func:`qampy_amd.core.phaserecovery.bps`."""
    return core.phaserecovery.bps(E, Mtestangles, E.coded_symbols, N, **kwargs)


def bps_twostage(E, Mtestangles, N, B=4, **kwargs):
    """Two-stage blind phase search on a signal object; see :func:`qampy_amd.core.phaserecovery.bps_twostage`."""
    return core.phaserecovery.bps_twostage(E, Mtestangles, E.coded_symbols, N, B=B, **kwargs)


def find_freq_offset(sig, average_over_modes=False, fft_size=4096):
    """Frequency offset from the spectrum of the signal raised to the 4th power (qampy/phaserec.py:94-116)."""
    return core.phaserecovery.find_freq_offset(sig, sig.os, average_over_modes=average_over_modes, fft_size=fft_size)


def comp_freq_offset(sig, freq_offset):
    """Remove a frequency offset (per mode, or one for all) from a signal object (qampy/phaserec.py:118-136)."""
    return sig.recreate_from_np_array(core.phaserecovery.comp_freq_offset(np.asarray(sig), freq_offset, sig.os))


def pilot_cpe(signal, N=3, pilot_rat=1, max_blocks=None, nframes=1, use_seq=False):
    """Pilot-based carrier-phase estimation on a frame-aligned 1 sample/symbol pilot signal: ``(signal_out, phase_trace)``
    (qampy/phaserec.py:156-192 -> core.pilotbased_receiver.pilot_based_cpe_new)."""
    positions = np.flatnonzero(signal._idx_pil)
    if use_seq:                                          # sequence + phase pilots, or the phase pilots behind the sequence only
        reference, where, seq = signal.pilots, positions, signal._pilot_seq_len
    else:
        reference, where, seq = signal.ph_pilots, positions[signal._pilot_seq_len:], None
    corrected, trace = core.pilotbased_receiver.pilot_based_cpe_new(np.asarray(signal), np.asarray(reference), where, signal.frame_len, seq_len=seq,
                                                                    num_average=N, use_pilot_ratio=pilot_rat, max_num_blocks=max_blocks, nframes=nframes)
    return signal.recreate_from_np_array(corrected), trace


def find_pilot_const_phase(rec_pilots, ref_pilots):
    """Constant phase offset per mode between received and transmitted pilots (qampy/phaserec.py:194-218)."""
    rec, ref = np.atleast_2d(rec_pilots), np.atleast_2d(ref_pilots)
    return np.array([[np.mean(np.unwrap(np.angle(ref[m].conj() * rec[m])))] for m in range(rec.shape[0])], dtype=np.float64)


def correct_pilot_const_phase(signal, phase_offsets):
    """Remove the constant phase offsets found by :func:`find_pilot_const_phase` (qampy/phaserec.py:220-238)."""
    if signal.shape[0] != np.size(phase_offsets):
        raise ValueError("Number of signal modes and phase offsets must be the same")
    return signal * np.exp(-1j * np.asarray(phase_offsets))
