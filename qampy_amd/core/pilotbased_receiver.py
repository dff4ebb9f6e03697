"""
Core pilot-based receiver on plain arrays, behaviour of ``qampy.core.pilotbased_receiver``
(qampy/core/pilotbased_receiver.py): frame synchronisation (:329-434), two-step pilot-sequence equalisation (:454-554),
pilot-aided carrier-phase estimation (:258-327) and frequency-offset fit (:32-73).  These functions are *callers* of the
equaliser hot path; the numerically heavy parts are the HIP kernels, everything else is host numpy.

MI355X-native difference: ``frame_sync`` trains all its ~frame_len*os/step search windows in ONE kernel launch
(``equalise_signal_windows``) - every window is an independent sequential chain, so the search that is a Python loop over
``equalise_signal`` calls in the reference fills hundreds of SIMDs at once.
"""
import warnings

import numpy as np

from . import ber_functions, phaserecovery
from .filter import moving_average
from .equalisation import equalisation

FRAME_SYNC_THRS = 120        # correlation-peak threshold of the reference (:354)


def pilot_based_foe(rec_symbs, pilot_symbs):
    """Frequency offset from a straight-line fit of the unwrapped pilot phase, per mode and averaged (:32-73)."""
    rec_symbs = np.atleast_2d(rec_symbs)
    pilot_symbs = np.atleast_2d(pilot_symbs)
    npols = rec_symbs.shape[0]
    cond = np.zeros([npols, 1])
    per_mode = np.zeros([npols, 1])
    for k in range(npols):
        ph = np.unwrap(np.angle(pilot_symbs[k, :].conj() * rec_symbs[k, :]))
        fit = np.polyfit(np.arange(0, len(ph)), ph, 1)
        per_mode[k, 0] = fit[0] / (2 * np.pi)
        cond[k, 0] = fit[1]
    return np.mean(per_mode), per_mode, cond


def pilot_based_cpe_new(signal, pilot_symbs, pilot_idx, frame_len, seq_len=None, num_average=1, use_pilot_ratio=1,
                        max_num_blocks=None, nframes=1):
    """
    Carrier-phase recovery from periodically inserted pilots: unwrapped pilot phase -> moving average -> linear
    interpolation over the frame -> de-rotation (:258-327).  Returns ``(compensated signal, phase trace)``.
    """
    assert num_average > 1, "need to take average over at least 3"
    if not (num_average % 2):
        num_average += 1
        warnings.warn("Number of averages should be odd, adding one average, num_average={}".format(num_average))
    signal = np.atleast_2d(signal)
    pilot_symbs = np.atleast_2d(pilot_symbs)
    sel = pilot_idx[:max_num_blocks:use_pilot_ratio]
    nlen = min(frame_len * nframes, signal.shape[-1])
    starts = np.arange(nframes) * frame_len
    idx_full = np.ravel(np.broadcast_to(sel, (nframes, sel.shape[-1])) + starts[:, None])
    idx_full = idx_full[idx_full < nlen]
    rec = signal[:, idx_full]
    ref = np.tile(pilot_symbs[:, ::use_pilot_ratio], nframes)[:, :rec.shape[-1]]
    assert rec.shape == ref.shape, "Inproper pilot configuration, the number of received pilots differs from reference ones"
    assert ref.shape[-1] >= num_average, "Larger averaging block size than total number of pilot symbols"
    res = np.unwrap(np.angle(ref.conjugate() * rec), axis=-1)
    avg = moving_average(res, num_average)
    half = int((num_average - 1) / 2)
    idx_avg = idx_full[half:-half]
    assert idx_avg.shape[-1] == avg.shape[-1], "averaged phase and new indices are not the same shape"
    trace = np.zeros((ref.shape[0], nlen), dtype=ref.dtype)
    grid = np.arange(0, nlen)
    for k in range(ref.shape[0]):
        trace[k] = np.interp(grid, idx_avg, avg[k])
    out = signal[:, :nlen] * np.exp(-1j * trace)
    return out[:, :nframes * frame_len], trace[:, :nframes * frame_len]


def frame_sync(rx_signal, ref_symbs, os, frame_len=2 ** 16, M_pilot=4, mu=1e-3, Ntaps=17, **eqargs):
    """
    Locate the pilot sequence inside the frame (:329-434): blind (CMA-type) equaliser runs on half-overlapping search
    windows, the window with the smallest error variance per mode is equalised, its coarse frequency offset removed and
    the pilot sequence found by cross-correlation.  Returns ``(shift_factor, foe_coarse, mode_sync_order, wx, sync_ok)``.
    """
    sync_ok = True
    rx_signal = np.atleast_2d(rx_signal)
    ref_symbs = np.atleast_2d(ref_symbs)
    seq_len = ref_symbs.shape[-1]
    nmodes = rx_signal.shape[0]
    assert rx_signal.shape[-1] >= (frame_len + 2 * seq_len) * os, "Signal must be at least as long as frame"
    if "method" in eqargs:
        if eqargs["method"] in equalisation.REAL_VALUED:
            if np.iscomplexobj(rx_signal):
                raise ValueError("Equaliser method is {}, but using a real-valued equaliser in frame sync is unsupported"
                                 .format(eqargs["method"]))
        elif eqargs["method"] in equalisation.DATA_AIDED:
            raise ValueError("Equaliser method is {}, but using a data-aided equaliser in frame sync is unsupported"
                             .format(eqargs["method"]))
    order = np.zeros(nmodes, dtype=int)
    open_modes = np.arange(0, nmodes)
    overlap = 2
    window = seq_len * os
    step = window // overlap
    num_steps = (frame_len * os) // step + 1
    shift = np.zeros(nmodes, dtype=int)
    var = np.ones((nmodes, num_steps)) * 1e2
    wxys = np.zeros((num_steps, nmodes, nmodes, Ntaps), dtype=rx_signal.dtype)
    # all search windows in one launch (the first `overlap` steps are skipped like in the reference)
    steps = np.arange(overlap, num_steps)
    w_all, e_all = equalisation.equalise_signal_windows(rx_signal, os, mu, M_pilot, steps * step, window, Ntaps=Ntaps, **eqargs)
    wxys[steps] = w_all
    var[:, steps] = np.var(e_all, axis=-1).T
    best = np.argmin(var, axis=-1)
    wxy = wxys[best]
    for k in range(nmodes):
        i_min = best[k]
        long_seq = rx_signal[:, i_min * step - window: i_min * step + window]
        wx1 = wxy[k]
        syms = equalisation.apply_filter(long_seq, os, wx1)
        foe = phaserecovery.find_freq_offset(syms)
        syms = phaserecovery.comp_freq_offset(syms, foe)
        peak = np.zeros(nmodes, dtype=np.float64)
        delay = np.zeros(nmodes, dtype=np.int32)
        for ref in open_modes:
            ix, _, _, ac = ber_functions.find_sequence_offset_complex(ref_symbs[ref], syms[k])
            delay[ref] = -ix
            peak[ref] = ac
        hit = np.argmax(peak)
        if peak[hit] < FRAME_SYNC_THRS:
            warnings.warn("Very low autocorrelation, likely the frame-sync failed")
            sync_ok = False
        order[k] = hit
        open_modes = open_modes[open_modes != hit]
        shift[k] = i_min * step + os * delay[hit] - window
    return shift, foe, order, wx1, sync_ok


def correct_shifts(shift_factors, ntaps, os):
    """Account for the different tap counts of the search and the convergence equaliser (:436-443)."""
    shift_factors = np.asarray(shift_factors)
    if not ((ntaps[1] - ntaps[0]) % os == 0):
        raise ValueError("Taps for search and convergence impropper configured")
    return shift_factors - int((ntaps[1] - ntaps[0]) / 2)


def shift_signal(sig, shift_factors):
    """Roll every mode to its frame start (:445-452)."""
    k = len(shift_factors)
    if k > 1:
        for i in range(k):
            sig[i] = np.roll(sig[i], -shift_factors[i])
    else:
        sig = np.roll(sig, shift_factors, axis=-1)
    return sig


def equalize_pilot_sequence(rx_signal, ref_symbs, shift_fctrs, os, foe_comp=False, mu=(1e-4, 1e-4), M_pilot=4, Ntaps=45,
                            Niter=30, adaptive_stepsize=True, methods=('cma', 'cma'), wxinit=None):
    """
    Train the equaliser on the pilot sequence in two steps (pre-convergence with ``methods[0]``, optional pilot-based
    frequency-offset removal, then ``methods[0]`` and ``methods[1]`` again with the pilots as reference symbols), per
    mode when the modes start at different offsets (:454-554).  Returns ``(taps, foe per mode)``.
    """
    rx_signal = np.atleast_2d(rx_signal)
    ref_symbs = np.atleast_2d(ref_symbs)
    npols = rx_signal.shape[0]
    seq_len = ref_symbs.shape[-1]
    wx = wxinit
    if methods[0] in equalisation.REAL_VALUED:
        if methods[1] not in equalisation.REAL_VALUED:
            raise ValueError("Using a complex and real-valued equalisation method is not supported")
    elif methods[1] in equalisation.REAL_VALUED:
        raise ValueError("Using a complex and real-valued equalisation method is not supported")
    span = seq_len * os + Ntaps - 1
    per_mode = np.unique(shift_fctrs).shape[0] > 1
    if per_mode:
        syms_out = np.zeros_like(ref_symbs)
        for i in range(npols):
            seg = rx_signal[:, shift_fctrs[i]: shift_fctrs[i] + span]
            syms_out[i], wx, _ = equalisation.equalise_signal(seg, os, mu[0], M_pilot, wxy=wx, Ntaps=Ntaps, Niter=Niter,
                                                               method=methods[0], adaptive_stepsize=adaptive_stepsize,
                                                               apply=True, modes=[i])
    else:
        seg = rx_signal[:, shift_fctrs[0]: shift_fctrs[0] + span]
        syms_out, wx, _ = equalisation.equalise_signal(seg, os, mu[0], M_pilot, wxy=wxinit, Ntaps=Ntaps, Niter=Niter,
                                                       method=methods[0], adaptive_stepsize=adaptive_stepsize, apply=True)
    if foe_comp:
        foe, foe_modes, _ = pilot_based_foe(syms_out, ref_symbs)
        foe_all = np.ones(foe_modes.shape) * foe
    else:
        foe_all = np.zeros([npols, 1])
    taps = wx.copy()
    if per_mode:
        for i in range(npols):
            seg = rx_signal[:, shift_fctrs[i]: shift_fctrs[i] + span]
            if foe_comp:
                seg = phaserecovery.comp_freq_offset(seg, np.ones(foe_modes.shape) * foe, os=os)
            taps, _ = equalisation.equalise_signal(seg, os, mu[0], M_pilot, wxy=taps, Ntaps=Ntaps, Niter=Niter, method=methods[0],
                                                   adaptive_stepsize=adaptive_stepsize, modes=[i], symbols=ref_symbs, apply=False)
            taps, _ = equalisation.equalise_signal(seg, os, mu[1], 4, wxy=taps, Ntaps=Ntaps, Niter=Niter, method=methods[1],
                                                   adaptive_stepsize=adaptive_stepsize, modes=[i], symbols=ref_symbs, apply=False)
    else:
        seg = rx_signal[:, shift_fctrs[0]: shift_fctrs[0] + span]
        if foe_comp:
            seg = phaserecovery.comp_freq_offset(seg, np.ones(foe_modes.shape) * foe, os=os)
        taps, _ = equalisation.equalise_signal(seg, os, mu[0], M_pilot, wxy=taps, Ntaps=Ntaps, Niter=Niter, method=methods[0],
                                               adaptive_stepsize=adaptive_stepsize, symbols=ref_symbs, apply=False)
        taps, _ = equalisation.equalise_signal(seg, os, mu[1], M_pilot, wxy=taps, Niter=Niter, method=methods[1],
                                               adaptive_stepsize=adaptive_stepsize, symbols=ref_symbs, apply=False)
    return np.array(taps), foe_all
