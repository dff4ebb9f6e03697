"""
Pilot-based receiver on plain arrays: frame synchronisation, pilot-sequence equalisation, pilot-aided carrier-phase and
frequency-offset estimation.  Same call surface and results as ``qampy.core.pilotbased_receiver``
(frame_sync :329-434, equalize_pilot_sequence :454-554, pilot_based_cpe_new :258-327, pilot_based_foe :32-73,
correct_shifts :436-443, shift_signal :445-452); these functions are *callers* of the equaliser hot path.

How the work is organised here (MI355X):
  * the blind search of ``frame_sync`` is ONE launch over all candidate windows (every window an independent chain); the
    error traces stay in HBM and only the variance table, the winning window per mode and its taps come back
    (``qh_train_equaliser_windows_search_*``) - the reference loops over ``equalise_signal`` calls and keeps every trace;
  * ``equalize_pilot_sequence`` describes its training as a list of jobs (one joint job, or one per mode when the modes
    start at different offsets); a job uploads its slice of the capture once and runs all of its stages on the resident
    copy (:class:`qampy_amd.core.equalisation.equalisation._Field`);
  * the small post-processing (a few thousand symbols: spectral peak, correlation against the pilot sequence, phase fit)
    is host numpy.
"""
import warnings
from collections import namedtuple

import numpy as np

from . import ber_functions, phaserecovery
from .filter import moving_average
from .equalisation import equalisation as _eq

#: below this correlation peak the synchronisation is reported as failed (the reference's threshold, :354)
FRAME_SYNC_THRS = 120

_Grid = namedtuple("_Grid", "window hop nsteps first")


def _search_grid(seq_len, frame_len, os):
    """Candidate windows of the blind search: one pilot sequence long, hopping by half of that over one frame (plus one hop);
    the first two hops are left out so that a neighbourhood of +-window exists around every candidate."""
    window = seq_len * os
    hop = window // 2
    return _Grid(window, hop, (frame_len * os) // hop + 1, 2)


def _blind_search(rx, grid, os, mu, M_pilot, Ntaps, eqargs):
    """Train a blind equaliser on every candidate window; per mode the step with the smallest error variance and the taps
    that window ended with."""
    steps = np.arange(grid.first, grid.nsteps)
    var, best, taps = _eq.search_windows(rx, os, mu, M_pilot, steps * grid.hop, grid.window, Ntaps=Ntaps, **eqargs)
    # The winner is the DEVICE's choice - the window whose taps came back (first minimum with a strict `<`: a NaN variance, i.e. a
    # diverged adaptive-step window, never wins there, whereas np.argmin would pick it and the centre would belong to one window,
    # the taps to another).  Steps that are never visited can never win: the placeholder of the reference's table is 1e2.
    best = np.asarray(best, dtype=np.int64)
    winner = steps[best]
    for m in range(rx.shape[0]):                              # degenerate capture: nothing beat the placeholder
        v = var[m, best[m]]
        if not (v < 1e2):
            winner[m] = 0
            taps[m] = 0
    return winner, taps


def _pilot_correlation(row, refs, candidates):
    """Delay and correlation peak of the received row against every still unassigned reference sequence."""
    delay = np.zeros(refs.shape[0], dtype=np.int32)
    peak = np.zeros(refs.shape[0], dtype=np.float64)
    for r in candidates:
        lag, _, _, height = ber_functions.find_sequence_offset_complex(refs[r], row)
        delay[r], peak[r] = -lag, height
    return delay, peak


def frame_sync(rx_signal, ref_symbs, os, frame_len=2 ** 16, M_pilot=4, mu=1e-3, Ntaps=17, **eqargs):
    """
    Locate the pilot sequence inside the frame, per mode.  Returns ``(shift_factor, foe_coarse, mode_sync_order, wx,
    sync_ok)``: sample index of the frame start for every signal mode, the coarse frequency offset, which reference
    sequence each signal mode carries, the taps of the last mode's best window and whether every correlation peak cleared
    the threshold.
    """
    rx = np.atleast_2d(rx_signal)
    refs = np.atleast_2d(ref_symbs)
    nmodes, seq_len = rx.shape[0], refs.shape[-1]
    if rx.shape[-1] < (frame_len + 2 * seq_len) * os:
        raise AssertionError("Signal must be at least as long as frame")
    method = eqargs.get("method")
    if method in _eq.REAL_VALUED and np.iscomplexobj(rx):
        raise ValueError("Equaliser method is {}, but using a real-valued equaliser in frame sync is unsupported".format(method))
    if method in _eq.DATA_AIDED and method not in _eq.REAL_VALUED:
        raise ValueError("Equaliser method is {}, but using a data-aided equaliser in frame sync is unsupported".format(method))
    grid = _search_grid(seq_len, frame_len, os)
    winner, taps = _blind_search(rx, grid, os, mu, M_pilot, Ntaps, eqargs)

    shift = np.zeros(nmodes, dtype=int)
    carries = np.zeros(nmodes, dtype=int)
    unassigned = list(range(nmodes))
    locked, foe, w_last = True, None, None
    for m in range(nmodes):
        centre = winner[m] * grid.hop
        w_last = taps[m]
        # equalise one window either side of the winner, take out the coarse offset, look for the pilots
        around = _eq.apply_filter(rx[:, centre - grid.window:centre + grid.window], os, w_last)
        foe = phaserecovery.find_freq_offset(around)
        around = phaserecovery.comp_freq_offset(around, foe)
        delay, peak = _pilot_correlation(around[m], refs, unassigned)
        hit = int(np.argmax(peak))
        if peak[hit] < FRAME_SYNC_THRS:
            warnings.warn("Very low autocorrelation, likely the frame-sync failed")
            locked = False
        carries[m] = hit
        if hit in unassigned:
            unassigned.remove(hit)
        shift[m] = centre + os * delay[hit] - grid.window
    return shift, foe, carries, w_last, locked


def correct_shifts(shift_factors, ntaps, os):
    """Frame starts seen through an equaliser of ``ntaps[1]`` instead of ``ntaps[0]`` taps: half the difference earlier."""
    grow = ntaps[1] - ntaps[0]
    if grow % os:
        raise ValueError("Taps for search and convergence impropper configured")
    return np.asarray(shift_factors) - int(grow / 2)


def shift_signal(sig, shift_factors):
    """Bring every mode's frame start to sample 0 (in place for several modes; a single factor rolls the whole array forward,
    like the reference)."""
    if len(shift_factors) == 1:
        return np.roll(sig, shift_factors, axis=-1)
    for m, s in enumerate(shift_factors):
        sig[m] = np.roll(sig[m], -s)
    return sig


def pilot_based_foe(rec_symbs, pilot_symbs):
    """Frequency offset (cycles per symbol) from the slope of the unwrapped phase of received against transmitted pilots:
    ``(mean over modes, per mode (n, 1), intercepts (n, 1))``."""
    rec, ref = np.atleast_2d(rec_symbs), np.atleast_2d(pilot_symbs)
    phase = np.unwrap(np.angle(ref.conj() * rec), axis=-1)
    n = np.arange(phase.shape[-1])
    fits = np.array([np.polyfit(n, row, 1) for row in phase])            # (nmodes, [slope, intercept])
    per_mode = fits[:, :1] / (2 * np.pi)
    return np.mean(per_mode), per_mode, fits[:, 1:].copy()


def pilot_based_cpe_new(signal, pilot_symbs, pilot_idx, frame_len, seq_len=None, num_average=1, use_pilot_ratio=1,
                        max_num_blocks=None, nframes=1):
    """
    Carrier phase from the pilots spread over the frame(s): phase of every used pilot against its reference, unwrapped,
    smoothed by a moving average of ``num_average`` pilots, interpolated linearly to every symbol and taken out.
    Returns ``(compensated signal, phase trace)``, both ``nframes * frame_len`` long.
    """
    if not num_average > 1:
        raise AssertionError("need to take average over at least 3")
    if num_average % 2 == 0:
        num_average += 1
        warnings.warn("Number of averages should be odd, adding one average, num_average={}".format(num_average))
    rows, refs = np.atleast_2d(signal), np.atleast_2d(pilot_symbs)
    span = min(frame_len * nframes, rows.shape[-1])          # symbols that get a phase
    # pilots in use: every use_pilot_ratio-th of the first max_num_blocks of a frame, in every frame, as far as the signal goes
    in_frame = pilot_idx[:max_num_blocks:use_pilot_ratio]
    where = (np.arange(nframes)[:, None] * frame_len + in_frame[None, :]).ravel()
    where = where[where < span]
    sent = np.tile(refs[:, ::use_pilot_ratio], nframes)[:, :where.size]
    if sent.shape[-1] != where.size:
        raise AssertionError("Inproper pilot configuration, the number of received pilots differs from reference ones")
    if where.size < num_average:
        raise AssertionError("Inpropper pilot symbol configuration. Larger averaging block size than total number of pilot symbols")
    # phase of every pilot against its reference, unwrapped and averaged over num_average neighbours: the knots of the phase trace
    knot_phase = moving_average(np.unwrap(np.angle(rows[:, where] * sent.conjugate()), axis=-1), num_average)
    half = num_average // 2
    knots = where[half:where.size - half]
    if knots.size != knot_phase.shape[-1]:
        raise AssertionError("averaged phase and new indices are not the same shape")
    # the phase at every symbol (linear between the knots, constant outside: np.interp) and its removal: one pass over the frame(s) on the device
    keep = nframes * frame_len
    # precision as numpy's promotion gives it in the reference (signal * exp(-1j trace), :327): the wider of the signal and the pilots
    field = np.ascontiguousarray(rows[:, :span], dtype=np.result_type(rows.dtype, sent.dtype, np.complex64))
    corrected, trace = phaserecovery._dsp.pilot_phase_trace(field, knots, knot_phase)
    return corrected[:, :keep], trace[:, :keep]


def _equalize_pilot_jobs(rx, refs, starts, span, os, foe_comp, mu, M_pilot, Ntaps, Niter, adaptive, methods, wxinit):
    """``equalize_pilot_sequence`` when every mode has its own start sample: the per-mode trainings are independent chains
    (an output mode's taps only see their own error), so the slices are uploaded together and every stage trains all of
    them in ONE launch (:class:`hip_equalisation.ResidentJobs`) - the reference, and the joint path below, run them one after
    the other."""
    nmodes = rx.shape[0]
    m0, m1 = methods
    dtype = np.asarray(rx).dtype

    def bank(pieces):
        return _eq._kernels.ResidentJobs(pieces, range(nmodes))

    pieces = [rx[:, s:s + span] for s in starts]

    def stage(jobs, mu_s, M, taps, method, symbols):
        n = taps.shape[-1]
        TrSyms = _eq._cal_training_symbol_len(os, n, span)
        sy = _eq._reshape_symbols(symbols, method, M, dtype, nmodes).copy()
        return jobs.train(TrSyms, Niter, os, dtype.type(0).real.dtype.type(mu_s), taps, adaptive, sy, method)

    # given taps are pre-converged IN PLACE and copied afterwards, like the reference (and the joint path below) do
    taps = _eq._init_taps(Ntaps, nmodes, nmodes, dtype) if wxinit is None else np.ascontiguousarray(wxinit, dtype=dtype)
    jobs = bank(pieces)
    taps = stage(jobs, mu[0], M_pilot, taps, m0, None).copy()
    if foe_comp:
        foe, foe_modes, _ = pilot_based_foe(jobs.apply(os, taps), refs)
        offsets = np.ones(foe_modes.shape) * foe
        jobs = bank([phaserecovery.comp_freq_offset(p, offsets, os=os) for p in pieces])      # per slice, as the reference does (:527)
    else:
        offsets = np.zeros([nmodes, 1])
    taps = stage(jobs, mu[0], M_pilot, taps, m0, refs)
    taps = stage(jobs, mu[1], 4, taps, m1, refs)              # the reference hard-codes QPSK pilots here when it trains mode by mode (:540)
    return np.array(taps), offsets


def equalize_pilot_frames(rx_signal, ref_symbs, starts_per_frame, os, foe_comp=False, mu=(1e-4, 1e-4), M_pilot=4, Ntaps=45,
                          Niter=30, adaptive_stepsize=True, methods=('cma', 'cma'), wxinit=None):
    """
    :func:`equalize_pilot_sequence` for SEVERAL frames of one capture that are all handed the same array of start taps ``wxinit`` -
    what ``pilot_equaliser_nframes`` does from its second frame on (qampy/equalisation.py:386-390) - when the modes of a frame start
    at different samples.  What links the frames in the reference is its pre-convergence stage only: it trains the given taps IN
    PLACE (core/pilotbased_receiver.py:497-510: ``wx = wxinit`` goes through ``equalise_signal``, whose ``np.ascontiguousarray`` keeps
    the array and whose compiled trainer writes into it), so frame k+1 starts from the taps frame k's pre-convergence left behind;
    the two pilot-aided stages work on a copy (``out_taps = wx.copy()``) and depend on nothing but their own frame.  Hence:
      * the pre-convergence stages run frame after frame (the modes of a frame together: independent chains), on ``wxinit`` in place,
      * the 2 x Niter pilot-aided sweeps of ALL frames - two thirds of the work - run together: every (frame, mode) pair is an
        independent chain of ONE launch per stage (:class:`hip_equalisation.ResidentJobs`).
    Per frame the result is exactly :func:`equalize_pilot_sequence`'s, ``wxinit`` ends as the reference leaves it.
    Returns ``(taps (nframes, nmodes, nmodes, Ntaps), offsets (nframes, nmodes, 1))``; ``None`` when the frames cannot be batched
    (real-valued methods, modes that share a start sample, no start taps: the caller then goes frame by frame).
    """
    rx, refs = np.atleast_2d(rx_signal), np.atleast_2d(ref_symbs)
    nmodes, seq_len = rx.shape[0], refs.shape[-1]
    starts_per_frame = [np.asarray(s, dtype=int) for s in starts_per_frame]
    m0, m1 = _eq._method_name(methods[0]), _eq._method_name(methods[1])
    dtype = np.asarray(rx).dtype
    if wxinit is None or m0 in _eq.REAL_VALUED or m1 in _eq.REAL_VALUED or any(np.unique(s).shape[0] != nmodes for s in starts_per_frame) or nmodes < 2:
        return None
    if not (isinstance(wxinit, np.ndarray) and wxinit.dtype == dtype and wxinit.flags.c_contiguous):
        return None                                                   # (the in-place chain needs the caller's own array)
    span = seq_len * os + Ntaps - 1
    nfr = len(starts_per_frame)
    rt = dtype.type(0).real.dtype.type
    TrSyms = _eq._cal_training_symbol_len(os, Ntaps, span)
    sy0 = _eq._reshape_symbols(None, m0, M_pilot, dtype, nmodes).copy()
    pieces, pre = [], []
    offsets = np.zeros((nfr, nmodes, 1))
    for f, st in enumerate(starts_per_frame):                         # ---- pre-convergence: the chain through the frames
        mine = [rx[:, int(st[m]):int(st[m]) + span] for m in range(nmodes)]
        jobs = _eq._kernels.ResidentJobs(mine, range(nmodes))
        jobs.train(TrSyms, Niter, os, rt(mu[0]), wxinit, adaptive_stepsize, sy0, m0)          # in place, like the reference
        pre.append(wxinit.copy())
        if foe_comp:
            foe, foe_modes, _ = pilot_based_foe(jobs.apply(os, pre[f]), refs)
            offsets[f] = np.ones(foe_modes.shape) * foe
            mine = [phaserecovery.comp_freq_offset(p, offsets[f], os=os) for p in mine]
        pieces += mine
    job_modes = [m for _ in range(nfr) for m in range(nmodes)]       # ---- the pilot-aided stages of all frames together
    jobs = _eq._kernels.ResidentJobs(pieces, job_modes)
    bank = np.ascontiguousarray(np.repeat(np.array(pre), nmodes, axis=0))
    for mu_s, M, method in ((mu[0], M_pilot, m0), (mu[1], 4, m1)):   # (QPSK pilots hard-coded for the second method when training mode by mode, :540)
        sy = _eq._reshape_symbols(refs, method, M, dtype, nmodes).copy()
        bank = jobs.train_bank(TrSyms, Niter, os, rt(mu_s), bank, adaptive_stepsize, sy, method)
    taps = np.empty((nfr,) + wxinit.shape, dtype=dtype)
    for f in range(nfr):
        taps[f] = pre[f]
        for m in range(nmodes):
            taps[f, m] = bank[f * nmodes + m, m]
    return taps, offsets


def equalize_pilot_sequence(rx_signal, ref_symbs, shift_fctrs, os, foe_comp=False, mu=(1e-4, 1e-4), M_pilot=4, Ntaps=45,
                            Niter=30, adaptive_stepsize=True, methods=('cma', 'cma'), wxinit=None):
    """
    Train the equaliser on the pilot sequence: pre-convergence with ``methods[0]``, optionally the pilot-based frequency
    offset taken out, then ``methods[0]`` and ``methods[1]`` once more with the pilot sequence as the symbols argument.
    Modes whose sequences start at different samples are trained one after the other on their own slices.
    Returns ``(taps, frequency offset per mode (n, 1))``.
    """
    rx, refs = np.atleast_2d(rx_signal), np.atleast_2d(ref_symbs)
    nmodes, seq_len = rx.shape[0], refs.shape[-1]
    if (methods[0] in _eq.REAL_VALUED) != (methods[1] in _eq.REAL_VALUED):
        raise ValueError("Using a complex and real-valued equalisation method is not supported")
    span = seq_len * os + Ntaps - 1
    real = methods[0] in _eq.REAL_VALUED
    # jobs: (start sample, modes to train or None for all)
    split = np.unique(shift_fctrs).shape[0] > 1
    jobs = [(int(shift_fctrs[m]), [m]) for m in range(nmodes)] if split else [(int(shift_fctrs[0]), None)]
    m0, m1 = _eq._method_name(methods[0]), _eq._method_name(methods[1])

    if split and not real:
        return _equalize_pilot_jobs(rx, refs, [j[0] for j in jobs], span, os, foe_comp, mu, M_pilot, Ntaps, Niter, adaptive_stepsize, (m0, m1), wxinit)

    # ---- pre-convergence; the equalised pilot sequence is only needed for the frequency-offset estimate
    taps = wxinit
    seen = np.zeros_like(refs)
    for start, modes in jobs:
        field = _eq._Field(rx[:, start:start + span], real)
        rows = field.mode_rows(modes)
        taps, _ = field.train(os, mu[0], M_pilot, field.taps(taps, Ntaps), None, Niter, m0,
                              adaptive_stepsize, None, rows)
        if foe_comp or modes is None:
            eq = field.filtered(os, taps, rows)
            if modes is None:
                seen = eq
            else:
                seen[modes[0]] = eq
    if foe_comp:
        foe, foe_modes, _ = pilot_based_foe(seen, refs)
        offsets = np.ones(foe_modes.shape) * foe
    else:
        offsets = np.zeros([nmodes, 1])

    # ---- with the pilots: both methods on the (offset-free) slice, one upload per job
    taps = taps.copy()
    for start, modes in jobs:
        piece = rx[:, start:start + span]
        if foe_comp:
            piece = phaserecovery.comp_freq_offset(piece, offsets, os=os)
        field = _eq._Field(piece, real)
        rows = field.mode_rows(modes)
        taps, _ = field.train(os, mu[0], M_pilot, field.taps(taps, Ntaps), None, Niter, m0, adaptive_stepsize, refs, rows)
        # the reference hard-codes QPSK pilots for the second method when it trains mode by mode (:540)
        taps, _ = field.train(os, mu[1], 4 if split else M_pilot, field.taps(taps, Ntaps), None, Niter, m1, adaptive_stepsize, refs, rows)
    return np.array(taps), offsets
