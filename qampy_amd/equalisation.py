"""
Basic (signal-object aware) equaliser API, mirror of ``qampy.equalisation`` for the hot path
(qampy/equalisation.py:89-119 apply_filter, :122-192 equalise_signal, :194-264 dual_mode_equalisation).

The wrappers pull ``os``, ``M`` and the alphabet off the signal object, call the core functions and re-wrap the 1
sample/symbol result with ``sig.recreate_from_np_array(out, fs=sig.fb)``.  Any object with the attributes listed in
:mod:`qampy_amd.signals` works, including QAMpy's own signal classes.  Pilot frames (qampy/equalisation.py:42-87,
:268-397): :func:`pilot_equaliser`, :func:`pilot_equaliser_nframes` and ``apply_filter(..., frames=...)``.
"""
import numpy as np

from . import core, phaserec
from .core import pilotbased_receiver
from .core.equalisation import DATA_AIDED, DECISION_BASED, NONDECISION_BASED, REAL_VALUED, TRAINING_FCTS  # noqa: F401


def _alphabet_for(sig, methods):
    symbols = None
    for method in methods:
        if method in DATA_AIDED:
            symbols = sig.symbols
        else:
            symbols = getattr(sig, "coded_symbols", None)
    return symbols


def _apply_to_pilotsignal(sig, wxy, frames):
    """
    Filter the requested frames of a synced pilot signal (behaviour of qampy/equalisation.py:42-87): every mode starts at its
    own shift factor (moved by half the difference between equaliser and sync taps), consecutive frames are filtered in one
    go, others frame by frame; returns a 1 sample/symbol pilot signal of ``len(frames)`` frames.
    """
    Ntaps = wxy.shape[-1]
    E = np.asarray(sig)
    flen = sig.os * sig.frame_len
    shifts = np.array(sig.shiftfctrs, dtype=int)
    if Ntaps != sig.synctaps:
        shifts = shifts - (Ntaps - sig.synctaps) // 2
    if np.min(shifts) < 0:
        shifts += flen
    frames = list(frames)
    if not shifts.max() + flen * (max(frames) + 1) < E.shape[-1] - (Ntaps - 1):
        raise ValueError("Trying to equalise frame {}, but signal is not long enough".format(max(frames)))
    runs = [(frames[0], frames[-1] - frames[0] + 1)] if np.all(np.diff(frames) == 1) else [(f, 1) for f in frames]
    mode_groups = np.arange(wxy.shape[0]).reshape(-1, E.shape[0]).T          # rows of real-valued taps travel together
    pieces = []
    for f0, n in runs:
        if np.unique(shifts).shape[0] > 1:
            rows = []
            for grp in mode_groups:
                i0 = shifts[grp[0]] + f0 * flen
                rows.append(core.equalisation.apply_filter(E[:, i0:i0 + n * flen + Ntaps - 1], sig.os, wxy, modes=grp))
            pieces.append(np.squeeze(np.array(rows)))
        else:
            i0 = shifts[0] + f0 * flen
            pieces.append(core.equalisation.apply_filter(E[:, i0:i0 + n * flen + Ntaps - 1], sig.os, wxy))
    return sig.recreate_from_np_array(np.hstack(pieces), fs=sig.fb)


def apply_filter(sig, wxy, method="pyt", frames=[0]):
    """Apply taps to a signal object; returns a signal object at the symbol rate (qampy/equalisation.py:89-119)."""
    if hasattr(sig, "pilots") and frames:            # pilot signals: filter whole frames, starting at the synced positions
        return _apply_to_pilotsignal(sig, wxy, frames)
    out = core.equalisation.apply_filter(sig, sig.os, wxy, method=method)
    return sig.recreate_from_np_array(out, fs=sig.fb)


def equalise_signal(sig, mu, wxy=None, Ntaps=None, TrSyms=None, Niter=1, method="mcma", adaptive_stepsize=False,
                    symbols=None, modes=None, apply=False, **kwargs):
    """``(wxy, err)`` or ``(sig_out, wxy, err)``; see :func:`qampy_amd.core.equalisation.equalise_signal`."""
    if symbols is None:
        symbols = _alphabet_for(sig, (method,))
    res = core.equalisation.equalise_signal(sig, sig.os, mu, sig.M, wxy=wxy, Ntaps=Ntaps, TrSyms=TrSyms, Niter=Niter,
                                            method=method, adaptive_stepsize=adaptive_stepsize, symbols=symbols,
                                            modes=modes, apply=apply, **kwargs)
    if apply:
        out, wxy, err = res
        return sig.recreate_from_np_array(out, fs=sig.fb), wxy, err
    return res


def dual_mode_equalisation(sig, mu, Ntaps, TrSyms=(None, None), Niter=(1, 1), methods=("mcma", "sbd"),
                           adaptive_stepsize=(False, False), symbols=None, modes=None, apply=True, **kwargs):
    """``(sig_out, wxy, (err1, err2))`` or ``(wxy, (err1, err2))``; two-stage equaliser on a signal object."""
    if symbols is None:
        symbols = _alphabet_for(sig, methods)
    res = core.equalisation.dual_mode_equalisation(sig, sig.os, mu, sig.M, Ntaps=Ntaps, TrSyms=TrSyms, Niter=Niter,
                                                   methods=methods, adaptive_stepsize=adaptive_stepsize, symbols=symbols,
                                                   modes=modes, apply=apply, **kwargs)
    if apply:
        out, wxy, err = res
        return sig.recreate_from_np_array(out, fs=sig.fb), wxy, err
    return res


def pilot_equaliser(signal, mu, Ntaps, apply=True, foe_comp=True, wxinit=None, frame=0, verbose=False, **eqkwargs):
    """
    Pilot-based equalisation of one frame of a synced pilot signal (qampy/equalisation.py:268-338): data-aided training on
    the pilot sequence (:func:`qampy_amd.core.pilotbased_receiver.equalize_pilot_sequence`), optional frequency-offset
    compensation, then the filter over the frame.  Returns ``taps`` (``apply=False``), ``(taps, sig_out)`` or, with
    ``verbose``, additionally the frequency offsets and ``(Ntaps, synctaps)``.
    """
    if signal.shiftfctrs is None:
        raise ValueError("the signal has to be synced to the frame first (sync2frame)")
    shifts = np.array(signal.shiftfctrs, dtype=int)
    mu = np.atleast_1d(mu)
    if len(mu) == 1:
        mu = np.repeat(mu, 2)
    if wxinit is not None:
        Ntaps = wxinit.shape[-1]
    if abs(Ntaps - signal.synctaps) % 2 != 0:
        raise ValueError("Tap difference need to be an integer of the oversampling")
    elif Ntaps != signal.synctaps:
        shifts = shifts - (Ntaps - signal.synctaps) // 2 + signal.os * signal.frame_len * frame
    if not signal.shape[-1] - shifts.max() > signal.frame_len * signal.os:
        raise ValueError("You are trying to equalise an incomplete frame which does not work")
    taps, foe = pilotbased_receiver.equalize_pilot_sequence(np.asarray(signal), np.asarray(signal.pilot_seq), shifts, os=signal.os, mu=mu,
                                                            foe_comp=foe_comp, Ntaps=Ntaps, wxinit=wxinit, **eqkwargs)
    out_sig = phaserec.comp_freq_offset(signal, foe) if foe_comp else signal
    if not apply:
        return (taps, foe, (Ntaps, signal.synctaps)) if verbose else taps
    eq = apply_filter(out_sig, taps, frames=[frame])
    return (taps, eq, foe, (Ntaps, signal.synctaps)) if verbose else (taps, eq)


def pilot_equaliser_nframes(signal, mu, Ntaps, apply=True, foe_comp=True, frames=[0], wxinit=None, verbose=True, **eqkwargs):
    """Pilot-based equalisation frame by frame; the taps of frame 0 initialise the later frames (qampy/equalisation.py:340-397)."""
    if signal.shiftfctrs is None:
        raise ValueError("the signal has to be synced to the frame first (sync2frame)")
    if frames is None:
        frames = np.arange((signal.shape[-1] - np.max(signal.shiftfctrs)) // (signal.os * signal.frame_len))
    frames = np.atleast_1d(frames)
    if not signal.shape[-1] - (np.max(signal.shiftfctrs) + np.max(frames) * signal.frame_len * signal.os) > signal.frame_len * signal.os:
        raise ValueError("The last frame must be complete for equalisation")
    if wxinit is not None:
        Ntaps = wxinit.shape[-1]
    rets = []
    for i in frames:
        ret = pilot_equaliser(signal, mu, Ntaps, apply=apply, foe_comp=foe_comp, wxinit=wxinit, verbose=verbose, frame=i, **eqkwargs)
        if i == 0:
            wxinit = ret[0] if isinstance(ret, tuple) else ret
        rets.append(ret if isinstance(ret, tuple) else (ret,))
    out = tuple(zip(*rets))
    if apply:
        sout = signal.recreate_from_np_array(np.array(np.hstack([np.asarray(o) for o in out[1]])), fs=signal.fb)
        return out[0], sout, out[2:]
    return out
