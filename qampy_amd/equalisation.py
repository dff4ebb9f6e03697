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


class _FrameLayout:
    """Where the frames of a frame-synchronised pilot capture start for a filter of ``ntaps`` taps (the shifts found by
    ``sync2frame`` hold for its ``synctaps``-tap search filter; a longer filter reaches further back by half the difference)."""

    def __init__(self, signal, ntaps):
        if signal.shiftfctrs is None:
            raise ValueError("the signal has to be synced to the frame first (sync2frame)")
        self.ntaps, self.synctaps = int(ntaps), int(signal.synctaps)
        if abs(self.ntaps - self.synctaps) % 2:
            raise ValueError("Tap difference need to be an integer of the oversampling")
        self.shifts = np.array(signal.shiftfctrs, dtype=int)
        self.frame_samples = signal.os * signal.frame_len
        self.length = signal.shape[-1]

    def starts(self, frame):
        # like the reference (qampy/equalisation.py:318-321) the frame offset only enters together with the tap correction
        if self.ntaps == self.synctaps:
            return self.shifts
        return self.shifts - (self.ntaps - self.synctaps) // 2 + self.frame_samples * frame

    def holds_whole_frame(self, first_sample):
        return self.length - first_sample > self.frame_samples


def pilot_equaliser(signal, mu, Ntaps, apply=True, foe_comp=True, wxinit=None, frame=0, verbose=False, **eqkwargs):
    """
    Pilot-based equalisation of one frame of a synced pilot signal (qampy/equalisation.py:268-338): data-aided training on
    the pilot sequence (:func:`qampy_amd.core.pilotbased_receiver.equalize_pilot_sequence`), optional frequency-offset
    compensation, then the filter over the frame.  Returns ``taps`` (``apply=False``), ``(taps, sig_out)`` or, with
    ``verbose``, additionally the frequency offsets and ``(Ntaps, synctaps)``.
    """
    layout = _FrameLayout(signal, Ntaps if wxinit is None else wxinit.shape[-1])
    starts = layout.starts(frame)
    if not layout.holds_whole_frame(starts.max()):
        raise ValueError("You are trying to equalise an incomplete frame which does not work")
    steps = np.atleast_1d(mu)
    steps = np.repeat(steps, 2) if len(steps) == 1 else steps
    taps, foe = pilotbased_receiver.equalize_pilot_sequence(np.asarray(signal), np.asarray(signal.pilot_seq), starts, os=signal.os, mu=steps,
                                                            foe_comp=foe_comp, Ntaps=layout.ntaps, wxinit=wxinit, **eqkwargs)
    result = [taps]
    if apply:
        source = phaserec.comp_freq_offset(signal, foe) if foe_comp else signal
        result.append(apply_filter(source, taps, frames=[frame]))
    if verbose:
        result += [foe, (layout.ntaps, layout.synctaps)]
    return result[0] if len(result) == 1 else tuple(result)


def _pilot_equaliser_frames(signal, mu, layout, frames, seed, apply, foe_comp, verbose, eqkwargs):
    """The frames ``frames`` of a synced pilot signal, all from the taps ``seed``, trained together; {frame: what pilot_equaliser returns}
    or None when they cannot be batched (then the caller goes frame by frame)."""
    lay = _FrameLayout(signal, seed.shape[-1])
    starts = [lay.starts(f) for f in frames]
    if any(not lay.holds_whole_frame(st.max()) for st in starts):
        return None
    steps = np.atleast_1d(mu)
    steps = np.repeat(steps, 2) if len(steps) == 1 else steps
    kw = dict(eqkwargs)
    res = pilotbased_receiver.equalize_pilot_frames(np.asarray(signal), np.asarray(signal.pilot_seq), starts, os=signal.os, mu=steps, foe_comp=foe_comp,
                                                    Ntaps=lay.ntaps, wxinit=seed, **kw)
    if res is None:
        return None
    taps, foes = res
    out = {}
    for k, f in enumerate(frames):
        result = [taps[k]]
        if apply:
            source = phaserec.comp_freq_offset(signal, foes[k]) if foe_comp else signal
            result.append(apply_filter(source, taps[k], frames=[f]))
        if verbose:
            result += [foes[k], (lay.ntaps, lay.synctaps)]
        out[f] = tuple(result)
    return out


def pilot_equaliser_nframes(signal, mu, Ntaps, apply=True, foe_comp=True, frames=[0], wxinit=None, verbose=True, **eqkwargs):
    """Pilot-based equalisation frame by frame; the taps of frame 0 initialise the later frames (qampy/equalisation.py:340-397)."""
    layout = _FrameLayout(signal, Ntaps if wxinit is None else wxinit.shape[-1])
    last_shift = int(np.max(layout.shifts))
    if frames is None:
        frames = np.arange((layout.length - last_shift) // layout.frame_samples)
    frames = np.atleast_1d(frames)
    if not layout.holds_whole_frame(last_shift + int(np.max(frames)) * layout.frame_samples):
        raise ValueError("The last frame must be complete for equalisation")
    seed, per_frame = wxinit, []
    batched = {}
    for k, f in enumerate(frames):
        if f in batched:
            got = batched[f]
        else:
            got = pilot_equaliser(signal, mu, layout.ntaps, apply=apply, foe_comp=foe_comp, wxinit=seed, verbose=verbose, frame=f, **eqkwargs)
            got = got if isinstance(got, tuple) else (got,)
        if f == 0:
            seed = got[0]
        per_frame.append(got)
        # every frame after this one starts from the same taps (`seed` only changes at frame 0) and the frames do not depend on each
        # other: train them together - one launch per stage over all (frame, mode) chains (core.pilotbased_receiver.equalize_pilot_frames)
        rest = [int(g) for g in frames[k + 1:]]
        if not batched and seed is not None and len(rest) > 1 and 0 not in rest:
            batched = _pilot_equaliser_frames(signal, mu, layout, rest, seed, apply, foe_comp, verbose, eqkwargs) or {}
    columns = tuple(zip(*per_frame))                   # (taps of every frame, [equalised frames], [offsets, tap counts])
    if not apply:
        return columns
    joined = np.concatenate([np.asarray(x) for x in columns[1]], axis=-1)
    return columns[0], signal.recreate_from_np_array(joined, fs=signal.fb), columns[2:]
