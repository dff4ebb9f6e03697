"""
Basic (signal-object aware) equaliser API, mirror of ``qampy.equalisation`` for the hot path
(qampy/equalisation.py:89-119 apply_filter, :122-192 equalise_signal, :194-264 dual_mode_equalisation).

The wrappers pull ``os``, ``M`` and the alphabet off the signal object, call the core functions and re-wrap the 1
sample/symbol result with ``sig.recreate_from_np_array(out, fs=sig.fb)``.  Any object with the attributes listed in
:mod:`qampy_amd.signals` works, including QAMpy's own signal classes.  Pilot-signal plumbing (qampy/equalisation.py:42-87,
:268-397) is out of scope for this round.
"""
from . import core
from .core.equalisation import DATA_AIDED, DECISION_BASED, NONDECISION_BASED, REAL_VALUED, TRAINING_FCTS  # noqa: F401


def _alphabet_for(sig, methods):
    symbols = None
    for method in methods:
        if method in DATA_AIDED:
            symbols = sig.symbols
        else:
            symbols = getattr(sig, "coded_symbols", None)
    return symbols


def apply_filter(sig, wxy, method="pyt", frames=[0]):
    """Apply taps to a signal object; returns a signal object at the symbol rate (qampy/equalisation.py:89-119)."""
    if hasattr(sig, "pilots") and frames:
        raise NotImplementedError("pilot-frame equalisation is not part of the hot path (SURVEY.md §8f)")
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
