"""
Host layer of the adaptive equaliser: argument normalisation, constants, tap initialisation and result assembly around
the HIP kernels.  Mirrors the *behaviour* of qampy/core/equalisation/equalisation.py for the hot-path functions

    equalise_signal         (:468-566)      dual_mode_equalisation (:400-466)      apply_filter (:138-188)
    generate_symbols_for_eq (:101-136)      _reshape_symbols       (:568-594)      _init_taps   (:364-367)
    _cal_training_symbol_len (:361-362)     real<->complex packing (:253-260)

with the same signatures, return values and error behaviour, so that reference code calling ``qampy.core.equalisation``
can switch to ``qampy_amd.core.equalisation`` unchanged.  Everything numerically heavy is delegated to
:mod:`.hip_equalisation` (the drop-in for the pythran extension).  Methods are selected by the same strings.
"""
import numpy as np

from ... import theory
from . import hip_equalisation as _kernels

#: decision based methods (equalisation.py:86)
DECISION_BASED = ("sbd", "mddma", "dd", "sbd_data", "dd_real", "dd_data_real")
#: non-decision based methods (:89)
NONDECISION_BASED = ("cma", "cma2", "mcma", "rde", "mrde", "cma_real", "sgncma_real", "sgncma")
#: methods that run on the real-stacked signal (:92)
REAL_VALUED = ("cma_real", "dd_real", "dd_data_real", "sgncma_real")
#: methods that need the transmitted symbols (:95)
DATA_AIDED = ("dd_data_real", "sbd_data")
#: every available method (:98)
TRAINING_FCTS = DECISION_BASED + NONDECISION_BASED


def generate_symbols_for_eq(method, M, dtype):
    """Per-method constant array handed to the trainer (equalisation.py:101-136): radius, codes+partitions or alphabet."""
    if method in ("cma", "cma2", "sgncma"):
        return np.atleast_2d(theory.cal_Rconstant(M) + 0j).astype(dtype)
    if method == "mcma":
        return np.atleast_2d(theory.cal_Rconstant_complex(M)).astype(dtype)
    if method == "rde":
        return np.atleast_2d(theory.generate_partition_codes_radius(M) + 0j).astype(dtype)
    if method == "mrde":
        return np.atleast_2d(theory.generate_partition_codes_complex(M)).astype(dtype)
    if method in ("sbd", "mddma", "dd"):
        return np.atleast_2d(theory.cal_symbols_qam(M) / np.sqrt(theory.cal_scaling_factor_qam(M))).astype(dtype)
    if method in ("sgncma_real", "cma_real"):
        return np.repeat([np.atleast_1d(theory.cal_Rconstant_complex(M).real.astype(dtype))], 2, axis=0)
    if method == "dd_real":
        s = theory.cal_symbols_qam(M) / np.sqrt(theory.cal_scaling_factor_qam(M))
        return np.vstack([s.real, s.imag]).astype(dtype)
    if method in DATA_AIDED:
        raise ValueError("%s is a data-aided method and needs the symbols to be passed" % method)
    raise ValueError("%s is unknown method" % method)


def _reshape_symbols(symbols, method, M, dtype, nmodes):
    """Bring ``symbols`` into the ``(nmodes, K)`` layout of the trainer (equalisation.py:568-594)."""
    if symbols is None or method in NONDECISION_BASED:      # caller-supplied arrays are ignored for blind methods (:569)
        symbols = generate_symbols_for_eq(method, M, dtype)
    symbols = np.asarray(symbols)
    if method not in REAL_VALUED:
        if symbols.ndim == 1 or symbols.shape[0] == 1:
            symbols = np.tile(symbols, (nmodes, 1))
        elif symbols.shape[0] != nmodes:
            raise ValueError("Symbols array is shape {} but signal has {} modes, symbols must be 1d or of shape (1, N) "
                             "or ({}, N)".format(symbols.shape, nmodes, nmodes))
        return np.atleast_2d(symbols.astype(dtype))
    half = nmodes // 2
    if np.iscomplexobj(symbols):
        if symbols.ndim == 1 or symbols.shape[0] == 1:
            symbols = np.repeat([symbols.real, symbols.imag], half, axis=0).squeeze().reshape(nmodes, -1)
        elif symbols.shape[0] == half:
            symbols = np.vstack([symbols.real, symbols.imag])
        else:
            raise ValueError("Symbols array is  complex and has {} modes, but needs to either have one mode or the same "
                             "modes as the signal ({})".format(symbols.shape[0], half))
    else:
        if symbols.shape[0] == 2 and nmodes > 2:
            symbols = np.repeat([symbols[0], symbols[1]], half, axis=0).squeeze().reshape(nmodes, -1)
        elif symbols.shape[0] != nmodes:
            raise ValueError("Symbols array is shape {} but signal has {} modes, symbols must be 1d or of shape (1, N) "
                             "or ({}, N)".format(symbols.shape, nmodes, nmodes))
    return symbols.astype(dtype)


def _cal_training_symbol_len(os, ntaps, L):
    return int(L // os // ntaps - 1) * int(ntaps)


def _init_taps(Ntaps, nmodes, nmodes2, dtype):
    """Centre-spike initial taps ``[0 .. 0 1 0 .. 0]`` on the diagonal (equalisation.py:364-367)."""
    wxy = np.zeros((nmodes, nmodes2, Ntaps), dtype=dtype)
    wxy[np.arange(nmodes), np.arange(nmodes), Ntaps // 2] = 1
    return wxy


def _convert_sig_to_real(E):
    """Stack ``[Re(mode 0..n-1); Im(mode 0..n-1)]`` (equalisation.py:253-257)."""
    return np.ascontiguousarray(np.vstack([E.real, E.imag]).astype(E.real.dtype))


def _convert_sig_to_cmplx(E, modes, Im=np.complex128(1j)):
    return E[:modes // 2, :] + Im * E[modes // 2:, :]


def apply_filter(E, os, wxy, method="pyt", modes=None):
    """
    Apply equaliser taps: ``(n_sel, (L-Ntaps+1)//os)`` equalised, decimated signal (equalisation.py:138-188).

    ``method`` is accepted for signature compatibility ("pyt" and "hip" both run the HIP kernel; "py" is the reference's
    broken NumPy branch and is not provided).
    """
    E = np.array(E, copy=True, order="C", subok=False)
    wxy = np.array(wxy, copy=True, order="C", subok=False)
    modes = np.arange(wxy.shape[0]) if modes is None else np.copy(np.atleast_1d(modes))
    nmodes = modes.shape[0]
    if method not in ("pyt", "hip"):
        raise NotImplementedError("Only the compiled (pyt/hip) method is implemented")
    if np.iscomplexobj(E) and np.iscomplexobj(wxy):
        return _kernels.apply_filter_to_signal(E, os, wxy, modes)
    if np.iscomplexobj(E):
        E = _convert_sig_to_real(E)
    out = _kernels.apply_filter_to_signal(E, os, wxy, modes)
    if E.itemsize == 8:
        return _convert_sig_to_cmplx(out, nmodes, np.complex128(1j))
    if E.itemsize == 4:
        return _convert_sig_to_cmplx(out, nmodes, np.complex64(1j))
    raise ValueError("The field has an unknown data type")


def equalise_signal(E, os, mu, M, wxy=None, Ntaps=None, TrSyms=None, Niter=1, method="mcma", adaptive_stepsize=False,
                    symbols=None, modes=None, apply=False, **kwargs):
    """
    Blind / decision-directed / data-aided equaliser training (equalisation.py:468-566).

    Returns ``(wxy, err)`` or, with ``apply=True``, ``(E_equalised, wxy, err)``.  Unknown keyword arguments are
    swallowed like in the reference.
    """
    method = method.lower()
    E = np.asarray(E)
    if method in REAL_VALUED:
        E = _convert_sig_to_real(E)
    else:
        E = np.array(E, copy=True, order="C", subok=False)
    mu = E.real.dtype.type(mu)
    nmodes = E.shape[0]
    if modes is None:
        modes = np.arange(nmodes)
    else:
        modes = np.atleast_1d(modes)
        if method in REAL_VALUED:
            modes = np.hstack([modes, modes + nmodes // 2])
        assert np.max(modes) < nmodes, "largest mode number is larger than shape of signal"
    if wxy is None:
        wxy = _init_taps(Ntaps, nmodes, nmodes, E.dtype)
    else:
        wxy = np.ascontiguousarray(wxy, dtype=E.dtype)      # no copy when possible: taps are updated in place (:547)
        Ntaps = wxy.shape[-1]
        assert wxy.ndim == 3, "wxy needs to be three dimensional"
        assert wxy.shape[:2] == (nmodes, nmodes), "The first 2 dimensions of wxy need to be the same shape as E"
    if TrSyms is None:
        TrSyms = _cal_training_symbol_len(os, Ntaps, E.shape[-1])
    symbols = _reshape_symbols(symbols, method, M, E.dtype, nmodes)
    if method in REAL_VALUED:
        err, wxy, mu = _kernels.train_equaliser_realvalued(E, TrSyms, Niter, os, mu, wxy, modes, adaptive_stepsize,
                                                            symbols.copy(), method[:-5])
    else:
        err, wxy, mu = _kernels.train_equaliser(E, TrSyms, Niter, os, mu, wxy, modes, adaptive_stepsize, symbols.copy(),
                                                method)
    if apply:
        return apply_filter(E, os, wxy, modes=modes), wxy, err
    return wxy, err


def dual_mode_equalisation(E, os, mu, M, wxy=None, Ntaps=None, TrSyms=(None, None), Niter=(1, 1), methods=("mcma", "sbd"),
                           adaptive_stepsize=(False, False), symbols=None, modes=None, apply=True, **kwargs):
    """
    Two-stage equalisation: stage 2 continues from the taps of stage 1 on the SAME input from sample 0
    (equalisation.py:400-466).  Returns ``(E_eq, wxy, (err1, err2))`` or ``(wxy, (err1, err2))``.

    Deviation: with ``symbols=None`` the reference's ``np.atleast_1d(None)`` turns into a NaN alphabet for
    decision-directed stages; here ``None`` generates the alphabet like ``equalise_signal`` does.
    """
    if symbols is None:
        per_stage = (None, None)
    else:
        symbols = np.atleast_1d(symbols)
        if symbols.ndim < 3:
            symbols = np.tile(symbols, (2, 1, 1))
        per_stage = (symbols[0], symbols[1])
    wxy, err1 = equalise_signal(E, os, mu[0], M, wxy=wxy, Ntaps=Ntaps, TrSyms=TrSyms[0], Niter=Niter[0], method=methods[0],
                                adaptive_stepsize=adaptive_stepsize[0], symbols=per_stage[0], modes=modes, **kwargs)
    wxy2, err2 = equalise_signal(E, os, mu[1], M, wxy=wxy, TrSyms=TrSyms[1], Niter=Niter[1], method=methods[1],
                                 adaptive_stepsize=adaptive_stepsize[1], symbols=per_stage[1], modes=modes, **kwargs)
    if apply:
        return apply_filter(E, os, wxy2, modes=modes), wxy2, (err1, err2)
    return wxy2, (err1, err2)


def equalise_signal_windows(E, os, mu, M, starts, win_len, Ntaps=None, TrSyms=None, Niter=1, method="mcma",
                            adaptive_stepsize=False, symbols=None, modes=None, **kwargs):
    """
    ``equalise_signal(E[:, s:s + win_len], os, mu, M, Ntaps=Ntaps, ...)`` for every ``s`` in ``starts`` in ONE kernel launch
    (all windows are independent and start from centre-spike taps).  Returns ``(wxy (nwin, nmodes, nmodes, Ntaps),
    err (nwin, nmodes, TrSyms*Niter))``.  MI355X-native replacement of the Python loop of the frame synchronisation
    (qampy/core/pilotbased_receiver.py:395-400): a window is one sequential chain, so hundreds of them fill the chip.
    """
    method = method.lower()
    if method in REAL_VALUED or method in DATA_AIDED:
        raise ValueError("window batches support the complex blind / decision-directed methods, not %s" % method)
    E = np.array(np.asarray(E), copy=True, order="C", subok=False)
    mu = E.real.dtype.type(mu)
    nmodes = E.shape[0]
    modes = np.arange(nmodes) if modes is None else np.atleast_1d(modes)
    wxy0 = _init_taps(Ntaps, nmodes, nmodes, E.dtype)
    if TrSyms is None:
        TrSyms = _cal_training_symbol_len(os, Ntaps, win_len)
    symbols = _reshape_symbols(symbols, method, M, E.dtype, nmodes)
    err, wxy, _ = _kernels.train_equaliser_windows(E, starts, win_len, TrSyms, Niter, os, mu, wxy0, modes, adaptive_stepsize,
                                                   symbols.copy(), method)
    return wxy, err
