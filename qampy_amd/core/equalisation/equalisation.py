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
import warnings

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


def _rows_complex(arr, nmodes):
    """One row of constants / alphabet / training symbols per mode for the complex trainers."""
    if arr.ndim <= 1 or arr.shape[0] == 1:
        return np.broadcast_to(arr.reshape(1, -1), (nmodes, arr.size)).copy()          # shared by all modes
    if arr.shape[0] == nmodes:
        return arr
    raise ValueError("symbols of shape %s do not fit %d modes: pass one row (shared) or one row per mode" % (arr.shape, nmodes))


def _rows_real(arr, nrows):
    """Rows for the real-stacked trainers: the first half of the rows belongs to the in-phase, the second to the quadrature
    parts of the modes (packing of _convert_sig_to_real)."""
    half = nrows // 2
    if np.iscomplexobj(arr):
        if arr.ndim <= 1 or arr.shape[0] == 1:
            flat = arr.reshape(-1)
            return np.concatenate([np.broadcast_to(flat.real, (half, flat.size)), np.broadcast_to(flat.imag, (half, flat.size))])
        if arr.shape[0] == half:
            return np.concatenate([arr.real, arr.imag])
        raise ValueError("complex symbols with %d rows do not fit %d complex modes: pass one row or one per mode" % (arr.shape[0], half))
    if arr.shape[0] == 2 and nrows > 2:                                                   # (in-phase row, quadrature row) for all modes
        return np.concatenate([np.broadcast_to(arr[0], (half,) + arr[0].shape), np.broadcast_to(arr[1], (half,) + arr[1].shape)])
    if arr.shape[0] == nrows:
        return arr
    raise ValueError("real symbols of shape %s do not fit the %d rows of the real-stacked field" % (arr.shape, nrows))


def _reshape_symbols(symbols, method, M, dtype, nmodes):
    """``symbols`` in the ``(rows of the field, K)`` layout the trainer indexes (behaviour of equalisation.py:568-594;
    a caller-supplied array is ignored for the blind methods, :569)."""
    if symbols is None or method in NONDECISION_BASED:
        symbols = generate_symbols_for_eq(method, M, dtype)
    arr = np.asarray(symbols)
    rows = _rows_real(arr, nmodes) if method in REAL_VALUED else _rows_complex(arr, nmodes)
    return np.atleast_2d(rows).astype(dtype)


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
    if method not in ("pyt", "hip"):
        raise NotImplementedError("Only the compiled (pyt/hip) method is implemented")
    field = np.array(E, copy=True, order="C", subok=False)
    taps = np.array(wxy, copy=True, order="C", subok=False)
    rows = np.arange(taps.shape[0]) if modes is None else np.array(np.atleast_1d(modes))
    if np.iscomplexobj(field) and np.iscomplexobj(taps):
        return _kernels.apply_filter_to_signal(field, os, taps, rows)
    # real-valued taps act on the field stacked as [real rows; imaginary rows]; the result comes back in the same stacking
    stacked = _convert_sig_to_real(field) if np.iscomplexobj(field) else field
    unit = {8: np.complex128, 4: np.complex64}.get(stacked.itemsize)
    out = _kernels.apply_filter_to_signal(stacked, os, taps, rows)
    if unit is None:
        raise ValueError("The field has an unknown data type")
    return _convert_sig_to_cmplx(out, rows.shape[0], unit(1j))


class _Field:
    """
    The capture of one equaliser call, prepared once: a private C-ordered copy (real-stacked for the real-valued
    trainers), uploaded to HBM once for the complex trainers (:class:`hip_equalisation.ResidentField`) so that every stage
    of a multi-stage call and the final filter run on the same resident array.  The reference copies the field anew for
    each of its compiled calls (equalisation.py:166, :532, three times per dual-mode call).
    """

    def __init__(self, E, real, defer=False):
        """``defer=True`` (only :func:`equalise_signal` / :func:`dual_mode_equalisation`, which call :meth:`finish` before they return): big
        results come back on pinned memory with their copies still in flight.  Every other user gets complete arrays from each call."""
        E = np.asarray(E)
        self.real = real
        # (complex path: the capture is only read - uploaded once - so a C-contiguous array is used as it is; the reference's wrappers copy
        # it, equalisation.py:166-167, 532: 128 MiB of memcpy and page faults per call at C3)
        self.host = _convert_sig_to_real(E) if real else np.ascontiguousarray(np.asarray(E))
        self.rows, self.L = self.host.shape
        self.dtype = self.host.dtype
        self.dev = None if real else _kernels.ResidentField(self.host, defer=bool(defer))

    def finish(self):
        if self.dev is not None:
            self.dev.finish()

    def mode_rows(self, modes):
        """Rows of the field the call trains: the modes, plus their quadrature rows in the real-stacked layout."""
        if modes is None:
            return np.arange(self.rows)
        modes = np.atleast_1d(modes)
        if self.real:
            modes = np.concatenate([modes, modes + self.rows // 2])
        if modes.size and modes.max() >= self.rows:
            raise AssertionError("largest mode number is larger than shape of signal")
        return modes

    def taps(self, wxy, Ntaps):
        """Initial taps: centre spike unless given; given taps are used (and updated) in place when they already have the
        field's dtype and layout."""
        if wxy is None:
            return _init_taps(Ntaps, self.rows, self.rows, self.dtype)
        wxy = np.ascontiguousarray(wxy, dtype=self.dtype)
        if wxy.ndim != 3:
            raise AssertionError("wxy needs to be three dimensional")
        if wxy.shape[:2] != (self.rows, self.rows):
            raise AssertionError("The first 2 dimensions of wxy need to be the same shape as E")
        return wxy

    def train(self, os, mu, M, wxy, TrSyms, Niter, method, adaptive, symbols, rows, pit=None):
        """One training stage on the prepared field; returns ``(taps, err)``.  ``pit``: options of the parallel-in-time solver
        (tier b) or ``None`` for the exact recurrence."""
        n = wxy.shape[-1]
        if TrSyms is None:
            TrSyms = _cal_training_symbol_len(os, n, self.L)
        sy = _reshape_symbols(symbols, method, M, self.dtype, self.rows).copy()
        mu = self.dtype.type(0).real.dtype.type(mu)
        if self.real:
            # (tier b: the real-valued trainer has no parallel-in-time solver - the exact form, reported as such)
            err, wxy, _ = _kernels.train_equaliser_realvalued(self.host, TrSyms, Niter, os, mu, wxy, rows, adaptive, sy, method[:-len("_real")])
            if pit is not None:
                _PIT_REPORTS.append(dict(segments=1, passes=0, converged=True, exact_form=True, tol=float(pit.get("tol", 0) or 1e-3), defect=[], deviation_rms=[],
                                         acquisition=dict(steps=0, chunks=0, mu=0., diverged=False, mean_sq_err=[])))
        else:
            if pit is None:
                err, wxy, _ = self.dev.train(TrSyms, Niter, os, mu, wxy, rows, adaptive, sy, method)
            else:
                err, wxy, _ = self.dev.train(TrSyms, Niter, os, mu, wxy, rows, adaptive, sy, method, pit=pit)
                rep = self.dev.last_report
                _PIT_REPORTS.append(rep)
                # the report is on the host already: never hand back an uncertified result silently
                if rep["acquisition"]["diverged"]:
                    warnings.warn("tier b (%s): the gear-shifted acquisition diverged and was undone; the passes started from the given taps" % method, RuntimeWarning)
                if not rep["converged"]:                     # (only with pit=dict(exact_redo_off=1): by default such a sweep is redone in the exact form)
                    warnings.warn("tier b (%s): the parallel-in-time solver stopped after %d passes WITHOUT reaching its tolerance (%g; last estimate %s): the "
                                  "result is not certified as the sequential recurrence's - use tier='a' or raise max_passes"
                                  % (method, rep["passes"], rep["tol"], ("%.3g" % rep["deviation_rms"][-1]) if rep.get("deviation_rms") else "n/a"), RuntimeWarning)
        return wxy, err

    def filtered(self, os, wxy, rows):
        """The field through the taps, selected rows only; complex result."""
        if not self.real:
            return self.dev.apply(os, wxy, rows)
        out = _kernels.apply_filter_to_signal(self.host, os, np.array(wxy, copy=True, order="C"), rows)
        return _convert_sig_to_cmplx(out, rows.shape[0], (np.complex64 if self.dtype.itemsize == 4 else np.complex128)(1j))


#: device reports (qh_pit_report as dicts) of the tier-b stages of the most recent equalise_signal / dual_mode_equalisation call
_PIT_REPORTS = []
NONFINAL_TOL_FACTOR = 2.0        # tier b, one options dict for several stages: an earlier stage is certified at this multiple of `tol` (pipeline.ResidentReceiver shares it)



def _lib_default_tier():
    from ... import _lib
    return _lib.get_default_tier()


def last_pit_reports():
    """Tier b: what the device decided in the most recent call - one dict per stage (segments, passes, defect per pass,
    converged, acquisition).  ``converged`` False means the result is NOT certified as equivalent to the sequential recurrence."""
    return list(_PIT_REPORTS)


def _tier_options(kwargs, nstages, cold):
    """``tier="a"`` (the default unless ``qampy_amd.set_default_tier("b", tol)`` changed it): the exact sequential recurrence.  ``tier="b"``: the same recurrence solved in parallel in time
     - complex-valued blind / decision-directed methods, fixed step sizes or the reference's adaptive step
    (cma / mcma / sbd / mddma; a sweep the solver cannot certify is redone in the exact form); ``pit`` = one dict of solver options for
    all stages or one per stage.  Returns one options dict (or None) per stage.

    Tolerance per stage: ``tol`` bounds what the CALL returns - equaliser output <= tol, taps and error traces <= 3 tol.  With ONE dict for a
    multi-stage call the LAST stage is certified at ``tol`` and every earlier one at ``NONFINAL_TOL_FACTOR * tol`` (2 tol: it returns an error trace
    and hands on taps, both held to 3 tol; ``last_pit_reports()`` shows the tolerance each stage ran at).  ``pit=dict(tol=..., tol_exact=True)`` or one
    dict per stage holds every stage to exactly the tolerance given."""
    tier, pit = kwargs.pop("tier", None), kwargs.pop("pit", None)
    del _PIT_REPORTS[:]
    if tier is None:                              # the process-wide default (qampy_amd.set_default_tier; "a" unless set)
        tier, tol0 = _lib_default_tier()
        if tier == "b" and pit is None:
            pit = dict(tol=tol0)
    if tier == "a":
        return [None] * nstages
    if tier != "b":
        raise ValueError("tier must be 'a' (exact sequential recurrence) or 'b' (parallel in time)")
    per_stage = [dict(p) for p in pit] if isinstance(pit, (list, tuple)) else [dict(pit or {}) for _ in range(nstages)]
    if len(per_stage) != nstages:
        raise ValueError("pit needs one options dict per stage")
    for k, o in enumerate(per_stage):
        o.setdefault("acquire", 1 if (k == 0 and cold) else 0)     # centre-spike start taps: sequential acquisition first
    if not isinstance(pit, (list, tuple)):
        # `tol` bounds what the call returns (output <= tol, taps and error traces <= 3 tol): the last stage is certified at tol, an earlier one -
        # which returns an error trace and hands on taps - at 2 tol (pipeline.ResidentReceiver.NONFINAL_TOL_FACTOR; one dict per stage overrides)
        if not (pit or {}).get("tol_exact"):
            for o in per_stage[:-1]:
                o["tol"] = NONFINAL_TOL_FACTOR * float(o.get("tol") or 1e-3)
    for o in per_stage:
        o.pop("tol_exact", None)
    return per_stage


def _method_name(method):
    method = method.lower()
    if method not in TRAINING_FCTS:
        raise ValueError("%s is unknown method" % method)
    return method


def equalise_signal(E, os, mu, M, wxy=None, Ntaps=None, TrSyms=None, Niter=1, method="mcma", adaptive_stepsize=False,
                    symbols=None, modes=None, apply=False, **kwargs):
    """
    Blind / decision-directed / data-aided equaliser training (behaviour of equalisation.py:468-566).

    Returns ``(wxy, err)`` or, with ``apply=True``, ``(E_equalised, wxy, err)``.  Unknown keyword arguments are
    swallowed like in the reference; ``tier="b"`` / ``pit=...`` select the parallel-in-time solver (:func:`_tier_options`).
    """
    method = _method_name(method)
    pit = _tier_options(kwargs, 1, wxy is None)[0]
    field = _Field(E, method in REAL_VALUED, defer=True)          # finish() below, before the call returns
    rows = field.mode_rows(modes)
    taps, err = field.train(os, mu, M, field.taps(wxy, Ntaps), TrSyms, Niter, method, adaptive_stepsize, symbols, rows, pit=pit)
    out = field.filtered(os, taps, rows) if apply else None
    field.finish()                                # the error trace crossed PCIe while the filter ran
    if apply:
        return out, taps, err
    return taps, err


def dual_mode_equalisation(E, os, mu, M, wxy=None, Ntaps=None, TrSyms=(None, None), Niter=(1, 1), methods=("mcma", "sbd"),
                           adaptive_stepsize=(False, False), symbols=None, modes=None, apply=True, **kwargs):
    """
    Two training stages on the same field, the second continuing from the taps of the first, from sample 0 both times
    (behaviour of equalisation.py:400-466).  Returns ``(E_eq, wxy, (err1, err2))`` or ``(wxy, (err1, err2))``.
    The field is prepared and uploaded once for both stages and the filter.

    ``symbols``: ``None``, one array for both stages, or an array whose leading axis of length 2 selects the stage (3-d).
    Deviation: with ``symbols=None`` the reference hands decision-directed stages a NaN alphabet; here
    ``None`` generates the alphabet like ``equalise_signal`` does.
    """
    stages = [_method_name(m) for m in methods]
    pits = _tier_options(kwargs, 2, wxy is None)
    real = [m in REAL_VALUED for m in stages]
    if real[0] != real[1]:
        raise ValueError("the two stages must both be complex-valued or both real-valued methods")
    if symbols is None:
        sy = (None, None)
    else:
        s3 = np.asarray(symbols)
        sy = (s3[0], s3[1]) if s3.ndim == 3 else (s3, s3)
    field = _Field(E, real[0], defer=True)                        # finish() below, before the call returns
    rows = field.mode_rows(modes)
    taps = field.taps(wxy, Ntaps)
    errs = []
    for k in range(2):
        taps, err = field.train(os, mu[k], M, taps, TrSyms[k], Niter[k], stages[k], adaptive_stepsize[k], sy[k], rows, pit=pits[k])
        errs.append(err)
    out = field.filtered(os, taps, rows) if apply else None
    field.finish()                                # (stage 1's error trace crossed PCIe while stage 2 trained)
    if apply:
        return out, taps, tuple(errs)
    return taps, tuple(errs)


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


def search_windows(E, os, mu, M, starts, win_len, Ntaps=None, TrSyms=None, Niter=1, method="mcma", adaptive_stepsize=False,
                   symbols=None, modes=None, **kwargs):
    """
    The blind search of the frame synchronisation: an independent equaliser run from centre-spike taps on every window
    ``E[:, s:s + win_len]``, ``s`` in ``starts``, all in one launch, of which only the summary comes back - ``(var (nmodes,
    nwin), best (nmodes,), taps (nmodes, nmodes, nmodes, Ntaps))``: the variance of every window's error trace per mode, the
    index (into ``starts``) of the window with the smallest variance per mode, and the taps that window ended with.
    """
    method = _method_name(method)
    if method in REAL_VALUED or method in DATA_AIDED:
        raise ValueError("window batches support the complex blind / decision-directed methods, not %s" % method)
    E = np.array(np.asarray(E), copy=True, order="C", subok=False)
    nmodes = E.shape[0]
    if TrSyms is None:
        TrSyms = _cal_training_symbol_len(os, Ntaps, win_len)
    return _kernels.train_equaliser_windows_search(E, starts, win_len, TrSyms, Niter, os, E.real.dtype.type(mu), _init_taps(Ntaps, nmodes, nmodes, E.dtype),
                                                   np.arange(nmodes) if modes is None else np.atleast_1d(modes), adaptive_stepsize,
                                                   _reshape_symbols(symbols, method, M, E.dtype, nmodes).copy(), method)
