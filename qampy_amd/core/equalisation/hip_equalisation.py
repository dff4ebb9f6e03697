"""
Drop-in for the compiled module ``qampy.core.equalisation.pythran_equalisation`` on MI355X.

Same entry points, argument order and return values as the ``#pythran export`` lines of
qampy/core/equalisation/pythran_equalisation.py (:33-36, :78-79, :128-129, :238-239, :304-305); the work is done by the
HIP kernels in qampy_amd/csrc/{train,apply,bps}.hip through the C ABI of include/qampy_hip.h.  Like a pythran
extension, the functions insist on exactly matching dtypes (pythran raises TypeError otherwise) and on C-contiguous
inputs.  Arguments may also be :class:`qampy_amd._lib.DeviceArray` objects where noted (``*_dev`` functions).
"""
import ctypes as C

import numpy as np

from ... import _lib
from ..._lib import DeviceArray


def _as_modes(modes, nmax):
    if modes is None:
        return np.arange(nmax, dtype=np.int64)
    return np.ascontiguousarray(np.atleast_1d(modes), dtype=np.int64)


def _adaptive_flag(adaptive, allow_per_mode=True):
    """0 fixed step, 1 the reference's (sequential) adaptive step, 2 ``"per-mode"``: one adapted step size per mode."""
    if isinstance(adaptive, str):
        if adaptive.lower().replace("_", "-") not in ("per-mode", "private"):
            raise ValueError("adaptive step size must be a bool or 'per-mode'")
        if not allow_per_mode:
            raise ValueError("per-mode step sizes are implemented for the complex-valued trainer only")
        return 2
    return int(bool(adaptive))


def _need(arr, dtype, name):
    if not isinstance(arr, np.ndarray) or arr.dtype != dtype or not arr.flags.c_contiguous:
        raise TypeError("%s must be a C-contiguous %s array (got %s %s)" % (
            name, np.dtype(dtype).name, type(arr).__name__, getattr(arr, "dtype", "")))


def train_equaliser(E, TrSyms, Niter, os, mu, wx, modes, adaptive, symbols, method):
    """
    Stochastic-gradient tap training, complex field.  Returns ``(err, wx, mu)``; ``wx`` is also updated in place
    (pythran_equalisation.py:128-173).  ``modes`` are trained in the given order; with ``adaptive`` the adapted step is
    carried from one mode to the next (sequential semantics of the reference).  ``adaptive="per-mode"`` (beyond the
    reference's bool): every mode adapts its own step size from ``mu`` - the result of one call per mode - and the modes
    train concurrently; ``mu`` out is the last mode's.  (The compiled reference lets its OpenMP threads share and race on
    one ``mu``; this is the deterministic counterpart of that behaviour.)
    """
    if method not in _lib.METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    if not np.iscomplexobj(E):
        raise TypeError("train_equaliser needs a complex field; use train_equaliser_realvalued for real arrays")
    _need(E, ct, "E"); _need(wx, ct, "wx")
    symbols = np.ascontiguousarray(symbols)
    _need(symbols, ct, "symbols")
    if E.ndim != 2 or wx.ndim != 3 or symbols.ndim != 2:
        raise TypeError("E must be 2-d, wx 3-d and symbols 2-d")
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    if wx.shape[0] != nmodes or wx.shape[1] != nmodes:
        raise ValueError("wx needs to have at least as many dimensions as the maximum mode")
    if symbols.shape[0] != nmodes:
        raise ValueError("symbols must be at least size of modes")
    modes = _as_modes(modes, nmodes)
    err = np.zeros((nmodes, int(TrSyms) * int(Niter)), dtype=ct)
    mu_c = (C.c_float if rt is np.float32 else C.c_double)(mu)
    _lib.call("qh_train_equaliser_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), nmodes, L, int(TrSyms), int(Niter),
              int(os), C.byref(mu_c), _lib.ptr(wx), ntaps, _lib.ptr(modes), modes.size, _adaptive_flag(adaptive),
              _lib.ptr(symbols), symbols.shape[1], _lib.METHOD_ID[method], _lib.ptr(err))
    return err, wx, rt(mu_c.value)


def train_equaliser_windows(E, starts, win_len, TrSyms, Niter, os, mu, wx0, modes, adaptive, symbols, method):
    """
    Independent equaliser runs on the windows ``E[:, s : s + win_len]`` for ``s`` in ``starts``, all from the taps ``wx0`` and
    the step size ``mu`` - one launch instead of the reference's Python loop over ``equalise_signal`` calls
    (qampy/core/pilotbased_receiver.py:395-400).  Returns ``(err (nwin, nmodes, TrSyms*Niter), wx (nwin, nmodes, nmodes,
    ntaps), mu (nwin,))``, each window's result being what :func:`train_equaliser` returns for that slice.
    """
    if method not in _lib.METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    _need(E, ct, "E"); _need(wx0, ct, "wx0")
    symbols = np.ascontiguousarray(symbols)
    _need(symbols, ct, "symbols")
    nmodes, L = E.shape
    ntaps = wx0.shape[-1]
    modes = _as_modes(modes, nmodes)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    nwin = starts.size
    err = np.zeros((nwin, nmodes, int(TrSyms) * int(Niter)), dtype=ct)
    wx = np.zeros((nwin,) + wx0.shape, dtype=ct)
    mu_out = np.zeros(nwin, dtype=rt)
    _lib.call("qh_train_equaliser_windows_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), nmodes, L, _lib.ptr(starts), nwin,
              int(win_len), int(TrSyms), int(Niter), int(os), rt(mu), _lib.ptr(wx0), ntaps, _lib.ptr(modes), modes.size,
              _adaptive_flag(adaptive, False), _lib.ptr(symbols), symbols.shape[1], _lib.METHOD_ID[method], _lib.ptr(wx), _lib.ptr(err),
              _lib.ptr(mu_out))
    return err, wx, mu_out


def train_equaliser_windows_search(E, starts, win_len, TrSyms, Niter, os, mu, wx0, modes, adaptive, symbols, method):
    """
    :func:`train_equaliser_windows` for a search: the error traces never leave HBM.  Returns ``(var (nmodes, nwin), best
    (nmodes,), wx_best (nmodes, nmodes, nmodes, ntaps))`` - the variance of every window's error trace per mode, the window
    with the smallest one per mode (first minimum) and the taps those windows ended with.
    """
    if method not in _lib.METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    _need(E, ct, "E"); _need(wx0, ct, "wx0")
    symbols = np.ascontiguousarray(symbols)
    _need(symbols, ct, "symbols")
    nmodes, L = E.shape
    modes = _as_modes(modes, nmodes)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    var = np.zeros((nmodes, starts.size), dtype=np.float64)
    best = np.zeros(nmodes, dtype=np.int32)
    wx = np.zeros((nmodes,) + wx0.shape, dtype=ct)
    _lib.call("qh_train_equaliser_windows_search_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), nmodes, L, _lib.ptr(starts), starts.size,
              int(win_len), int(TrSyms), int(Niter), int(os), rt(mu), _lib.ptr(wx0), wx0.shape[-1], _lib.ptr(modes), modes.size,
              _adaptive_flag(adaptive, False), _lib.ptr(symbols), symbols.shape[1], _lib.METHOD_ID[method], _lib.ptr(var), _lib.ptr(best), _lib.ptr(wx))
    return var, best, wx


def train_equaliser_realvalued(E, TrSyms, Niter, os, mu, wx, modes, adaptive, symbols, method):
    """Real-valued trainer (pythran_equalisation.py:78-108); ``method`` without the ``_real`` suffix."""
    if method not in _lib.REAL_METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    if np.iscomplexobj(E):
        raise TypeError("train_equaliser_realvalued needs a real-valued (stacked re/im) field")
    _need(E, rt, "E"); _need(wx, rt, "wx")
    symbols = np.ascontiguousarray(symbols)
    _need(symbols, rt, "symbols")
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    if wx.shape[0] != nmodes or wx.shape[1] != nmodes:
        raise ValueError("wx needs to have at least as many dimensions as the maximum mode")
    if symbols.shape[0] != nmodes:
        raise ValueError("symbols must be at least size of modes")
    modes = _as_modes(modes, nmodes)
    err = np.zeros((nmodes, int(TrSyms) * int(Niter)), dtype=rt)
    mu_c = (C.c_float if rt is np.float32 else C.c_double)(mu)
    _lib.call("qh_train_equaliser_real_f" + suf, _lib.ptr(E), nmodes, L, int(TrSyms), int(Niter), int(os), C.byref(mu_c),
              _lib.ptr(wx), ntaps, _lib.ptr(modes), modes.size, _adaptive_flag(adaptive, False), _lib.ptr(symbols), symbols.shape[1],
              _lib.REAL_METHOD_ID[method], _lib.ptr(err))
    return err, wx, rt(mu_c.value)


def _host_result(shape, dtype):
    """Host array a kernel's output is copied into: from the library's pinned pool when it is large enough for the DMA rate to matter
    (an ordinary ndarray to the caller either way), zero-filled like the reference's ``np.zeros`` when small."""
    n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if n >= _lib.PINNED_MIN_BYTES:
        return _lib.pinned_empty(shape, dtype)
    return np.zeros(shape, dtype=dtype)


def apply_filter_to_signal(E, os, wx, modes=None):
    """Butterfly FIR + decimation, ``out (n_sel, (L-ntaps+1)//os)`` (pythran_equalisation.py:33-76); 4 dtypes."""
    suf, rt, ct = _lib.suffix(E.dtype)
    if os <= 0:
        raise ValueError("oversampling factor must be larger than 0")
    _need(E, E.dtype, "E"); _need(wx, E.dtype, "wx")
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    if wx.shape[1] != nmodes:
        raise ValueError("second dimension of wx must equal the number of input modes")
    modes = _as_modes(modes, wx.shape[0])
    if modes.size and modes.max() >= wx.shape[0]:
        raise ValueError("largest mode number is larger than shape of signal")
    N = max((L - ntaps + 1) // os, 0)
    out = _host_result((modes.size, N), E.dtype)
    name = "qh_apply_filter_" + (("c64" if suf == "32" else "c128") if np.iscomplexobj(E) else "f" + suf)
    if wx.shape[0] != nmodes:
        raise ValueError("wx must be (nmodes, nmodes, ntaps)")
    _lib.call(name, _lib.ptr(E), nmodes, L, int(os), _lib.ptr(wx), ntaps, _lib.ptr(modes), modes.size, _lib.ptr(out))
    return out


def make_decision(E, symbols):
    """Nearest alphabet point, its |distance| and index for every sample (pythran_equalisation.py:304-334)."""
    suf, rt, ct = _lib.suffix(E.dtype)
    E = np.ascontiguousarray(E)
    _need(E, ct, "E")
    symbols = np.ascontiguousarray(symbols)
    _need(symbols, ct, "symbols")
    L = E.shape[0]
    det = np.zeros(L, dtype=ct)
    dist = np.zeros(L, dtype=rt)
    idx = np.zeros(L, dtype=np.int32)
    _lib.call("qh_make_decision_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), L, _lib.ptr(symbols), symbols.size,
              _lib.ptr(det), _lib.ptr(dist), _lib.ptr(idx))
    return det, dist, idx


def det_symbol(X, symbs):
    """Scalar decision ``(symbol, squared distance)`` (pythran_equalisation.py:238-265), evaluated on the device."""
    symbs = np.ascontiguousarray(symbs)
    det, dist, _ = make_decision(np.array([X], dtype=symbs.dtype), symbs)
    return det[0], dist[0] ** 2


class ResidentField:
    """
    A complex capture uploaded to HBM once; trainer stages and the filter of one host-level call run on it there
    (``equalise_signal(apply=True)``: 2 kernels, ``dual_mode_equalisation``: 3 - the reference hands the field to each of its
    compiled calls separately).  ``train`` / ``apply`` keep the contracts of :func:`train_equaliser` /
    :func:`apply_filter_to_signal` minus the field argument: host arrays in and out, ``wx`` updated in place.
    """

    def __init__(self, E, defer=False):
        """``defer``: the big results (error traces, filter output) come back as arrays on pooled pinned memory whose copies are only ENQUEUED -
        on the library's stream 2, behind the stage that produced them, so that the error trace of one stage crosses PCIe while the next stage
        trains - and are complete after :meth:`finish` (the mirrored host layer calls it before it returns).  Default: every method returns
        complete arrays."""
        suf, rt, ct = _lib.suffix(E.dtype)
        if not np.iscomplexobj(E):
            raise TypeError("ResidentField holds a complex field")
        _need(E, ct, "E")
        if E.ndim != 2:
            raise TypeError("E must be 2-d")
        self.shape, self.ct, self.rt = E.shape, ct, rt
        self.dev = DeviceArray.from_host(E)
        self.defer = bool(defer)
        self._pending = []

    def _pit_options(self, pit, TrSyms, os, mu, ntaps, nsel, adaptive):
        """What the resident receiver hands the solver so that a call never waits for the device to tell it something the host knows
        (pipeline.ResidentReceiver): the host copy of the step, the segment grid, the acquisition chunk (the library's own rule on the capture's
        power) and ONE eigenbasis of the capture's window covariance for all stages of the call, built beside the first stage."""
        o = dict(pit)
        nmodes, L = self.shape
        if _adaptive_flag(adaptive) == 0 and float(mu) > 0:
            o.setdefault("mu_hint", float(mu))
            o.setdefault("segments", pit_auto_segments(TrSyms, float(mu), nsel, cold=bool(o.get("acquire"))))
            if o.get("acquire") and not o.get("acq_chunk"):
                o["acq_chunk"] = pit_acq_chunk(self._power(), float(mu), nmodes * int(ntaps), self.rt, o.get("gear"), o.get("acq_bound"))
        if nmodes * int(ntaps) <= 128 and not o.get("basis") and o.get("correction", -1) != 0:
            key = (int(os), int(ntaps), int(TrSyms))
            if getattr(self, "_basis_key", None) != key:
                self._basis = pit_basis_dev(self.dev, os, ntaps, TrSyms, getattr(self, "_basis", None), overlap=True)
                self._basis_key = key
            o["basis"] = self._basis.ptr
        return o

    def _power(self):
        if getattr(self, "_pw", None) is None:
            head = DeviceArray((self.shape[0], min(self.shape[1], 4096)), self.ct)
            for r in range(self.shape[0]):                     # the first 4096 samples of every row, as pit_setup_kernel averages them
                _lib.call("qh_memcpy_d2d", head.row(r).ptr, self.dev.row(r).ptr, head.row(r).nbytes)
            self._pw = float(np.mean(np.abs(head.to_host().astype(np.complex128)) ** 2))
        return self._pw

    def _result(self, darr):
        """Device -> host copy of a finished result: one DMA into pooled pinned memory, on stream 2 behind an event of the current stream."""
        if darr.nbytes < _lib.PINNED_MIN_BYTES:
            return darr.to_host()
        ev = _lib.Event()
        ev.record()
        _lib.call("qh_use_stream", 2)
        try:
            _lib.call("qh_stream_wait_event", ev.ptr)
            out = darr.to_host(pinned=True, wait=not self.defer)
        finally:
            _lib.call("qh_use_stream", 0)
        if self.defer:
            self._pending.append((darr, ev))           # the device array stays alive until the copy has run
        return out

    def finish(self):
        """Wait for the copies :meth:`train` / :meth:`apply` left in flight (``defer=True``)."""
        if self._pending:
            _lib.sync()
            del self._pending[:]

    def train(self, TrSyms, Niter, os, mu, wx, modes, adaptive, symbols, method, pit=None):
        """``pit``: ``None`` for the exact sequential recurrence, or a dict of ``qh_pit_opts`` fields for the parallel-in-time
        solver (tier b); its device report is kept in ``self.last_report``."""
        if method not in _lib.METHOD_ID:
            raise ValueError("Unknown method %s" % method)
        nmodes, L = self.shape
        _need(wx, self.ct, "wx")
        symbols = np.ascontiguousarray(symbols)
        _need(symbols, self.ct, "symbols")
        if wx.ndim != 3 or symbols.ndim != 2:
            raise TypeError("wx must be 3-d and symbols 2-d")
        if wx.shape[:2] != (nmodes, nmodes):
            raise ValueError("wx needs to have at least as many dimensions as the maximum mode")
        if symbols.shape[0] != nmodes:
            raise ValueError("symbols must be at least size of modes")
        dw, dsy = DeviceArray.from_host(wx), DeviceArray.from_host(symbols)
        dmu = DeviceArray.from_host(np.array([mu], dtype=self.rt))
        derr = DeviceArray((nmodes, int(TrSyms) * int(Niter)), self.ct)
        self.last_report = None
        if pit is None:
            train_equaliser_dev(self.dev, TrSyms, Niter, os, dmu, dw, modes, adaptive, dsy, method, derr, zero_err=True)
        else:
            rep = PitReportBuffer()
            train_equaliser_dev(self.dev, TrSyms, Niter, os, dmu, dw, modes, adaptive, dsy, method, derr, zero_err=True,
                                pit=self._pit_options(pit, TrSyms, os, mu, wx.shape[-1], _as_modes(modes, nmodes).size, adaptive), report=rep)
            self.last_report = rep.read()
        wx[...] = dw.to_host()
        mu_out = self.rt(dmu.to_host()[0])
        err = self._result(derr)
        if self.defer:
            # freeing device memory waits for the whole device (like hipFree): with the error trace's copy in flight that wait would serialise
            # it with the next stage - the small operands of this stage are therefore released in finish(), after the copies
            self._pending.append((dw, dsy, dmu, locals().get("rep")))
        return err, wx, mu_out

    def apply(self, os, wx, modes=None):
        if os <= 0:
            raise ValueError("oversampling factor must be larger than 0")
        nmodes, L = self.shape
        wx = np.ascontiguousarray(wx, dtype=self.ct)
        if wx.ndim != 3 or wx.shape[0] != nmodes or wx.shape[1] != nmodes:
            raise ValueError("wx must be (nmodes, nmodes, ntaps)")
        modes = _as_modes(modes, nmodes)
        if modes.size and modes.max() >= nmodes:
            raise ValueError("largest mode number is larger than shape of signal")
        N = max((L - wx.shape[-1] + 1) // os, 0)
        out = DeviceArray((modes.size, N), self.ct)
        dw = DeviceArray.from_host(wx)
        apply_filter_to_signal_dev(self.dev, os, dw, modes, out)
        res = self._result(out)
        if self.defer:
            self._pending.append((dw,))
        return res


class ResidentJobs:
    """
    Independent training jobs of identical shape - slices of one capture that start at different samples, ONE output mode
    each (``equalize_pilot_sequence`` with unequal shifts, qampy/core/pilotbased_receiver.py:497-547) - resident together and
    trained together: one launch per stage over all jobs through the channel-bank entry points, instead of one chain after
    the other.  Job ``j`` trains row ``job_modes[j]`` of the taps on ``slices[j]``; rows of different output modes do not
    interact, so the result is what one :func:`train_equaliser` call per job with ``modes=[job_modes[j]]`` returns (the
    adaptive step is therefore run with one step size per chain, each from the initial ``mu``).
    """

    def __init__(self, slices, job_modes):
        host = np.ascontiguousarray(np.stack([np.asarray(x) for x in slices]))
        suf, rt, ct = _lib.suffix(host.dtype)
        if host.ndim != 3 or not np.iscomplexobj(host):
            raise TypeError("ResidentJobs holds complex slices of identical shape")
        self.shape, self.ct, self.rt = host.shape, ct, rt
        self.job_modes = [int(m) for m in job_modes]
        self.dev = DeviceArray.from_host(host)

    def train(self, TrSyms, Niter, os, mu, wx, adaptive, symbols, method):
        if method not in _lib.METHOD_ID:
            raise ValueError("Unknown method %s" % method)
        nj, nmodes, L = self.shape
        _need(wx, self.ct, "wx")
        symbols = np.ascontiguousarray(symbols)
        _need(symbols, self.ct, "symbols")
        bank = np.ascontiguousarray(np.broadcast_to(wx, (nj,) + wx.shape))
        dw, dsy = DeviceArray.from_host(bank), DeviceArray.from_host(symbols)
        dmu = DeviceArray.from_host(np.full(nj, mu, dtype=self.rt))
        derr = DeviceArray((nj, nmodes, int(TrSyms) * int(Niter)), self.ct)
        train_equaliser_batch_dev(self.dev, TrSyms, Niter, os, dmu, dw, None, "per-mode" if adaptive else False, dsy, method, derr, zero_err=True)
        out = dw.to_host()
        for j, m in enumerate(self.job_modes):
            wx[m] = out[j, m]
        return wx

    def train_bank(self, TrSyms, Niter, os, mu, bank, adaptive, symbols, method):
        """Like :meth:`train` with one tap set PER JOB in and out (``bank (njobs, nmodes, nmodes, ntaps)``): jobs that do not share their
        start taps, e.g. the frames of a capture (job ``j`` still only moves row ``job_modes[j]`` of its set)."""
        if method not in _lib.METHOD_ID:
            raise ValueError("Unknown method %s" % method)
        nj, nmodes, L = self.shape
        bank = np.ascontiguousarray(bank, dtype=self.ct)
        symbols = np.ascontiguousarray(symbols)
        _need(symbols, self.ct, "symbols")
        dw, dsy = DeviceArray.from_host(bank), DeviceArray.from_host(symbols)
        dmu = DeviceArray.from_host(np.full(nj, mu, dtype=self.rt))
        derr = DeviceArray((nj, nmodes, int(TrSyms) * int(Niter)), self.ct)
        # every job trains ITS mode only: one launch per output mode over the jobs that carry it would need a gather; the bank entry point
        # trains all selected modes of all jobs - rows of other modes are discarded below (they never feed back into the kept row)
        train_equaliser_batch_dev(self.dev, TrSyms, Niter, os, dmu, dw, None, "per-mode" if adaptive else False, dsy, method, derr, zero_err=True)
        out = dw.to_host()
        res = np.array(bank, copy=True)
        for j, m in enumerate(self.job_modes):
            res[j, m] = out[j, m]
        return res

    def apply_bank(self, os, bank):
        """Row ``job_modes[j]`` of slice ``j`` through job ``j``'s own taps, stacked."""
        nj, nmodes, L = self.shape
        N = max((L - bank.shape[-1] + 1) // os, 0)
        rows = []
        for j, m in enumerate(self.job_modes):
            out = DeviceArray((1, N), self.ct)
            apply_filter_to_signal_dev(self.dev.row(j), os, DeviceArray.from_host(np.ascontiguousarray(bank[j], dtype=self.ct)), np.array([m]), out)
            rows.append(out.to_host()[0])
        return np.array(rows)

    def apply(self, os, wx):
        """Row ``job_modes[j]`` of slice ``j`` through the taps, stacked."""
        nj, nmodes, L = self.shape
        N = max((L - wx.shape[-1] + 1) // os, 0)
        dw = DeviceArray.from_host(np.ascontiguousarray(wx, dtype=self.ct))
        rows = []
        for j, m in enumerate(self.job_modes):
            out = DeviceArray((1, N), self.ct)
            apply_filter_to_signal_dev(self.dev.row(j), os, dw, np.array([m]), out)
            rows.append(out.to_host()[0])
        return np.array(rows)


# ------------------------------------------------------------------------------------------------ device-resident forms
def gram_build_dev(E, os, ntaps, TrSyms):
    """
    Build the Gram terms of the look-ahead trainer for a resident capture (they depend on ``E, os, ntaps, TrSyms`` only)
    and return the opaque device pointer to hand to :func:`train_equaliser_dev` for every mode / stage / sweep over the
    same capture.  The buffer is library-owned and valid until the next call of this function.
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    nmodes, L = E.shape
    g = C.c_void_p()
    _lib.call("qh_gram_build_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nmodes, L, int(os), int(ntaps), int(TrSyms),
              C.byref(g))
    return g.value


def train_equaliser_dev(E, TrSyms, Niter, os, mu, wx, modes, adaptive, symbols, method, err, zero_err=False, gram=None, pit=None,
                        report=None):
    """
    Same as :func:`train_equaliser` with every array (and the scalar ``mu``, a 1-element DeviceArray) already in HBM.
    Only enqueues work on the library stream.

    ``pit`` (a dict, default ``None`` = the reference's exact sequential recurrence) selects the opt-in parallel-in-time
    trainer (``qh_train_equaliser_*_pit_dev``): concurrently trained segments made consistent by waveform
    relaxation until the boundary defect is below ``tol``.  Keys (all optional): ``segments`` (0 = automatic),
    ``max_passes``, ``tol``, ``acquire`` (cold start: gear-shifted acquisition first), ``phase_seed``, ``gear``,
    ``acq_bound``, ``acq_plateau``, ``acq_chunk``, ``acq_max``.  ``report``: a :class:`PitReportBuffer` the device fills.
    With ``adaptive=True`` (the reference's shared step size) the output modes are solved in turn, each from the step size the one
    before it ended with.  Tier b is total: a sweep the passes do not certify is redone in the exact form inside the call, and a
    call no parallel-in-time solver exists for (data-aided methods, one step size per mode, the adaptive step with cma2 / rde / mrde /
    dd or complex128) takes the exact form right away - the report says ``exact_form`` in both cases, the result is the reference's.
    """
    if method not in _lib.METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    modes = _as_modes(modes, nmodes)
    args = (E.ptr, nmodes, L, int(TrSyms), int(Niter), int(os), mu.ptr, wx.ptr, ntaps, _lib.ptr(modes), modes.size,
            _adaptive_flag(adaptive), symbols.ptr, symbols.shape[1], _lib.METHOD_ID[method], err.ptr, int(bool(zero_err)))
    name = "qh_train_equaliser_c" + ("64" if suf == "32" else "128")
    if pit is not None:
        o = _lib.PitOpts()
        o.phase_seed = -1
        o.correction = -1
        o.corr_beta = -1.
        for k, v in pit.items():
            if not hasattr(o, k):
                raise ValueError("unknown parallel-in-time option %s" % k)
            setattr(o, k, v)
        if _adaptive_flag(adaptive) == 2:
            # one step size per mode (the extension): no parallel-in-time solver - the library takes the exact form and says so in the report
            o.adaptive = 2
        elif _adaptive_flag(adaptive):
            # the reference carries ONE step size from mode to mode (pythran_equalisation.py:163-172): the modes are solved in turn, each
            # from the step size the one before it ended with (one report per mode, PitReportBuffer.read aggregates)
            o.adaptive = 1
            if report is not None:
                report.per_mode = []
            for m in modes:
                one = np.array([m], dtype=np.int64)
                _lib.call(name + "_pit_dev", E.ptr, nmodes, L, int(TrSyms), int(Niter), int(os), mu.ptr, wx.ptr, ntaps, _lib.ptr(one), 1,
                          symbols.ptr, symbols.shape[1], _lib.METHOD_ID[method], err.ptr, int(bool(zero_err)), None, C.byref(o),
                          report.ptr if report is not None else None)
                if report is not None:      # one report per mode (the device buffer holds the last call's): the adaptive solve synchronises anyway
                    report.per_mode.append(report.read_one())
                zero_err = False
            return
        if report is not None:
            report.per_mode = None
        _lib.call(name + "_pit_dev", E.ptr, nmodes, L, int(TrSyms), int(Niter), int(os), mu.ptr, wx.ptr, ntaps, _lib.ptr(modes), modes.size,
                  symbols.ptr, symbols.shape[1], _lib.METHOD_ID[method], err.ptr, int(bool(zero_err)), gram, C.byref(o),
                  report.ptr if report is not None else None)
    elif gram:
        _lib.call(name + "_gram_dev", *args, gram)
    else:
        _lib.call(name + "_dev", *args)


class PitReportBuffer(DeviceArray):
    """Device memory for one ``qh_pit_report``; :meth:`read` synchronises and returns it as a dict.  After a solve with the
    adaptive step (one library call per output mode, in the reference's mode order) the dict is the LAST mode's report with the
    fields that must not be lost aggregated over the modes - ``converged`` (all), ``exact_form`` (any), ``passes`` (largest) - and
    every mode's own report under ``per_mode``."""

    per_mode = None

    def __init__(self):
        super().__init__((C.sizeof(_lib.PitReport),), np.uint8, zero=True)

    def read_one(self):
        _lib.sync()
        raw = self.to_host()
        return _lib.PitReport.from_buffer_copy(raw.tobytes()).as_dict()

    def read(self):
        if not self.per_mode:
            return self.read_one()
        rep = dict(self.per_mode[-1])
        rep["converged"] = all(r["converged"] for r in self.per_mode)
        rep["exact_form"] = any(r["exact_form"] for r in self.per_mode)
        rep["passes"] = max(r["passes"] for r in self.per_mode)
        rep["per_mode"] = [dict(r) for r in self.per_mode]
        return rep


def pit_basis_dev(E, os, ntaps, TrSyms, basis=None, overlap=False):
    """
    Eigenbasis of the input covariance of a resident capture for the coarse correction of the parallel-in-time trainer
    (``qh_pit_basis_*_dev``); depends on ``E, os, ntaps, TrSyms`` only, so one build serves every stage.  Returns the
    DeviceArray holding it (pass ``basis`` to reuse an allocation); hand its ``ptr`` to ``pit=dict(basis=...)``.
    ``overlap``: build it on the library's other stream while the current one goes on (the trainer waits where it needs it).
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    nmodes, L = E.shape
    if basis is None:
        n = C.c_size_t(0)
        _lib.call("qh_pit_basis_bytes", nmodes * int(ntaps), C.byref(n))
        basis = DeviceArray((n.value,), np.uint8)
    _lib.call("qh_pit_basis_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nmodes, L, int(os), int(ntaps), int(TrSyms), basis.ptr, int(bool(overlap)))
    return basis


def _pit_opts(pit):
    o = _lib.PitOpts()
    o.phase_seed = -1
    o.correction = -1
    o.corr_beta = -1.
    for k, v in pit.items():
        if not hasattr(o, k):
            raise ValueError("unknown parallel-in-time option %s" % k)
        setattr(o, k, v)
    return o


def pit_prepare_bytes(nmodes, ntaps, acq_steps, dtype=np.complex64):
    n = C.c_size_t(0)
    _lib.call("qh_pit_prepare_bytes", int(nmodes), int(ntaps), int(acq_steps), np.dtype(dtype).itemsize, C.byref(n))
    return n.value


def pit_prepare_dev(E, TrSyms, os, mu, wx0, modes, symbols, method, pit, prep):
    """The acquisition of a cold tier-b sweep AHEAD of the training call (``qh_pit_prepare_c64_dev``), enqueued on the current library stream: ``wx0``
    the start taps (not modified), ``mu`` the 1-element DeviceArray of the step, ``pit`` the options of the training call (``segments``, ``mu_hint``,
    ``acq_chunk`` given), ``prep`` a DeviceArray of :func:`pit_prepare_bytes`.  Hand ``prep.ptr`` to the training call as ``pit["prepared"]`` (after
    ordering the two streams).  Returns False - nothing enqueued - when the sweep's acquisition cannot run ahead."""
    suf, rt, ct = _lib.suffix(E.dtype)
    if suf != "32":
        return False
    nmodes, L = E.shape
    modes = _as_modes(modes, nmodes)
    o = _pit_opts(pit)
    try:
        _lib.call("qh_pit_prepare_c64_dev", E.ptr, nmodes, L, int(TrSyms), int(os), mu.ptr, wx0.ptr, wx0.shape[-1], _lib.ptr(modes), modes.size,
                  symbols.ptr, symbols.shape[1], _lib.METHOD_ID[method], C.byref(o), prep.ptr, prep.nbytes)
    except ValueError as e:
        if "not preparable" in str(e):
            return False
        raise
    return True


def pit_last_timing():
    """Kernel time of the trainer launches of the most recent parallel-in-time call: ``(pass_ms list, acquisition ms)``."""
    buf = (C.c_float * _lib.PIT_MAXPASS)()
    n, acq = C.c_int(0), C.c_float(0)
    _lib.call("qh_pit_last_timing", buf, _lib.PIT_MAXPASS, C.byref(n), C.byref(acq))
    return [float(buf[i]) for i in range(n.value)], float(acq.value)


def pit_acq_chunk(power, mu, ntot, rt=np.float32, gear=None, bound=None):
    """Chunk length of a cold sweep's acquisition by the library's own rule (csrc/train_pit.h: pit_setup_kernel + train_pit_dev):
    ``mu_acq = clamp(gear mu, mu, bound / (ntot power))`` in the capture's precision, chunk = ``2 / mu_acq`` rounded to the nearest power of two,
    256 .. 4096.  ``power``: mean |sample|^2 over the first ``min(L, 4096)`` samples of all rows."""
    gear, bound = float(gear or 8.0), float(bound or 0.08)
    ma = min(gear * mu, bound / (int(ntot) * max(float(power), 1e-30)))
    if not ma > mu:
        ma = mu
    ma = max(float(rt(ma)), 1e-12)
    return int(min(max(2 ** int(np.floor(np.log2(2.0 / ma) + 0.5)), 256), 4096))


def pit_effective_segments(S, TrSyms):
    """Segments the library actually uses when asked for ``S``: at least four 64-step blocks per segment (csrc/train_pit.h)."""
    return max(min(int(S), (int(TrSyms) // 64) // 4), 1)


def pit_auto_segments(TrSyms, mu, nsel=1, cold=False):
    """The library's automatic segment count for a sweep of ``TrSyms`` steps at step size ``mu`` (1 = sequential); ``cold``:
    the sweep starts from unconverged taps (acquisition first)."""
    n = C.c_int(0)
    _lib.call("qh_pit_auto_segments", int(TrSyms), float(mu), int(nsel), int(bool(cold)), C.byref(n))
    return n.value


def gram_build_batch_dev(E, os, ntaps, TrSyms):
    """Gram terms of a channel bank ``E (nch, nmodes, L)``: one table per channel, one opaque pointer for the bank."""
    suf, rt, ct = _lib.suffix(E.dtype)
    nch, nmodes, L = E.shape
    g = C.c_void_p()
    _lib.call("qh_gram_build_c" + ("64" if suf == "32" else "128") + "_batch_dev", E.ptr, nch, nmodes, L, int(os), int(ntaps), int(TrSyms),
              C.byref(g))
    return g.value


def train_equaliser_batch_dev(E, TrSyms, Niter, os, mu, wx, modes, adaptive, symbols, method, err, zero_err=False, gram=None):
    """
    :func:`train_equaliser_dev` for a bank of independent captures with identical shapes: ``E (nch, nmodes, L)``,
    ``wx (nch, nmodes, nmodes, ntaps)``, ``err (nch, nmodes, TrSyms*Niter)``, ``mu (nch,)``; ``symbols`` and ``modes`` are
    shared.  Every channel gets exactly the single-capture result; the exact trainers run all channels concurrently
    (one workgroup per channel and output mode).
    """
    if method not in _lib.METHOD_ID:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _lib.suffix(E.dtype)
    nch, nmodes, L = E.shape
    ntaps = wx.shape[-1]
    modes = _as_modes(modes, nmodes)
    _lib.call("qh_train_equaliser_c" + ("64" if suf == "32" else "128") + "_batch_dev", E.ptr, nch, nmodes, L, int(TrSyms), int(Niter), int(os),
              mu.ptr, wx.ptr, ntaps, _lib.ptr(modes), modes.size, _adaptive_flag(adaptive), symbols.ptr, symbols.shape[1],
              _lib.METHOD_ID[method], err.ptr, int(bool(zero_err)), gram)


def apply_filter_to_signal_dev(E, os, wx, modes, out):
    suf, rt, ct = _lib.suffix(E.dtype)
    nmodes, L = E.shape
    modes = _as_modes(modes, wx.shape[0])
    _lib.call("qh_apply_filter_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nmodes, L, int(os), wx.ptr, wx.shape[-1],
              _lib.ptr(modes), modes.size, out.ptr)


def make_decision_dev(E, symbols, det, dist, idx):
    suf, rt, ct = _lib.suffix(E.dtype)
    L = int(np.prod(E.shape))
    _lib.call("qh_make_decision_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, L, symbols.ptr, int(np.prod(symbols.shape)),
              det.ptr if det is not None else None, dist.ptr if dist is not None else None, idx.ptr)
