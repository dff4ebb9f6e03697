"""
Device-resident receiver: dual-mode equalisation + filter application + blind phase search without leaving HBM.

The reference chains ``dual_mode_equalisation`` (qampy/core/equalisation/equalisation.py:400-466) and ``bps``
(qampy/core/phaserecovery.py:93-159) through host ndarrays; on MI355X the capture, the taps, the error traces and the
recovered symbols stay in HBM between the stages (SURVEY.md §8b "optional persistent device buffers") and every stage is
one or a few kernel launches on the library stream.  Host-side logic (constants, tap initialisation, default training
length) is shared with :mod:`qampy_amd.core.equalisation.equalisation`, so both paths take identical decisions.

Layout in HBM (complex64 for the headline configs):
    E        (nmodes, L)            capture, 2 samples/symbol, row-major (coalesced along time)
    wxy      (nmodes, nmodes, Nt)   taps (1312 B at 2x2x41)
    err1/2   (nmodes, TrSyms*Niter) error traces of the two stages
    eq       (nmodes, N)            equalised, decimated signal, N = (L - Nt + 1)//os
    idx, ph, out (nmodes, N)        BPS index, applied phase, phase-recovered symbols
"""
import os as _os

import numpy as np

from . import _lib
from ._lib import DeviceArray
from .core.equalisation import equalisation as _host
from .core.equalisation import hip_equalisation as _k
from .core import hip_dsp as _dsp


# Tier b holds EVERY stage to the same tolerance (library default 1e-3, estimated rms deviation of the equaliser output from
# the sequential recurrence): what a seeding stage leaves in the weakly excited tap directions is handed to the next stage
# undamped, so a looser first stage would show up one-to-one in the result (round 2 used 0.06 there).


class ResidentReceiver:
    """
    One capture's worth of preallocated device state.

    Parameters mirror ``dual_mode_equalisation`` + ``bps``: ``mu``, ``Niter``, ``methods``, ``adaptive_stepsize`` are
    2-tuples (pass ``methods=("mcma",)`` etc. with 1-tuples for a single stage), ``Mtestangles``/``Nbps`` the BPS
    parameters (``Mtestangles=None`` skips carrier recovery).
    """

    NONFINAL_TOL_FACTOR = _host.NONFINAL_TOL_FACTOR

    def __init__(self, nmodes, L, os, M, Ntaps, mu, methods=("cma", "mrde"), Niter=(1, 1), adaptive_stepsize=(False, False),
                 TrSyms=(None, None), Mtestangles=64, Nbps=20, dtype=np.complex64, alphabet=None, modes=None, tier="a", pit=None):
        suf, self.rt, self.ct = _lib.suffix(dtype)
        self.nmodes, self.L, self.os, self.M, self.Ntaps = int(nmodes), int(L), int(os), int(M), int(Ntaps)
        # run(overlap=True), tier b: the pending phase search in parts between the next capture's relaxation passes.  0: as many parts as the previous
        # capture had passes, less one; 1: one launch beside the first passes (rounds 3-4); n > 1: n parts.  (QAMPY_POST_PARTS: initial value, measurements)
        self.post_parts = int(_os.environ.get("QAMPY_POST_PARTS", "0"))
        self.nstage = len(methods)
        self.methods = tuple(m.lower() for m in methods)
        for m in self.methods:
            if m in _host.REAL_VALUED or m in _host.DATA_AIDED:
                raise ValueError("the resident pipeline runs the blind / decision-directed complex methods; %s is not one" % m)
            if m not in _host.TRAINING_FCTS:
                raise ValueError("%s is unknown method" % m)
        self.Niter = tuple(int(n) for n in Niter)
        self.adaptive = tuple(a if isinstance(a, str) else bool(a) for a in adaptive_stepsize)      # True, False or "per-mode"
        self.modes = np.arange(nmodes) if modes is None else np.atleast_1d(modes)
        self.TrSyms = tuple(_host._cal_training_symbol_len(os, Ntaps, L) if t is None else int(t) for t in TrSyms[:self.nstage])
        self.N = (self.L - self.Ntaps + 1) // self.os
        self.Mtestangles, self.Nbps = Mtestangles, Nbps
        # tier "a" (default): the reference's exact sequential recurrence.  tier "b" (opt-in): parallel-in-time training
        #  - concurrently trained segments + waveform relaxation until the boundary defect is below `tol`; the
        # first stage is a cold start (gear-shifted acquisition), later stages start from the previous stage's taps.
        # `pit`: options of hip_equalisation.train_equaliser_dev(pit=...), one dict for all stages or one per stage
        if tier not in ("a", "b"):
            raise ValueError("tier must be 'a' (exact) or 'b' (parallel in time)")
        self.tier = tier
        pit = pit or {}
        self.pit = [dict(p) for p in pit] if isinstance(pit, (tuple, list)) else [dict(pit) for _ in methods]
        for o in self.pit:
            if o.get("acq_chunk"):
                o["_acq_chunk_user"] = True        # the caller fixed the chunk length: load() leaves it alone
        self._owner_thread = None                  # set by ReceiverGroup: the stages of this receiver are enqueued by that thread only
        # Tolerance per stage (round 5).  `tol` bounds what the CALL returns: equaliser output <= tol, taps <= 3 tol, error traces <= 3 tol.
        # The device's stop rule holds the estimated output deviation of a sweep to its `tol` - which is what the LAST stage's filter output needs; an
        # earlier stage returns an error trace and hands on taps, both held to 3 tol, so it is certified at NONFINAL_TOL_FACTOR x tol (2: the error
        # functions turn an output deviation into 1.3-1.5 x as much trace deviation, measured).  One `pit` dict per stage overrides this.
        if tier == "b" and not isinstance(pit, (tuple, list)):
            for s_ in range(self.nstage - 1):
                self.pit[s_]["tol"] = self.NONFINAL_TOL_FACTOR * float(self.pit[s_].get("tol") or 1e-3)
        self.pit_report = None
        self.pit_timing = [([], 0.) for _ in methods]
        self.mu0 = tuple(self.rt(m) for m in mu)
        if alphabet is None:
            alphabet = _host.generate_symbols_for_eq("sbd", M, self.ct)[0]
        self.alphabet_host = np.ascontiguousarray(alphabet, dtype=self.ct)
        # ---- device state
        self.E = DeviceArray((nmodes, L), self.ct)
        self.wxy = DeviceArray((nmodes, nmodes, Ntaps), self.ct)
        self.wxy0 = DeviceArray.from_host(_host._init_taps(Ntaps, nmodes, nmodes, self.ct))
        self.mu = [DeviceArray.from_host(np.array([m], dtype=self.rt)) for m in self.mu0]
        self.mu_init = [DeviceArray.from_host(np.array([m], dtype=self.rt)) for m in self.mu0]
        self.symbols = []
        for m in self.methods:
            sy = _host._reshape_symbols(self.alphabet_host if m in _host.DECISION_BASED else None, m, M, self.ct, nmodes)
            self.symbols.append(DeviceArray.from_host(sy))
        self.err = [DeviceArray((nmodes, self.TrSyms[s] * self.Niter[s]), self.ct, zero=True) for s in range(self.nstage)]
        self.eq = DeviceArray((self.modes.size, self.N), self.ct)
        if Mtestangles:
            self.alphabet = DeviceArray.from_host(self.alphabet_host)
            self.angles = DeviceArray.from_host(_dsp.test_angle_grid(Mtestangles, self.rt))
            self.idx = DeviceArray((self.modes.size, self.N), np.int32)
            self.ph = DeviceArray((self.modes.size, self.N), self.rt)
            self.out = DeviceArray((self.modes.size, self.N), self.ct)
        if tier == "b":
            self.pit_report = [_k.PitReportBuffer() for _ in methods]
            for s_, o in enumerate(self.pit):          # segment grid from the host copy of mu: the call then never synchronises
                o.setdefault("acquire", 1 if s_ == 0 else 0)
                if not self.adaptive[s_]:              # fixed step: the host copy spares the call the read-back of *mu (qh_pit_opts.mu_hint)
                    o.setdefault("mu_hint", float(self.mu0[s_]))
                if self.adaptive[s_]:                  # adaptive step: the library's own grid (exact head + 2048-step segments), modes in turn
                    o.setdefault("segments", 0)
                else:
                    o.setdefault("segments", _k.pit_auto_segments(self.TrSyms[s_], float(self.mu0[s_]), self.modes.size, cold=bool(o["acquire"])))
        _lib.sync()

    # ------------------------------------------------------------------------------------------ data movement
    def load(self, E):
        """Host -> HBM copy of the capture (outside any timed region).  Tier b: the chunk length of a cold stage's acquisition follows the
        capture's power (2 / the gear-shifted step, rounded to a power of two: csrc/train_pit.h) - it is re-derived here for EVERY capture from
        the host copy, by the library's own rule, so that the result of a capture never depends on which capture the receiver saw first and
        no call has to wait for the device to say it."""
        E = np.ascontiguousarray(np.asarray(E), dtype=self.ct)
        assert E.shape == (self.nmodes, self.L)
        self._filter_done()
        self.E.set(E)
        if getattr(self, "_prep", None) is not None:
            self.invalidate()
        self._acq_asked = [False] * self.nstage
        if self.tier == "b":
            npow = min(self.L, 4096)
            power = float(np.mean(np.abs(E[:, :npow].astype(np.complex128)) ** 2))
            for s_, o in enumerate(self.pit):
                if o.get("acquire") and not self.adaptive[s_] and not o.get("_acq_chunk_user"):
                    o["acq_chunk"] = self._acq_chunk_rule(power, float(self.mu0[s_]), o)
                    self._acq_asked[s_] = True

    def _acq_chunk_rule(self, power, mu, o):
        return _k.pit_acq_chunk(power, mu, self.nmodes * self.Ntaps, self.rt, o.get("gear"), o.get("acq_bound"))

    # ------------------------------------------------------------------------------------------ stages (enqueue only)
    def reset(self):
        """Centre-spike taps and initial step sizes (start of ``dual_mode_equalisation``)."""
        self.wxy.copy_from(self.wxy0)
        for s_, (m, m0) in enumerate(zip(self.mu, self.mu_init)):
            if self.adaptive[s_]:                  # (a fixed step is only read by the trainers: nothing to restore - a copy is ~6 us of stream 0 between two captures)
                m.copy_from(m0)

    def _bound(self):
        """A receiver of a ReceiverGroup caches state of its worker thread (that thread's library streams, events, scratch slots - a Gram table
        pointer into them): its stages must be enqueued by that thread (``group.map(fn, which=[i])``), or nothing orders them against the
        worker's streams."""
        import threading
        if self._owner_thread is not None and threading.get_ident() != self._owner_thread:
            raise RuntimeError("this receiver belongs to a ReceiverGroup: enqueue its stages on its worker thread (group.map(lambda rx: ..., which=[i]) or group.run)")

    def build_gram(self):
        """Gram terms of the look-ahead trainer: once per capture, shared by all modes, stages and sweeps."""
        self._bound()
        self._gram = None
        if self.tier == "a":
            # shared table only when a block form will read it (>= 128 steps) and it fits the library's scratch budget - otherwise
            # the trainers chunk the sweep themselves (csrc/train_impl.h: gram_budget), which a caller's table would switch off
            import os as _os
            budget = _lib.gram_budget_gb() * 2 ** 30
            fits = self.TrSyms[0] * 1024 * (np.dtype(self.ct).itemsize // 8) <= budget
            if len(set(self.TrSyms)) == 1 and self.TrSyms[0] >= 128 and fits and _lib.get_form("trainer") != 1:
                self._gram = _k.gram_build_dev(self.E, self.os, self.Ntaps, self.TrSyms[0])
        elif len(set(self.TrSyms)) == 1 and self.nmodes * self.Ntaps <= 128:
            # tier b builds what its passes need itself (no Gram table in the throughput form, csrc/train_seg.h); what the stages
            # share is the eigenbasis of the capture's input covariance for the coarse correction
            self._basis = _k.pit_basis_dev(self.E, self.os, self.Ntaps, self.TrSyms[0], getattr(self, "_basis", None), overlap=True)
            for o in self.pit:
                o["basis"] = self._basis.ptr

    def train(self, stage, hook=None, pass_hook=None):
        """``hook`` (tier b): a callable the library invokes once, right after it has enqueued the sweep's first relaxation pass
        (``qh_pit_opts.on_pass0``) - work for the other library streams is enqueued there, while the device is busy, instead of in front of
        this sweep's launches; called after the sweep if the library never got to a first pass (exact form).

        ``pass_hook(sweep, p)`` (tier b): invoked right before the trainer launch of EVERY relaxation pass (``qh_pit_opts.on_pass``) - see
        ``run(overlap=True)``."""
        tb = self.tier == "b"
        self._bound()
        opts = {k: v for k, v in self.pit[stage].items() if not k.startswith("_")} if tb else None
        if tb and stage == 0 and getattr(self, "_use_prep", None) is not None:
            opts["prepared"] = self._use_prep.ptr
            self._use_prep = None
        fired, failed = [False], []
        if pass_hook is not None and tb:
            import ctypes as _C

            def _pcb(_user, sweep, p):
                try:
                    pass_hook(int(sweep), int(p))
                except BaseException as e:            # (never through the C frames: re-raised below)
                    failed.append(e)
                finally:
                    _lib.call("qh_use_stream", 0)
            self._pass_hook_keepalive = _lib.PIT_PASS_HOOK(_pcb)
            opts["on_pass"] = _C.cast(self._pass_hook_keepalive, _C.c_void_p).value
        if hook is not None and tb:
            import ctypes as _C

            def _cb(_user):
                if not fired[0]:
                    fired[0] = True
                    try:
                        hook()
                    except BaseException as e:        # (never through the C frames: re-raised below)
                        failed.append(e)
                    finally:
                        _lib.call("qh_use_stream", 0)
            self._hook_keepalive = _lib.PIT_HOOK(_cb)
            opts["on_pass0"] = _C.cast(self._hook_keepalive, _C.c_void_p).value
        _k.train_equaliser_dev(self.E, self.TrSyms[stage], self.Niter[stage], self.os, self.mu[stage], self.wxy, self.modes,
                               self.adaptive[stage], self.symbols[stage], self.methods[stage], self.err[stage],
                               gram=getattr(self, "_gram", None), pit=opts, report=self.pit_report[stage] if tb else None)
        if hook is not None and not fired[0]:
            fired[0] = True
            hook()
        if failed:
            raise failed[0]
        if tb:
            self.pit_timing[stage] = _k.pit_last_timing()      # host-side copy of the HIP-event times: no synchronisation
            o = self.pit[stage]
            asked = getattr(self, "_acq_asked", None) or [False] * self.nstage
            if o.get("acquire") and not o.get("acq_chunk") and not self.adaptive[stage] and not asked[stage]:
                # a capture that was put into self.E directly (device-side synthesis, tests): the chunk length the device derived for it is
                # read back ONCE (this synchronises: the one read of the receiver's life, whatever the report says - a sweep the library
                # sent to the exact form has no acquisition to report and is not asked again); load() derives it on the host instead
                asked[stage] = True
                self._acq_asked = asked
                a = self.pit_report[stage].read()["acquisition"]
                if a["chunks"] > 0 and a["steps"] > 0:
                    o["acq_chunk"] = int(a["steps"] // a["chunks"])

    def pit_reports(self):
        """Tier b: what the device decided in the last run, one dict per stage (segments, passes, boundary defects,
        acquisition); synchronises."""
        return [r.read() for r in self.pit_report] if self.pit_report else None

    def apply(self):
        """Filter the capture with the current taps (a phase search still pending from run(overlap=True) is completed first: it reads
        the filter output this call overwrites)."""
        self.wait_post()
        self._apply()

    def recover(self):
        """Phase search, unwrap and de-rotation of the filter output (after a pending one, see apply)."""
        self.wait_post()
        self._recover()

    def _apply(self):
        self._bound()
        _k.apply_filter_to_signal_dev(self.E, self.os, self.wxy, self.modes, self.eq)

    def _recover(self, part=0, nparts=1):
        self._bound()
        _dsp.bps_recover_dev(self.eq, self.Mtestangles, self.alphabet, self.Nbps, self.idx, self.ph, self.out, angles=self.angles, part=part, nparts=nparts)

    # ------------------------------------------------------------------------------------------ the next capture's sequential prologue, ahead of time
    def load_next(self, E):
        """Host -> HBM copy of the capture that follows the one in ``self.E`` (second input buffer): ``run(prefetch=True)`` processes ``E``, prepares
        this one beside it and swaps the two buffers when it is done with ``E`` - so the call sequence of a streaming receiver is ``load(c0)``,
        ``load_next(c1)``, ``run()``, ``load_next(c2)``, ``run()``, ..."""
        E = np.ascontiguousarray(np.asarray(E), dtype=self.ct)
        assert E.shape == (self.nmodes, self.L)
        if getattr(self, "E_next", None) is None:
            self.E_next = DeviceArray((self.nmodes, self.L), self.ct)
        self._filter_done()                        # (the buffer may be the capture a filter on stream 2 is still reading)
        self.E_next.set(E)
        self._next_loaded = True

    def _filter_done(self):
        """run(overlap=True) leaves the filter of the last capture on stream 2: stream 0 waits for it before an input buffer is written."""
        if getattr(self, "_filter_pending", False):
            _lib.call("qh_stream_wait_event", self._ev_ready.ptr)
            self._filter_pending = False

    def invalidate(self):
        """Forget what was prepared ahead (after the capture in ``self.E`` / ``self.E_next`` was changed by hand)."""
        self._prep_ready = [False, False]

    def _prepare_next(self):
        """Tier b, cold first stage: the acquisition and the eigenbasis of the NEXT capture - both depend on the capture only, not on anything this
        capture's training produces - are enqueued on library stream 1 now, so that they run beside this capture's cold stage (whose passes leave
        32 compute units free for exactly that, csrc/train_pit.h) instead of in front of the next capture's passes (qh_pit_prepare_*_dev).  The next
        capture is ``E_next`` (``load_next``), or - a receiver fed the same resident capture again and again, as bench.py does - ``E`` itself."""
        o = self.pit[0]
        if self.tier != "b" or not o.get("acquire") or self.adaptive[0] or self.ct != np.complex64 or not getattr(self, "_basis", None):
            return
        if not (o.get("acq_chunk") and o.get("mu_hint") and o.get("segments", 0) > 1):
            return
        self._prep_init()
        slot = 1 - self._prep_cur
        amax = 2 * ((int(o["acq_chunk"]) + 63) // 64 * 64) if not o.get("acq_max") else int(o["acq_max"])
        need = _k.pit_prepare_bytes(self.nmodes, self.Ntaps, amax, self.ct)
        if self._prep[slot] is None or self._prep[slot].nbytes < need:
            self._prep[slot] = DeviceArray((need,), np.uint8)
        nxt = self.E_next if getattr(self, "_next_loaded", False) else self.E
        _lib.call("qh_use_stream", 1)             # (_ev_main: recorded by run() on stream 0 when the next capture was in place)
        try:
            _lib.call("qh_stream_wait_event", self._ev_main.ptr)
            if getattr(self, "_post_running", False):
                _lib.call("qh_stream_wait_event", self._ev_post.ptr)        # (the previous capture's phase search: see run)
            # (NOT behind the first trainer launch like the parts of the phase search: its chip-wide kernels - setup, Gram terms, covariance - then share
            # the ~70 us of analysis behind passes 0 and 1 with the control path, which is the critical path: gaps of 130 / 123 instead of 90 us,
            # 1120-1124 against 1135-1144 MSym/s; beside pass 0 they cost that pass 2 %)
            ok = _k.pit_prepare_dev(nxt, self.TrSyms[0], self.os, self.mu_init[0], self.wxy0, self.modes, self.symbols[0], self.methods[0],
                                    {k: v for k, v in o.items() if not k.startswith("_") and k not in ("basis", "prepared")}, self._prep[slot])
            if ok:
                self._basis_alt = _k.pit_basis_dev(nxt, self.os, self.Ntaps, self.TrSyms[0], self._basis_alt, overlap=False)
                self._ev_prep[slot].record()
        finally:
            _lib.call("qh_use_stream", 0)
        self._prep_ready[slot] = bool(ok)

    def _prep_init(self):
        if getattr(self, "_prep", None) is None:
            self._prep, self._prep_ready, self._prep_cur = [None, None], [False, False], 0
            self._basis_alt, self._ev_main, self._ev_prep = None, _lib.Event(), [_lib.Event(), _lib.Event()]

    def _adopt_prepared(self):
        """Start of a run: if the previous run prepared this capture, its eigenbasis and its acquisition are there already."""
        self._use_prep = None
        if getattr(self, "_prep", None) is None:
            return False
        slot = 1 - self._prep_cur
        if not self._prep_ready[slot]:
            return False
        self._prep_ready[slot] = False
        self._prep_cur = slot
        self._basis, self._basis_alt = self._basis_alt, self._basis
        _lib.call("qh_stream_wait_event", self._ev_prep[slot].ptr)
        for o in self.pit:
            o["basis"] = self._basis.ptr
        self._use_prep = self._prep[slot]
        return True

    def run(self, overlap=False, mark=None, prefetch=False):
        """One pass of the hot path over the resident capture; returns without synchronising.

        ``overlap=True`` (a receiver that is handed capture after capture): the phase search of this pass is left PENDING and goes
        onto the library's stream 2 when the next ``run`` has enqueued its covariance kernel (the one chip-wide streaming kernel of
        a tier-b pass, which would otherwise queue up behind the phase search and hold back the eigen-solver), so that it runs
        beside the training of the next pass on stream 0 - the trainers are latency-bound chains of one wave per SIMD that leave
        half of the register file and most issue slots free, the phase search is a chip-wide streaming kernel of 64 registers a
        wave that fits into them (csrc/train_seg.h; stream 2 stays off 32 of the 256 compute units, csrc/api.hip).  The results of
        a pass are complete after ``wait_post()`` (enqueues what is pending; stream 0 waits for it) or ``fetch()`` (the host waits
        too).  Bit-identical to ``overlap=False``: the same kernels on the same data, in another order.

        ``prefetch=True`` (tier b, consecutive captures): the acquisition and the eigenbasis of the NEXT capture are enqueued on stream 1 at the
        start of this run and adopted by the next one (``_prepare_next``); results bit-identical to ``prefetch=False``.

        ``mark(name)`` (bench.py): called on the stream the stage was enqueued on after "start", "gram", "train<s>", "apply", "bps",
        and around the overlapped phase search ("post_begin", "post_end")."""
        m = mark or (lambda name: None)
        self.reset()
        m("start")
        adopted = self._adopt_prepared() if prefetch else False
        if not adopted:
            self._use_prep = None
            self.build_gram()
        m("gram")
        # The pending phase search IN PARTS (tier b): one part beside each relaxation pass of this capture's training (pass_hook below).  A pass keeps
        # the chip's SIMDs busy with one latency-bound wave each, and the whole search beside it - as many single-wave workgroups per SIMD as the LDS
        # allows - costs it a third of its speed (0.36 instead of 0.27 ms per pass).  A part of an eighth of the search is ONE wave per SIMD: gated by an
        # event in front of a pass's trainer launch it starts with the trainer, lives on the issue slots the trainer's lone waves leave (0-3 % of the
        # pass) and the events space the parts one pass apart.  (Gated BEHIND the trainer launch the parts start in the ~70 us of analysis between two
        # passes and share the chip with the control path, which is the critical path: 1137-1146 against 1163-1167 MSym/s at C3.)  Number of parts: the
        # trainer launches of the previous capture less two (one may have been enqueued in vain ahead of the decision that ended a sweep; the last part
        # also unwraps and de-rotates and should not hold up this capture's filter, which overwrites what it reads); what is left when the training is
        # over is enqueued then.  Bit-identical to one launch.
        parts_mode = self.tier == "b" and int(getattr(self, "post_parts", 0)) != 1
        self._hook_calls = 0
        if parts_mode and getattr(self, "_post_pending", False):
            fixed = int(getattr(self, "post_parts", 0))
            self._post_n = fixed if fixed > 1 else max(1, min(16, int(getattr(self, "_hook_calls_prev", 10)) - 2))
            self._post_next, self._post_mark = 0, m
        if getattr(self, "_ev_pass", None) is None:
            self._ev_pass = _lib.Event()

        def on_pass(sweep, p):
            self._hook_calls += 1
            pend = parts_mode and getattr(self, "_post_pending", False)
            if pend:                              # (nothing to gate: no packet on stream 0 either)
                self._ev_pass.record()            # stream 0: in front of the trainer launch of this pass
            if pend:
                self._post_part(gated=True)

        def side_work():
            if not parts_mode:
                self._enqueue_post(m)             # phase search of the previous overlapped pass, beside the stages below
            if prefetch:
                # ... and BEHIND that phase search: started together with it, the next capture's covariance kernel (chip-wide, streaming), eigen-solver
                # and acquisition crowd the first pass of this capture - 454 instead of 290 us, the model kernels 319 instead of 47
                # (profiles/r05_timeline_prefetch_v1.txt); the phase search is through ~1 ms into the capture, the preparation then has the rest of
                # the cold stage (whose passes leave 32 CUs free)
                self._prepare_next()
        # With prefetch the ~25 launches of that side work are made from INSIDE the first stage's call, right after its first pass is enqueued
        # (qh_pit_opts.on_pass0): in front of it they cost ~0.2 ms of host time during which stream 0 had nothing to run.
        defer = bool(prefetch and self.tier == "b" and adopted)
        if prefetch and self.tier == "b":
            self._prep_init()
            self._ev_main.record()                # stream 0 up to here: the capture the preparation reads is in place
        if not defer:
            side_work()
        for s in range(self.nstage):
            self.train(s, hook=side_work if (defer and s == 0) else None, pass_hook=on_pass if parts_mode else None)
            m("train%d" % s)
        if parts_mode:
            self._hook_calls_prev = self._hook_calls if self._hook_calls > 0 else getattr(self, "_hook_calls_prev", 9)
            self._enqueue_post(m)                 # parts the passes did not take
        # The filter of this capture.  One capture at a time it follows the training on stream 0.  Consecutive captures (tier b, overlap): it goes onto
        # stream 2 - IN FRONT of this capture's phase search, which reads its output, and behind the previous capture's search, which read the buffer it
        # overwrites: the stream's own order is all the ordering those need - with a private copy of the taps (the next run's reset() restores the start
        # taps), so that stream 0 goes straight on to the next capture: the filter (87 us chip-wide at C3) and its launch gaps were ~0.1 ms of the ~0.2 ms
        # between the last pass of a capture and the first pass of the next one (round 6; profiles/r06_c3_timeline.txt).  Bit-identical.
        aside = bool(overlap and self.Mtestangles and self.tier == "b" and getattr(self, "filter_aside", True))
        if aside:
            if getattr(self, "_wxy_f", None) is None:
                self._wxy_f = DeviceArray(self.wxy.shape, self.ct)
            if getattr(self, "_ev_trained", None) is None:
                self._ev_trained = _lib.Event()
            if getattr(self, "_ev_post", None) is None:
                self._ev_ready, self._ev_post = _lib.Event(), _lib.Event()
            self._wxy_f.copy_from(self.wxy)
            self._ev_trained.record()
            _lib.call("qh_use_stream", 2)
            try:
                _lib.call("qh_stream_wait_event", self._ev_trained.ptr)
                _k.apply_filter_to_signal_dev(self.E, self.os, self._wxy_f, self.modes, self.eq)
                m("apply")
                self._ev_ready.record()            # stream 2: the filter has read the capture and written its output (load_next / load wait for it)
            finally:
                _lib.call("qh_use_stream", 0)
            self._post_running = False             # (the previous search is in front of this filter on stream 2; wait_post() waits for the NEW search's event)
            self._filter_pending = True
        else:
            if getattr(self, "_post_running", False):
                _lib.call("qh_stream_wait_event", self._ev_post.ptr)        # the filter output of the previous pass has been consumed
                self._post_running = False
            self._apply()
            m("apply")
        if prefetch and getattr(self, "_next_loaded", False):
            # the capture load_next() brought - prepared above, where that was possible - is the next one to be processed; the old one's buffer takes
            # the next load_next()
            self.E, self.E_next = self.E_next, self.E
            self._next_loaded = False
        if not self.Mtestangles:
            return
        if not overlap:
            self._recover()
            m("bps")
            return
        if getattr(self, "_ev_post", None) is None:
            self._ev_ready, self._ev_post = _lib.Event(), _lib.Event()
        if not aside:
            self._ev_ready.record()               # stream 0 up to here: the filter output the pending phase search reads
        self._post_pending = True
        self._post_n = self._post_next = 0        # (not started: the next run decides whether it goes in parts)

    def _post_part(self, gated):
        """The next part of the pending phase search onto stream 2; ``gated``: behind ``_ev_pass`` (recorded in front of the trainer launch that is
        about to go onto stream 0)."""
        i, n = self._post_next, self._post_n
        mark = getattr(self, "_post_mark", None)
        _lib.call("qh_use_stream", 2)
        try:
            if gated:
                _lib.call("qh_stream_wait_event", self._ev_pass.ptr)
            if i == 0:
                _lib.call("qh_stream_wait_event", self._ev_ready.ptr)      # the filter output the search reads
                if mark:
                    mark("post_begin")
            self._recover(part=i, nparts=n)
            if i == n - 1:
                if mark:
                    mark("post_end")
                self._ev_post.record()
        finally:
            _lib.call("qh_use_stream", 0)
        self._post_next = i + 1
        if i == n - 1:
            self._post_pending, self._post_running = False, True

    def _enqueue_post(self, mark=None):
        if not getattr(self, "_post_pending", False):
            return
        if getattr(self, "_post_n", 0) and getattr(self, "_post_next", 0) > 0:      # a search already under way in parts: the rest of them, now
            while getattr(self, "_post_pending", False):
                self._post_part(gated=False)
            self._post_n = 0
            return
        self._post_n = 0
        _lib.call("qh_use_stream", 2)             # (_ev_ready: recorded behind the filter of the pass whose phase search is pending)
        try:
            _lib.call("qh_stream_wait_event", self._ev_ready.ptr)
            if mark:
                mark("post_begin")
            self._recover()
            if mark:
                mark("post_end")
            self._ev_post.record()
        finally:
            _lib.call("qh_use_stream", 0)
        self._post_pending, self._post_running = False, True

    def wait_post(self, mark=None):
        """After ``run(overlap=True)``: enqueue the pending phase search; work enqueued from here on (stream 0) sees its results (and the filter's)."""
        self._enqueue_post(mark)
        self._filter_done()
        if getattr(self, "_post_running", False):
            _lib.call("qh_stream_wait_event", self._ev_post.ptr)
            self._post_running = False

    # ------------------------------------------------------------------------------------------ results
    def fetch(self):
        """Synchronise and copy the results to the host as a dict of ndarrays."""
        self.wait_post()
        _lib.sync()
        res = dict(wxy=self.wxy.to_host(), err=tuple(e.to_host() for e in self.err), eq=self.eq.to_host(),
                   mu=tuple(m.to_host()[0] for m in self.mu))
        if self.Mtestangles:
            res.update(out=self.out.to_host(), ph=self.ph.to_host(), idx=self.idx.to_host())
        return res

    def ser(self, symbols_tx, maxlag=256, window=4096, trim=0):
        """Symbol error rate of the recovered (else equalised) rows against the transmitted symbols, computed in HBM
        (``qampy_amd.core.ber_functions.cal_ser_dev``); needs the alphabet.  Returns one dict per row."""
        from .core import ber_functions as _ber
        if self.alphabet_host is None:
            raise ValueError("ser() needs the alphabet")
        if getattr(self, "alphabet", None) is None:
            self.alphabet = DeviceArray.from_host(self.alphabet_host)
        self.wait_post()
        if getattr(self, "_idx_tx", None) is None or self._idx_tx_src is not symbols_tx:
            self._idx_tx = _ber.tx_indices_dev(np.ascontiguousarray(symbols_tx, dtype=self.ct), self.alphabet)
            self._idx_tx_src = symbols_tx
        return _ber.cal_ser_dev(self.out if self.Mtestangles else self.eq, self._idx_tx, self.alphabet, maxlag, window, trim)

    def bytes_per_symbol(self):
        """Algorithmic HBM bytes per symbol period of one run() (general formula)."""
        cs = np.dtype(self.ct).itemsize
        nsel = self.modes.size
        train = sum(self.Niter[s] * cs * (self.nmodes * self.os + nsel) for s in range(self.nstage))
        apply_ = cs * (self.nmodes * self.os + nsel)
        bps = nsel * (cs + cs + cs // 2) if self.Mtestangles else 0
        return dict(train=train, apply=apply_, bps=bps, total=train + apply_ + bps)


class ReceiverGroup:
    """
    ``n`` captures IN FLIGHT on one GPU: ``n`` :class:`ResidentReceiver` objects of identical shape, each driven by its own host thread.
    The library keeps streams, scratch buffers and the tier-b solver's events per host thread (csrc/api.hip), so the receivers share
    nothing but the chip: what one capture leaves idle (between the relaxation passes of its trainer, during acquisition and the
    eigen-solver) the others use.  Every capture gets exactly the result of a receiver that runs alone (same kernels, same data); what
    changes is when the work is scheduled.  Measured at C3 (profiles/r04_in_flight.txt): 2 receivers 1.1x the throughput of one; two
    receiver PROCESSES on the GPU reach 1.2x (their streams do not share hardware queues).

        group = ReceiverGroup(2, nmodes, L, os, M, Ntaps, mu, tier="b", ...)      # the arguments of ResidentReceiver
        group.load(E)                          # the same capture into every receiver (or group.rx[i].load(E_i))
        group.run(steps)                       # `steps` passes in total, dealt round robin; returns when all of them are complete
        group.rx[i].fetch()                    # results of the last pass of receiver i
    """

    def __init__(self, n, *args, factory=None, sync=None, release=None, **kw):
        """``factory`` / ``sync`` / ``release`` (tests): what builds a receiver, waits for the calling thread's streams and releases them -
        ResidentReceiver, the library's qh_sync and qh_thread_release by default."""
        import queue
        import threading
        if n < 1:
            raise ValueError("at least one receiver")
        self._sync = sync if sync is not None else _lib.sync
        self._release = release if release is not None else (lambda: _lib.call("qh_thread_release"))
        self.rx = [(factory or ResidentReceiver)(*args, **kw) for _ in range(int(n))]
        if len(self.rx) > 1:
            # several captures in flight fill one another's gaps; a phase search held back for the gaps of ONE of them only waits (C3, three receivers:
            # 1175 MSym/s with the search in one launch, 993 in parts)
            for r in self.rx:
                if hasattr(r, "post_parts") and "QAMPY_POST_PARTS" not in _os.environ:
                    r.post_parts = 1
        self._sync()
        self._jobs = [queue.SimpleQueue() for _ in self.rx]
        self._done = queue.SimpleQueue()
        self._threads = [threading.Thread(target=self._work, args=(i,), daemon=True, name="qampy-receiver-%d" % i) for i in range(len(self.rx))]
        for t in self._threads:
            t.start()

    def _work(self, i):
        import threading
        if hasattr(self.rx[i], "_owner_thread"):
            self.rx[i]._owner_thread = threading.get_ident()
        while True:
            job = self._jobs[i].get()
            if job is None:
                self._release()                    # this thread's streams and scratch buffers
                return
            try:
                self._done.put((i, job(self.rx[i]), None))
            except BaseException as e:             # handed to the caller of run() / map()
                self._done.put((i, None, e))

    def map(self, fn, which=None):
        """``fn(receiver)`` on the thread of every receiver (or of those in ``which``) at the same time; the list of results."""
        which = range(len(self.rx)) if which is None else list(which)
        for i in which:
            self._jobs[i].put(fn)
        res, err = {}, None
        for _ in which:
            i, r, e = self._done.get()
            res[i], err = r, (e if e is not None else err)
        if err is not None:
            raise err
        return [res[i] for i in which]

    def load(self, E):
        for r in self.rx:
            r.load(E)
        self._sync()

    def run(self, steps, overlap=True, mark=None, prefetch=False, feed=None):
        """``steps`` passes of the hot path in total, receiver ``i`` taking passes i, i + n, ...; returns when all are complete on the
        device.  ``overlap``: as ResidentReceiver.run.  ``mark(i, k)``: a mark callback for pass k of receiver i (bench.py).
        ``feed(i, k, rx)``: called on receiver ``i``'s thread before its ``k``-th pass - where a caller hands it the capture that follows
        (``rx.load_next(...)``, or a resident array as ``rx.E_next`` with ``rx._next_loaded = True``)."""
        n = len(self.rx)
        share = [len(range(i, int(steps), n)) for i in range(n)]

        def job(i):
            def go(rx):
                for k in range(share[i]):
                    if feed is not None:
                        feed(i, k, rx)
                    rx.run(overlap=overlap, mark=mark(i, k) if mark else None, **({"prefetch": True} if prefetch else {}))
                rx.wait_post(mark(i, share[i]) if mark else None)
                self._sync()                   # this thread's streams
            return go
        for i in range(n):
            self._jobs[i].put(job(i))
        err = None
        for _ in range(n):
            _, _, e = self._done.get()
            err = e if e is not None else err
        if err is not None:
            raise err

    def pit_reports(self):
        return [r.pit_reports() for r in self.rx]

    def close(self):
        for q_ in self._jobs:
            q_.put(None)
        for t in self._threads:
            t.join(timeout=10.)
        self._threads = []
        for r in self.rx:                          # what pointed into the worker threads' (now released) scratch buffers and streams
            for name in ("_gram", "_ev_post", "_ev_ready", "_ev_trained"):
                if hasattr(r, name):
                    setattr(r, name, None)
            for name in ("_post_pending", "_post_running", "_filter_pending"):
                if hasattr(r, name):
                    setattr(r, name, False)
            if hasattr(r, "_owner_thread"):
                r._owner_thread = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChannelBank:
    """
    ``nch`` independent captures of identical shape (WDM channels) resident on ONE GPU and processed
    together.  One exact training chain occupies one workgroup, i.e. 1/256 of the chip, so the natural way to fill an
    MI355X with the exact recurrence is many channels side by side: the trainers take the channel as ``blockIdx.y``
    (one launch per stage for the whole bank), the Gram tables, the filter and the phase search - already chip-wide per
    capture - run channel after channel on the same stream.  Per channel the results are bit-identical to a
    :class:`ResidentReceiver` of that capture.

    HBM per channel (complex64, 2 modes, 2 samples/symbol, N symbols): capture 32 N, Gram table 1024 N, error traces 16 N per
    stage, outputs 40 N bytes - about 1.1 kB per symbol, so 288 GB hold e.g. 64 channels of 2^22 symbols.
    """

    def __init__(self, nch, nmodes, L, os, M, Ntaps, mu, trainer="auto", **kw):
        self.nch = int(nch)
        # trainer: "auto" = per-method choice of the single-capture path (look-ahead for cma-type stages: lowest latency, but
        # its Gram table is 1 KiB per step); "iterative" = block-iterative form for every stage: 10-25 % longer stages,
        # a quarter of the Gram table (256 B per step, built 2.8x faster) - more channels fit and aggregate throughput is higher
        self.trainer = {"auto": 0, "direct": 1, "lookahead": 2, "iterative": 3}[trainer]
        # one receiver object provides the shared constants and the per-channel views; its own big buffers are replaced
        self.rx = ResidentReceiver(nmodes, 8 * Ntaps * os, os, M, Ntaps, mu, **kw)       # tiny dummy capture
        r = self.rx
        if r.tier != "a":
            raise ValueError("a channel bank runs the exact recurrence (tier A)")
        self.nmodes, self.L, self.os, self.Ntaps, self.ct, self.rt = int(nmodes), int(L), int(os), int(Ntaps), r.ct, r.rt
        TrSyms = kw.get("TrSyms", (None,) * r.nstage)
        self.TrSyms = tuple(_host._cal_training_symbol_len(os, Ntaps, L) if t is None else int(t) for t in TrSyms[:r.nstage])
        self.N = (self.L - self.Ntaps + 1) // self.os
        nsel = r.modes.size
        self.E = DeviceArray((self.nch, nmodes, L), self.ct)
        self.wxy = DeviceArray((self.nch, nmodes, nmodes, Ntaps), self.ct)
        self.wxy0 = DeviceArray.from_host(np.tile(_host._init_taps(Ntaps, nmodes, nmodes, self.ct), (self.nch, 1, 1, 1)))
        self.mu = [DeviceArray((self.nch,), self.rt) for _ in range(r.nstage)]
        self.mu_init = [DeviceArray.from_host(np.full(self.nch, m, dtype=self.rt)) for m in r.mu0]
        self.err = [DeviceArray((self.nch, nmodes, self.TrSyms[s] * r.Niter[s]), self.ct, zero=True) for s in range(r.nstage)]
        self.eq = DeviceArray((self.nch, nsel, self.N), self.ct)
        if r.Mtestangles:
            self.idx = DeviceArray((self.nch, nsel, self.N), np.int32)
            self.ph = DeviceArray((self.nch, nsel, self.N), self.rt)
            self.out = DeviceArray((self.nch, nsel, self.N), self.ct)
        self._gram = None
        _lib.sync()

    def load(self, ch, E):
        E = np.ascontiguousarray(np.asarray(E), dtype=self.ct)
        assert E.shape == (self.nmodes, self.L)
        self.E.row(ch).set(E)

    def run(self):
        """One pass of the hot path over all channels; enqueue only."""
        r = self.rx
        self.wxy.copy_from(self.wxy0)
        for m, m0 in zip(self.mu, self.mu_init):
            m.copy_from(m0)
        _lib.call("qh_set_trainer", self.trainer)
        try:
            self._run_stages(r)
        finally:
            _lib.call("qh_set_trainer", 0)

    def run_pipelined(self, steps):
        """
        ``steps`` passes over the bank, software-pipelined over the library's two streams: the trainers (and Gram tables) of
        pass k+1 run on stream 0 while filter + phase search of pass k run on stream 1 with a copy of that pass's taps.  The
        trainers keep every CU busy with one or two workgroups of 8 wavefronts - issue-limited, not throughput-limited - so
        chip-wide per-channel kernels can run beside them as far as registers and LDS allow: measured at 128 channels, 8 passes,
        987 instead of 1131 ms per pass with a 512-thread phase search, nothing with the (faster) 1024-thread one that is
        built in, whose workgroups claim every vector register of a SIMD.  Same results as :meth:`run`; enqueue only
        (``_lib.sync()`` to wait).
        """
        from ._lib import Event
        r = self.rx
        if getattr(self, "_taps_b", None) is None:
            self._taps_b = [DeviceArray(self.wxy.shape, self.ct) for _ in range(2)]
            self._ev_trained, self._ev_post = [Event(), Event()], [Event(), Event()]
        try:
            for k in range(steps):
                p = k & 1
                _lib.call("qh_use_stream", 0)
                if k >= 2:
                    _lib.call("qh_stream_wait_event", self._ev_post[p].ptr)        # the taps copy of pass k-2 has been consumed
                self.wxy.copy_from(self.wxy0)
                for m, m0 in zip(self.mu, self.mu_init):
                    m.copy_from(m0)
                _lib.call("qh_set_trainer", self.trainer)
                self._train_stages(r)
                _lib.call("qh_set_trainer", 0)
                self._taps_b[p].copy_from(self.wxy)
                self._ev_trained[p].record()
                _lib.call("qh_use_stream", 1)
                _lib.call("qh_stream_wait_event", self._ev_trained[p].ptr)
                self._post_stages(r, self._taps_b[p])
                self._ev_post[p].record()
        finally:
            _lib.call("qh_set_trainer", 0)
            _lib.call("qh_use_stream", 0)

    def _run_stages(self, r):
        self._train_stages(r)
        self._post_stages(r, self.wxy)

    def _train_stages(self, r):
        # one Gram table per channel shared by the stages when the bank's tables fit the library's scratch budget; otherwise
        # the trainers build them per time chunk themselves (csrc/train_impl.h: gram_budget)
        budget = _lib.gram_budget_gb() * 2 ** 30
        per_step = (256 if self.trainer == 3 else 1024) * (np.dtype(self.ct).itemsize // 8)
        fits = self.nch * self.TrSyms[0] * per_step <= budget
        self._gram = _k.gram_build_batch_dev(self.E, self.os, self.Ntaps, self.TrSyms[0]) if (len(set(self.TrSyms)) == 1 and fits) else None
        for s in range(r.nstage):
            _k.train_equaliser_batch_dev(self.E, self.TrSyms[s], r.Niter[s], self.os, self.mu[s], self.wxy, r.modes, r.adaptive[s],
                                         r.symbols[s], r.methods[s], self.err[s], gram=self._gram)

    def _post_stages(self, r, wxy):
        for c in range(self.nch):
            _k.apply_filter_to_signal_dev(self.E.row(c), self.os, wxy.row(c), r.modes, self.eq.row(c))
            if r.Mtestangles:
                _dsp.bps_recover_dev(self.eq.row(c), r.Mtestangles, r.alphabet, r.Nbps, self.idx.row(c), self.ph.row(c), self.out.row(c), angles=r.angles)

    def ser(self, ch, symbols_tx, maxlag=256, window=4096, trim=0):
        """Per-row symbol errors of channel ``ch`` (device harness, see :meth:`ResidentReceiver.ser`)."""
        from .core import ber_functions as _ber
        r = self.rx
        if getattr(r, "alphabet", None) is None:
            r.alphabet = DeviceArray.from_host(r.alphabet_host)
        idx_tx = _ber.tx_indices_dev(np.ascontiguousarray(symbols_tx, dtype=self.ct), r.alphabet)
        return _ber.cal_ser_dev((self.out if r.Mtestangles else self.eq).row(ch), idx_tx, r.alphabet, maxlag, window, trim)

    def fetch(self, ch):
        _lib.sync()
        res = dict(wxy=self.wxy.row(ch).to_host(), err=tuple(e.row(ch).to_host() for e in self.err), eq=self.eq.row(ch).to_host(),
                   mu=tuple(m.to_host()[ch] for m in self.mu))
        if self.rx.Mtestangles:
            res.update(out=self.out.row(ch).to_host(), ph=self.ph.row(ch).to_host(), idx=self.idx.row(ch).to_host())
        return res
