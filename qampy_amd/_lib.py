"""
ctypes binding of ``libqampy_hip.so`` (C ABI: include/qampy_hip.h).

There is deliberately NO fallback: if the shared library is missing, or no gfx950 device is visible when a kernel entry
point is called, the call raises.  (The CPU restatement under ``oracle/`` is test infrastructure and is never imported
from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqampy_hip.so")

QH_OK, QH_ERR_METHOD, QH_ERR_ARG, QH_ERR_HIP, QH_ERR_NODEVICE = 0, 1, 2, 3, 4

# method ids of include/qampy_hip.h (order of the QH_M_* / QH_RM_* enums)
METHOD_ID = {m: i for i, m in enumerate(("cma", "cma2", "sgncma", "mcma", "rde", "mrde", "sbd", "mddma", "dd", "sbd_data"))}
REAL_METHOD_ID = {m: i for i, m in enumerate(("cma", "sgncma", "dd", "dd_data"))}

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_pf, _pd = C.POINTER(C.c_float), C.POINTER(C.c_double)

_TRAIN = [_vp, _i, _i64, _i64, _i, _i, None, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _vp]      # [6] = mu pointer
_APPLY = [_vp, _i, _i64, _i, _vp, _i, _vp, _i, _vp]
_BPS = [_vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp]
_RECOVER = [_vp, _i, _i64, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]
_SELECT = [_vp, _i64, _i, _vp, _i64, _vp]
_DECIDE = [_vp, _i64, _vp, _i, _vp, _vp, _vp]


def _train_sig(mu_ptr, dev=False):
    sig = list(_TRAIN)
    sig[6] = _vp if dev else mu_ptr
    return sig + ([_i] if dev else [])


SIGNATURES = {
    "qh_device_count": [C.POINTER(_i)],
    "qh_init": [_i],
    "qh_device_name": [C.c_char_p, _sz],
    "qh_sync": [],
    "qh_malloc": [C.POINTER(_vp), _sz],
    "qh_free": [_vp],
    "qh_memset": [_vp, _i, _sz],
    "qh_memcpy_h2d": [_vp, _vp, _sz],
    "qh_memcpy_d2h": [_vp, _vp, _sz],
    "qh_memcpy_d2d": [_vp, _vp, _sz],
    "qh_event_create": [C.POINTER(_vp)],
    "qh_event_destroy": [_vp],
    "qh_event_record": [_vp],
    "qh_event_elapsed_ms": [_vp, _vp, _pf],
    "qh_train_equaliser_c64": _train_sig(_pf),
    "qh_train_equaliser_c128": _train_sig(_pd),
    "qh_train_equaliser_c64_dev": _train_sig(_pf, dev=True),
    "qh_train_equaliser_c128_dev": _train_sig(_pd, dev=True),
    "qh_gram_build_c64_dev": [_vp, _i, _i64, _i, _i, _i64, C.POINTER(_vp)],
    "qh_gram_build_c128_dev": [_vp, _i, _i64, _i, _i, _i64, C.POINTER(_vp)],
    "qh_train_equaliser_c64_gram_dev": _train_sig(_pf, dev=True) + [_vp],
    "qh_train_equaliser_c128_gram_dev": _train_sig(_pd, dev=True) + [_vp],
    "qh_train_equaliser_windows_c64": [_vp, _i, _i64, _vp, _i, _i64, _i64, _i, _i, C.c_float, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _vp, _vp, _vp],
    "qh_train_equaliser_windows_c128": [_vp, _i, _i64, _vp, _i, _i64, _i64, _i, _i, C.c_double, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _vp, _vp, _vp],
    "qh_train_equaliser_windows_search_c64": [_vp, _i, _i64, _vp, _i, _i64, _i64, _i, _i, C.c_float, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _vp, _vp, _vp],
    "qh_train_equaliser_windows_search_c128": [_vp, _i, _i64, _vp, _i, _i64, _i64, _i, _i, C.c_double, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _vp, _vp, _vp],
    "qh_train_equaliser_real_f32": _train_sig(_pf),
    "qh_train_equaliser_real_f64": _train_sig(_pd),
    "qh_apply_filter_c64": _APPLY, "qh_apply_filter_c128": _APPLY, "qh_apply_filter_f32": _APPLY, "qh_apply_filter_f64": _APPLY,
    "qh_apply_filter_c64_dev": _APPLY, "qh_apply_filter_c128_dev": _APPLY,
    "qh_bps_c64": _BPS, "qh_bps_c128": _BPS, "qh_bps_c64_dev": _BPS, "qh_bps_c128_dev": _BPS,
    "qh_bps_recover_c64_dev": _RECOVER, "qh_bps_recover_c128_dev": _RECOVER,
    "qh_bps_recover_part_c64_dev": _RECOVER + [_i, _i], "qh_bps_recover_part_c128_dev": _RECOVER + [_i, _i],
    "qh_comp_freq_offset_c64": [_vp, _i, _i64, _vp, _i, _vp], "qh_comp_freq_offset_c128": [_vp, _i, _i64, _vp, _i, _vp],
    "qh_pilot_phase_trace_c64": [_vp, _i, _i64, _vp, _vp, _i, _vp, _vp], "qh_pilot_phase_trace_c128": [_vp, _i, _i64, _vp, _vp, _i, _vp, _vp],
    "qh_select_angles_f32": _SELECT, "qh_select_angles_f64": _SELECT,
    "qh_make_decision_c64": _DECIDE, "qh_make_decision_c128": _DECIDE,
    "qh_make_decision_c64_dev": _DECIDE, "qh_make_decision_c128_dev": _DECIDE,
    "qh_count_errors_dev": [_vp, _vp, _i64, _i64, _i64, _vp],
    "qh_gram_build_c64_batch_dev": [_vp, _i, _i, _i64, _i, _i, _i64, C.POINTER(_vp)],
    "qh_gram_build_c128_batch_dev": [_vp, _i, _i, _i64, _i, _i, _i64, C.POINTER(_vp)],
    "qh_train_equaliser_c64_batch_dev": [_vp, _i] + _train_sig(_pf, dev=True)[1:] + [_vp],
    "qh_train_equaliser_c128_batch_dev": [_vp, _i] + _train_sig(_pd, dev=True)[1:] + [_vp],
    "qh_synth_capture_c64_dev": [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, C.c_double, C.c_double, _i, C.c_double, C.c_double, _i, C.c_double,
                                 C.c_uint64],
    "qh_train_equaliser_c64_pit_dev": [_vp, _i, _i64, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp],
    "qh_train_equaliser_c128_pit_dev": [_vp, _i, _i64, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp],
    "qh_pit_auto_segments": [_i64, C.c_double, _i, _i, C.POINTER(_i)],
    "qh_pit_prepare_bytes": [_i, _i, _i64, _sz, C.POINTER(_sz)],
    "qh_pit_prepare_c64_dev": [_vp, _i, _i64, _i64, _i, _vp, _vp, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _sz],
    "qh_pit_last_timing": [_pf, _i, C.POINTER(_i), _pf],
    "qh_pit_basis_bytes": [_i, C.POINTER(_sz)],
    "qh_pit_basis_c64_dev": [_vp, _i, _i64, _i, _i, _i64, _vp, _i],
    "qh_pit_basis_c128_dev": [_vp, _i, _i64, _i, _i, _i64, _vp, _i],
    "qh_pinned_alloc": [C.POINTER(_vp), _sz],
    "qh_pinned_free": [_vp],
    "qh_memcpy_h2d_async": [_vp, _vp, _sz],
    "qh_memcpy_d2h_async": [_vp, _vp, _sz],
    "qh_stream_sync": [],
    "qh_set_trainer": [_i],
    "qh_set_form": [C.c_char_p, C.c_char_p],
    "qh_get_form": [C.c_char_p, C.POINTER(_i)],
    "qh_set_pit_timing": [_i],
    "qh_set_reserved_cus": [_i],
    "qh_set_gram_budget_gb": [C.c_double],
    "qh_get_gram_budget_gb": [C.POINTER(C.c_double)],
    "qh_set_default_tier": [_i, C.c_double],
    "qh_get_default_tier": [C.POINTER(_i), C.POINTER(C.c_double)],
    "qh_last_pit_report": [_vp],
    "qh_use_stream": [_i],
    "qh_release_scratch": [],
    "qh_thread_release": [],
    "qh_stream_wait_event": [_vp],
    "qh_stream_handle": [C.POINTER(_vp)],
    "qh_ser_c64_dev": [_vp, _i64, _vp, _i, _i64, _vp, _i, _i, _i64, _i64, _vp],
    "qh_ser_c128_dev": [_vp, _i64, _vp, _i, _i64, _vp, _i, _i, _i64, _i64, _vp],
}

PIT_MAXPASS, PIT_MAXCHUNK = 24, 32
ABI_VERSION = 10             # QH_ABI_VERSION of include/qampy_hip.h


class PitOpts(C.Structure):
    """``qh_pit_opts`` of include/qampy_hip.h (zeros = the library's defaults)."""
    _fields_ = [("segments", C.c_int32), ("max_passes", C.c_int32), ("acquire", C.c_int32), ("phase_seed", C.c_int32),
                ("tol", C.c_double), ("gear", C.c_double), ("acq_bound", C.c_double), ("acq_plateau", C.c_double),
                ("acq_chunk", C.c_int64), ("acq_max", C.c_int64), ("correction", C.c_int32), ("head_steps", C.c_int32), ("basis", C.c_void_p), ("corr_beta", C.c_double),
                ("seg_first", C.c_int32), ("seg_count", C.c_int32), ("exchange", C.c_void_p), ("exchange_user", C.c_void_p),
                ("start", C.c_int32), ("exchange_on_stream", C.c_int32), ("dev_safety", C.c_double), ("adaptive", C.c_int32), ("exact_redo_off", C.c_int32), ("head_auto_off", C.c_int32), ("acq_anneal", C.c_int32), ("mu_hint", C.c_double), ("prepared", C.c_void_p), ("on_pass0", C.c_void_p), ("on_pass0_user", C.c_void_p),
                ("on_pass", C.c_void_p), ("on_pass_user", C.c_void_p)]


#: signature of ``qh_pit_opts.on_pass0``: (user) -> None
PIT_HOOK = C.CFUNCTYPE(None, C.c_void_p)

#: signature of ``qh_pit_opts.on_pass``: (user, sweep, pass) -> None
PIT_PASS_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)

#: signature of ``qh_pit_opts.exchange``: (user, device pointer of the segments' end taps, bytes) -> 0
PIT_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class PitReport(C.Structure):
    """``qh_pit_report`` of include/qampy_hip.h."""
    _fields_ = [("segments", C.c_int32), ("passes", C.c_int32), ("converged", C.c_int32), ("acq_chunks", C.c_int32),
                ("seg_len", C.c_int64), ("acq_steps", C.c_int64),
                ("mu", C.c_double), ("mu_acq", C.c_double), ("power", C.c_double), ("tol", C.c_double),
                ("defect", C.c_double * PIT_MAXPASS), ("acq_err", C.c_double * PIT_MAXCHUNK),
                ("gain", C.c_double), ("out_power", C.c_double),
                ("acq_done", C.c_int32), ("done", C.c_int32), ("diverged", C.c_int32), ("corr_on", C.c_int32),
                ("result_change", C.c_double * PIT_MAXPASS), ("deviation", C.c_double * PIT_MAXPASS),
                ("deviation_rms", C.c_double * PIT_MAXPASS), ("deviation_taps", C.c_double * PIT_MAXPASS),
                ("deviation_taps_worst", C.c_double * PIT_MAXPASS)]

    def as_dict(self):
        return dict(segments=int(self.segments), seg_len=int(self.seg_len), passes=int(self.passes), converged=bool(self.converged), exact_form=bool(self.converged == 2),
                    tol=float(self.tol), defect=[float(d) for d in self.defect if d >= 0],
                    acquisition=dict(steps=int(self.acq_steps), chunks=int(self.acq_chunks), mu=float(self.mu_acq),
                                     diverged=bool(self.diverged), mean_sq_err=[float(v) for v in self.acq_err if v >= 0]),
                    mu=float(self.mu), power=float(self.power), gain=float(self.gain), out_power=float(self.out_power),
                    correction=bool(self.corr_on), result_change=[float(d) for d in self.result_change if d >= 0],
                    deviation=[float(d) for d, q in zip(self.deviation, self.defect) if q >= 0],
                    deviation_rms=[float(d) for d, q in zip(self.deviation_rms, self.defect) if q >= 0],
                    deviation_taps=[float(d) for d, q in zip(self.deviation_taps, self.defect) if q >= 0],
                    deviation_taps_worst=[float(d) for d, q in zip(self.deviation_taps_worst, self.defect) if q >= 0])


_lib = None


def load():
    """Load the shared library (no device is touched yet).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: build it with qampy_amd/csrc/build.sh (or __graft_entry__.build()); "
                               "qampy_amd has no CPU fallback" % LIB_PATH)
        # Hardware queues: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one
        # run one after the other.  The library keeps FOUR streams per host thread (trainer / preparation / phase search / helper), so a second
        # thread's streams land on the first one's queues: a ReceiverGroup worker ran its phase search BEHIND its training instead of beside it
        # (round 6: 933 MSym/s on a worker thread against 1204 on the main thread, same receiver).  Raised before the runtime starts unless the
        # user set it; the library does the same in a static initialiser for callers that bind it without this module.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _i
        lib.qh_last_error.argtypes = []
        lib.qh_last_error.restype = C.c_char_p
        # the structs and signatures above belong to ONE version of include/qampy_hip.h: refuse any other build of the library
        ver = lib.qh_abi_version() if hasattr(lib, "qh_abi_version") else 0
        if ver != ABI_VERSION:
            raise RuntimeError("%s has C-ABI version %d, this package binds version %d: rebuild it (qampy_amd/csrc/build.sh)" % (LIB_PATH, ver, ABI_VERSION))
        _lib = lib
        import atexit
        atexit.register(lib.qh_thread_release)       # the main thread's streams, before the runtime's own teardown
    return _lib


def check(rc):
    """Map a status code to the exception the reference raises for the same condition."""
    if rc == QH_OK:
        return
    msg = load().qh_last_error().decode("utf-8", "replace")
    if rc == QH_ERR_METHOD:
        raise ValueError("Unknown method (%s)" % msg)
    if rc == QH_ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError("libqampy_hip: %s" % msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def set_form(key, value=None):
    """Test / measurement hook (``qh_set_form``, include/qampy_hip.h): force a kernel form - ``set_form("trainer", "direct")``,
    ``set_form("pit_form", "segment")``, ``set_form("seg_lanes", 8)``, ``set_form("bps", "tile")`` ...; ``None`` / ``""`` / ``0``: automatic."""
    v = "" if value is None else str(value)
    call("qh_set_form", str(key).encode(), v.encode())


def get_form(key):
    v = _i(0)
    call("qh_get_form", str(key).encode(), C.byref(v))
    return v.value


def ptr(a):
    """Raw pointer of a C-contiguous numpy array or of a DeviceArray."""
    if isinstance(a, DeviceArray):
        return a.ptr
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data


def device_count():
    n = _i(0)
    call("qh_device_count", C.byref(n))
    return n.value


def init(device=0):
    call("qh_init", int(device))


def device_name():
    buf = C.create_string_buffer(256)
    call("qh_device_name", buf, 256)
    return buf.value.decode()


def sync():
    call("qh_sync")


def set_default_tier(tier, tol=0.):
    """Process-wide default of the trainers (``qh_set_default_tier``): ``"a"`` - the reference's exact sequential recurrence (the initial
    default) - or ``"b"`` - the same recurrence solved in parallel in time to ``tol`` (0 = the library default 1e-3; SURVEY.md 8c's complex64
    bar is 1e-4).  Honoured by the drop-in module (``hip_equalisation.train_equaliser``, i.e. the C entry points a binding of the reference's
    ``train_equaliser`` export calls) and by every mirrored host-layer function that is not given ``tier=`` explicitly."""
    t = {"a": 0, "b": 1}.get(tier.lower() if isinstance(tier, str) else tier)
    if t is None:
        raise ValueError("tier must be 'a' (exact sequential recurrence) or 'b' (parallel in time)")
    call("qh_set_default_tier", t, float(tol))


def get_default_tier():
    """``(tier, tol)`` of :func:`set_default_tier`."""
    t, tol = _i(0), C.c_double(0)
    call("qh_get_default_tier", C.byref(t), C.byref(tol))
    return ("b" if t.value else "a"), tol.value


def last_pit_report():
    """The device's report (dict) of the calling thread's most recent host-array solve through the default tier b."""
    r = PitReport()
    call("qh_last_pit_report", C.byref(r))
    return r.as_dict()


def gram_budget_gb():
    v = C.c_double(0)
    call("qh_get_gram_budget_gb", C.byref(v))
    return v.value


def suffix(dtype):
    """(ABI suffix, real numpy type, complex numpy type) of a supported dtype."""
    dtype = np.dtype(dtype)
    if dtype in (np.dtype(np.complex64), np.dtype(np.float32)):
        return "32", np.float32, np.complex64
    if dtype in (np.dtype(np.complex128), np.dtype(np.float64)):
        return "64", np.float64, np.complex128
    raise TypeError("unsupported dtype %s (the hot path is exported for float32/64 and complex64/128 only)" % dtype)


PINNED_MIN_BYTES = 1 << 20        # results below 1 MiB are not worth a pinned buffer


def _pinned_release(p):
    try:
        if _lib is not None:
            _lib.qh_pinned_free(p)
    except Exception:
        pass


def pinned_empty(shape, dtype):
    """An ndarray on pinned host memory from the library's pool (``qh_pinned_alloc``); the buffer goes back to the pool when the last view of
    the array is garbage collected.  An ordinary writable ndarray in every other respect."""
    import weakref
    dtype = np.dtype(dtype)
    shape = tuple(int(x) for x in np.atleast_1d(shape))
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    p = _vp()
    try:
        call("qh_pinned_alloc", C.byref(p), max(n, 1))
    except (RuntimeError, MemoryError):
        # pinnable host memory exhausted (a caller keeping many large results alive): an ordinary pageable array, as the reference would
        # allocate - copies into it are still correct after the stream is synchronised, only slower
        return np.empty(shape, dtype=dtype)
    buf = (C.c_ubyte * max(n, 1)).from_address(p.value)
    weakref.finalize(buf, _pinned_release, p.value)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def stream_sync():
    call("qh_stream_sync")


class DeviceArray:
    """A dense C-ordered array living in HBM (thin owner of a ``qh_malloc`` allocation)."""

    def __init__(self, shape, dtype, zero=False):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = _vp()
        call("qh_malloc", C.byref(p), self.nbytes)
        self.ptr = p.value
        self._own = True
        if zero:
            call("qh_memset", self.ptr, 0, self.nbytes)

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        out = cls(arr.shape, arr.dtype)
        call("qh_memcpy_h2d", out.ptr, arr.ctypes.data, out.nbytes)
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        call("qh_memcpy_h2d", self.ptr, arr.ctypes.data, self.nbytes)

    def to_host(self, pinned=False, wait=True):
        """Copy to the host.  ``pinned``: into a pooled pinned buffer (:func:`pinned_empty`) - one DMA at the link rate; ``wait=False`` (with
        ``pinned``): only enqueue the copy on the current library stream - the array is complete after ``_lib.sync()`` / ``stream_sync()``."""
        if pinned and self.nbytes >= PINNED_MIN_BYTES:
            out = pinned_empty(self.shape, self.dtype)
            call("qh_memcpy_d2h_async", out.ctypes.data, self.ptr, self.nbytes)
            if wait:
                call("qh_stream_sync")
            return out
        out = np.empty(self.shape, dtype=self.dtype)
        call("qh_memcpy_d2h", out.ctypes.data, self.ptr, self.nbytes)
        return out

    def zero(self):
        call("qh_memset", self.ptr, 0, self.nbytes)

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        call("qh_memcpy_d2d", self.ptr, other.ptr, self.nbytes)

    def row(self, i):
        """Non-owning view of row ``i`` of a 2-D array."""
        v = object.__new__(DeviceArray)
        v.shape = self.shape[1:]
        v.dtype = self.dtype
        v.nbytes = self.nbytes // self.shape[0]
        v.ptr = self.ptr + i * v.nbytes
        v._own = False
        return v

    def free(self):
        if getattr(self, "_own", False) and self.ptr:
            call("qh_free", self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    """HIP event on the library stream (kernel timing in bench.py)."""

    def __init__(self):
        p = _vp()
        call("qh_event_create", C.byref(p))
        self.ptr = p.value

    def record(self):
        call("qh_event_record", self.ptr)

    def elapsed_ms(self, start):
        ms = C.c_float(0)
        call("qh_event_elapsed_ms", start.ptr, self.ptr, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                call("qh_event_destroy", self.ptr)
        except Exception:
            pass
