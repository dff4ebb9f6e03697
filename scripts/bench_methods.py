#!/usr/bin/env python3
"""Per-method cost of the exact trainers: every complex error function on one device-resident capture, in each kernel form
that takes it (qh_set_form("trainer", direct | lookahead | iterative)).  Prints one JSON line; cycles assume 2.4 GHz."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import _lib as _qlib
from qampy_amd import synth, _lib
from qampy_amd._lib import DeviceArray, Event
from qampy_amd.core.equalisation import equalisation as eq, hip_equalisation as hk

M, nsym, ntaps = int(os.environ.get("BM_M", 64)), 2 ** int(os.environ.get("BM_LOG2", 18)), int(os.environ.get("BM_TAPS", 41))
_lib.init(0)
sig = synth.make_capture(M, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000, dtype=np.complex64)
E = np.ascontiguousarray(np.asarray(sig))
tr = eq._cal_training_symbol_len(2, ntaps, E.shape[1])
# converged taps first (decision-directed / radius-directed functions need them)
w0 = eq._init_taps(ntaps, 2, 2, np.complex64)
sy0 = eq._reshape_symbols(None, "cma", M, np.complex64, 2)
_, w0, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(2e-4), w0, None, False, sy0, "cma")
dE = DeviceArray.from_host(E)
res = {}
for method in os.environ.get("BM_METHODS", "cma,mcma,cma2,rde,mrde,sbd,mddma,dd").split(","):
    sy = eq._reshape_symbols(sig.coded_symbols if method in ("sbd", "mddma", "dd") else None, method, M, np.complex64, 2)
    dsy = DeviceArray.from_host(np.ascontiguousarray(sy))
    derr = DeviceArray((2, tr), np.complex64)
    dmu = DeviceArray.from_host(np.array([1e-4 if method != "cma2" else 1e-6], np.float32))
    for form in ("direct", "lookahead", "iterative", "default"):
        _qlib.set_form("trainer", None)
        if form != "default":
            _qlib.set_form("trainer", form)
        for adaptive in (False, True, "per-mode"):
            if adaptive and form not in ("default", "iterative"):
                continue
            dw = DeviceArray.from_host(w0.copy())
            hk.train_equaliser_dev(dE, tr, 1, 2, dmu, dw, None, adaptive, dsy, method, derr)      # warm-up
            dw.set(w0.copy())
            dmu.set(np.array([1e-4 if method != "cma2" else 1e-6], np.float32))
            e0, e1 = Event(), Event()
            e0.record()
            hk.train_equaliser_dev(dE, tr, 1, 2, dmu, dw, None, adaptive, dsy, method, derr)
            e1.record()
            ms = e1.elapsed_ms(e0)
            wf = dw.to_host()
            res["%s/%s%s" % (method, form, "+adaptive" if adaptive is True else ("+adaptive(per-mode)" if adaptive else ""))] = dict(ms=round(ms, 2), cycles_per_step=round(ms * 1e-3 * 2.4e9 / tr, 1),
                                                                                   finite=bool(np.all(np.isfinite(wf))))
_qlib.set_form("trainer", None)
print(json.dumps(dict(what="%d-QAM 2-pol, %d symbols, %d taps, one sweep, both output modes concurrently (gram build included)" % (M, nsym, ntaps), results=res)))
