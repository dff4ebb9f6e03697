#!/usr/bin/env python3
"""DEBUG: sensitivity of the exact sbd recurrence to its start taps on the first segments of the 64-QAM mcma -> sbd recipe."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.core.equalisation import equalisation as eq, hip_equalisation as hk
_lib.init(0)
M, nsym, ntaps = 64, 2 ** 19, 41
sig = synth.make_capture(M, nsym, nmodes=2, snr_db=28, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000, dtype=np.complex64)
E = np.ascontiguousarray(np.asarray(sig))
tr = eq._cal_training_symbol_len(2, ntaps, E.shape[1])
w0 = eq._init_taps(ntaps, 2, 2, np.complex64)
sy1 = eq._reshape_symbols(None, "mcma", M, np.complex64, 2)
sy2 = eq._reshape_symbols(sig.coded_symbols, "sbd", M, np.complex64, 2)
_, w1, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(3e-4), w0.copy(), None, False, sy1, "mcma")
rng = np.random.default_rng(1)
for method, sy, mu in (("sbd", sy2, 1e-4), ("mrde", eq._reshape_symbols(None, "mrde", M, np.complex64, 2), 1e-4), ("dd", eq._reshape_symbols(sig.coded_symbols, "dd", M, np.complex64, 2), 1e-4)):
    for eps in (1e-3, 1e-2):
        d = (rng.standard_normal(w1.shape) + 1j * rng.standard_normal(w1.shape)).astype(np.complex64)
        d *= eps * np.linalg.norm(w1) / np.linalg.norm(d)
        row = []
        for n in (4032, 8064, 16128, 32256, 64512, 129024):
            ea, wa, _ = hk.train_equaliser(E, n, 1, 2, np.float32(mu), w1.copy(), None, False, sy, method)
            eb, wb, _ = hk.train_equaliser(E, n, 1, 2, np.float32(mu), (w1 + d).astype(np.complex64), None, False, sy, method)
            row.append((n, [round(float(np.linalg.norm(wa[m] - wb[m]) / np.linalg.norm(wa[m])), 5) for m in range(2)],
                        [round(float(np.mean(np.abs(ea[m, -2000:]) ** 2)), 5) for m in range(2)]))
        print(method, "eps", eps, row, flush=True)
