#!/usr/bin/env python3
"""Decision-directed training on NON-square alphabets (32- / 128-QAM crosses): the block-iterative form with the alphabet scan (round 5,
train_bi.h bi_nearest_general) against the direct form on the same capture - deviation of taps / error trace and time per sweep.

    python scripts/cross_qam_probe.py [nsym]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import _lib as _qlib
from qampy_amd import synth, _lib
from qampy_amd.core.equalisation import hip_equalisation as hk

_lib.init(0)
nsym = int((sys.argv[1:] or [2 ** 18])[0])
for M in (32, 128, 64):
    d = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=32, theta=np.pi / 7, dgd=20e-12, linewidth=0., seed=7)
    E0 = d["E"].to_host()
    sym = np.tile(d["alphabet_host"][None, :], (2, 1))
    ref128 = {}
    for dt in (np.complex128, np.complex64):
        E = np.ascontiguousarray(E0.astype(dt))
        rt = np.float32 if dt == np.complex64 else np.float64
        nt = 21
        wx0 = np.zeros((2, 2, nt), dt); wx0[0, 0, nt // 2] = 1; wx0[1, 1, nt // 2] = 1
        # blind pre-convergence so that the decisions mean something
        pre = wx0.copy()
        hk.train_equaliser(E, nsym - nt, 1, 2, rt(2e-3), pre, np.arange(2), 0, sym[:, :1].astype(dt) * 0 + dt(np.mean(np.abs(sym[0]) ** 4) / np.mean(np.abs(sym[0]) ** 2)), "cma")
        for method in ("sbd", "mddma", "dd"):
            for adaptive in (0, 1):
                res = {}
                for form in ("direct", "auto"):
                    if form == "direct":
                        _qlib.set_form("trainer", "direct")
                    else:
                        _qlib.set_form("trainer", None)
                    best = 1e9
                    for rep in range(2):
                        wx = pre.copy()
                        _lib.sync(); t0 = time.perf_counter()
                        err, _, mu = hk.train_equaliser(E, nsym - nt, 1, 2, rt(5e-4), wx, np.arange(2), adaptive, sym.astype(dt), method)
                        _lib.sync(); best = min(best, time.perf_counter() - t0)
                    res[form] = (wx, err, mu, best)
                a, b = res["direct"], res["auto"]
                if dt == np.complex128:
                    ref128[(method, adaptive)] = a
                else:       # how far single precision alone moves the trajectory (decisions flip where y sits on a boundary)
                    r = ref128[(method, adaptive)]
                    print("      against complex128 direct: direct taps %.2e err %.2e | auto taps %.2e err %.2e" % (
                        np.linalg.norm(a[0] - r[0]) / np.linalg.norm(r[0]), np.sqrt(np.mean(np.abs(a[1] - r[1]) ** 2)),
                        np.linalg.norm(b[0] - r[0]) / np.linalg.norm(r[0]), np.sqrt(np.mean(np.abs(b[1] - r[1]) ** 2))))
                print("M %3d %s %-5s adaptive %d: taps dev %.2e err dev %.2e (rms err %.3f) mu %g / %g | direct %.2f ms, auto %.2f ms" % (
                    M, dt.__name__, method, adaptive, np.linalg.norm(a[0] - b[0]) / np.linalg.norm(a[0]), np.sqrt(np.mean(np.abs(a[1] - b[1]) ** 2)),
                    np.sqrt(np.mean(np.abs(a[1]) ** 2)), a[2], b[2], a[3] * 1e3, b[3] * 1e3), flush=True)
