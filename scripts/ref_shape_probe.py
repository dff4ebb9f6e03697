#!/usr/bin/env python3
"""The reference's benchmark shape (test/test_benchmarks.py: QPSK, 10^5 symbols, 2 pol, 40 taps, mu 4e-4, adaptive step) through tier a and tier b:
where the time of the tier-b call goes (report of every mode's solve)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib, equalisation as api_eq
from qampy_amd.core.equalisation import equalisation as host
_lib.init(0)
for dt in (np.complex64,):
    sig = synth.make_capture(4, 10 ** 5, nmodes=2, os=2, snr_db=14, theta=np.pi / 5.45, dgd=75e-12, linewidth=0., fb=40e9, beta=0.1, seed=7, dtype=dt)
    for method in ("cma", "mcma"):
        for adaptive in (True, False):
            for tier, pit in (("a", None), ("b", dict(tol=1e-4)), ("b", dict(tol=1e-3))):
                kw = dict(tier=tier)
                if pit: kw["pit"] = pit
                best = 1e9
                for _ in range(4):
                    t0 = time.perf_counter(); w, e = api_eq.equalise_signal(sig, 4e-4, Ntaps=40, method=method, adaptive_stepsize=adaptive, **kw); best = min(best, time.perf_counter() - t0)
                rep = host.last_pit_reports() if tier == "b" else None
                print("%s %s adaptive %s tier %s %s: %.2f ms" % (np.dtype(dt).name, method, adaptive, tier, pit, best * 1e3), flush=True)
                if rep:
                    for r in rep:
                        print("    ", {k: r[k] for k in ("segments", "seg_len", "passes", "converged", "exact_form") if k in r}, "head", r.get("head_steps"), "acq", r.get("acquisition"),
                              "est", ["%.2g" % v for v in r.get("deviation_rms", [])])
