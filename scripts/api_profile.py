#!/usr/bin/env python3
"""Where the time of the drop-in call chain goes at C3 (numpy in -> dual_mode_equalisation -> bps -> numpy out, default tier b): cProfile of one call."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qampy_amd
from qampy_amd import synth, _lib, equalisation as api_eq, phaserec as api_ph
from qampy_amd.signals import SignalQAM
_lib.init(0)
nsym = int(os.environ.get("NSYM", 2 ** 22))
tol = float(os.environ.get("TOL", 1e-4))
d = synth.make_capture_dev(64, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000)
sig = SignalQAM(d["E"].to_host(), 64, fb=d["fb"], fs=d["fs"], symbols=d["symbols"].to_host(), coded_symbols=d["alphabet_host"])
qampy_amd.set_default_tier("b", tol)


def once():
    out, wxy, errs = api_eq.dual_mode_equalisation(sig, (2e-4, 2e-4), 41, methods=("cma", "mrde"))
    rec, ph = api_ph.bps(out, 64, 20)
    return out, wxy, errs, rec, ph


for _ in range(3):
    t0 = time.perf_counter(); r = once(); print("%.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable(); r = once(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
