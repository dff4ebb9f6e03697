#!/usr/bin/env python3
"""SURVEY.md 8d's literal C3 recipe (64-QAM, 41 taps, cma -> mrde, mu = (1e-3, 5e-4), 64-angle search) through the EXACT path at several capture
lengths, seeds and linewidths: symbol errors per mode.  Where does the recipe itself (the reference's recurrence) converge?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber
_lib.init(0)
mus = [tuple(float(x) for x in m.split(",")) for m in os.environ.get("MUS", "1e-3,5e-4").split(";")]
for mu in mus:
    for lg in (16, 18, 20, 22):
        for lw in (0., 1e3, 5e3):
            row = []
            for seed in (1000, 1001, 42):
                nsym = 2 ** lg
                d = synth.make_capture_dev(64, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=lw, seed=seed)
                rx = ResidentReceiver(2, 2 * nsym, 2, 64, 41, mu, methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"], tier="a")
                rx.E.copy_from(d["E"]); rx.run(); _lib.sync()
                row.append([s["errors"] for s in ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, min(8192, nsym // 4), 2000)])
                del rx, d
            print("mu %s 2^%d lw %5.0f Hz: errors per mode for seeds 1000, 1001, 42: %s" % (mu, lg, lw, row), flush=True)
