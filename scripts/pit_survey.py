#!/usr/bin/env python3
"""Tier b at SURVEY.md 8d's own step sizes for the C3 recipe (64-QAM, 41 taps, cma -> mrde, mu = (1e-3, 5e-4), linewidth 5 kHz and 0)
on the capture lengths where that recipe converges in the reference (2^14 .. 2^18 symbols), against the exact path on the same capture:
time, segments, passes, device certificate, measured deviation of output / taps, symbol errors.  -> profiles/r03_pit_survey_steps.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber

_lib.init(0)
print("# nsym linewidth | tier a: ms errors | tier b: ms errors (S, passes, converged, est. rms deviation) per stage | measured: out rms dev, tap rel dev per mode")
for lw in (0., 5e3):
    for lg in (14, 15, 16, 17, 18):
        nsym = 2 ** lg
        d = synth.make_capture_dev(64, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=lw, seed=42)
        kw = dict(methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"])
        res = {}
        for tier in ("a", "b"):
            rx = ResidentReceiver(2, 2 * nsym, 2, 64, 41, (1e-3, 5e-4), tier=tier, **kw)
            rx.E.copy_from(d["E"])
            rx.run(); _lib.sync()
            t0 = time.perf_counter(); rx.run(); _lib.sync(); el = time.perf_counter() - t0
            r = rx.fetch()
            r["ms"] = el * 1e3
            r["errors"] = [s["errors"] for s in ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, min(8192, nsym // 4), 1000)]
            r["rep"] = rx.pit_reports()
            res[tier] = r
            del rx
        a, b = res["a"], res["b"]
        dev = []
        for m in range(2):
            g = 1j ** int(np.rint(np.angle(np.vdot(b["wxy"][m].ravel(), a["wxy"][m].ravel())) / (np.pi / 2)))
            dev.append((float(np.sqrt(np.mean(np.abs(a["eq"][m] - g * b["eq"][m]) ** 2))), float(np.linalg.norm(a["wxy"][m] - g * b["wxy"][m]) / np.linalg.norm(a["wxy"][m]))))
        print("2^%d %5.0f Hz | a: %8.2f ms %s | b: %7.3f ms %s %s | %s" % (
            lg, lw, a["ms"], a["errors"], b["ms"], b["errors"],
            [(st["segments"], st["passes"], "exact_form" if st.get("exact_form") else st["converged"], round(st["deviation_rms"][-1], 5) if st["deviation_rms"] else None) for st in b["rep"]],
            [(round(x, 5), round(y, 5)) for x, y in dev]), flush=True)
