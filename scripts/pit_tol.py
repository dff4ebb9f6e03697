#!/usr/bin/env python3
"""Tier b against the exact path on the same capture at several tolerances and recipes (round 5: what does the tolerance of SURVEY.md 8c cost,
and what does SURVEY.md 8d's literal recipe do at full size): time of one capture handed over and waited for, segments, passes, the device's
estimates per pass, measured deviation of equaliser output / taps / error traces, symbol errors.

    python scripts/pit_tol.py [shape ...]          shapes: c3 ns c2 c3s5k c3s1k c3s0     (default: c3 c3s5k c3s1k)
    TOLS=1e-3,3e-4,1e-4   NSYM=<override>
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber

SHAPES = {
    "c3": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    "ns": dict(M=64, nsym=10 ** 7, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    "c2": dict(M=16, nsym=2 ** 20, ntaps=21, methods=("mcma",), mu=(1e-3,), A=32, snr=25, lw=50e3),
    "c3s5k": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(1e-3, 5e-4), A=64, snr=30, lw=5e3),
    "c3s1k": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(1e-3, 5e-4), A=64, snr=30, lw=1e3),
    "c3h1k": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(5e-4, 2.5e-4), A=64, snr=30, lw=1e3),
    "c3h0": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(5e-4, 2.5e-4), A=64, snr=30, lw=0.),
    "c3s18": dict(M=64, nsym=2 ** 18, ntaps=41, methods=("cma", "mrde"), mu=(1e-3, 5e-4), A=64, snr=30, lw=5e3),
    "c3s0": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(1e-3, 5e-4), A=64, snr=30, lw=0.),
}
_lib.init(0)
tols = [float(t) for t in os.environ.get("TOLS", "1e-3,3e-4,1e-4").split(",")]
shapes = sys.argv[1:] or ["c3", "c3s5k", "c3s1k"]
for key in shapes:
    c = dict(SHAPES[key])
    nsym = int(os.environ.get("NSYM", c["nsym"]))
    d = synth.make_capture_dev(c["M"], nsym, nmodes=2, snr_db=c["snr"], theta=np.pi / 5.6, dgd=30e-12, linewidth=c["lw"], seed=1000)
    kw = dict(methods=c["methods"], Niter=(1,) * len(c["methods"]), Mtestangles=c["A"], Nbps=20, alphabet=d["alphabet_host"])

    def go(tier, pit=None, reps=3):
        rx = ResidentReceiver(2, 2 * nsym, 2, c["M"], c["ntaps"], c["mu"], tier=tier, pit=pit, **kw)
        rx.E.copy_from(d["E"])
        rx.run(); _lib.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            rx.run(); _lib.sync()
        el = (time.perf_counter() - t0) / reps
        r = rx.fetch()
        r["ms"] = el * 1e3
        r["errors"] = [s["errors"] for s in ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)]
        r["rep"] = rx.pit_reports()
        return r
    a = go("a", reps=1)
    print("== %s nsym %d mu %s linewidth %g | exact: %.1f ms errors %s" % (key, nsym, c["mu"], c["lw"], a["ms"], a["errors"]), flush=True)
    for tol in tols:
        b = go("b", pit=dict(tol=tol))
        devs = []
        for m in range(2):
            g = 1j ** int(np.rint(np.angle(np.vdot(b["wxy"][m].ravel(), a["wxy"][m].ravel())) / (np.pi / 2)))
            eqd = float(np.sqrt(np.mean(np.abs(a["eq"][m] - g * b["eq"][m]) ** 2) / np.mean(np.abs(a["eq"][m]) ** 2)))
            tapd = float(np.linalg.norm(a["wxy"][m] - g * b["wxy"][m]) / np.linalg.norm(a["wxy"][m]))
            errd = [float(np.sqrt(np.mean(np.abs(a["err"][s][m] - g * b["err"][s][m]) ** 2))) for s in range(len(c["methods"]))]
            flip = float(np.mean(((a["idx"][m] - b["idx"][m]) % c["A"]) != 0)) if g == 1 else -1.
            devs.append("m%d eq %.2e taps %.2e err %s flip %.1e" % (m, eqd, tapd, ["%.2e" % e for e in errd], flip))
        print("  tol %g: %.3f ms = %.0f MSym/s errors %s" % (tol, b["ms"], nsym / b["ms"] / 1e3, b["errors"]))
        for st in b["rep"]:
            print("     S %d x %d P %d conv %s exact_form %s acq %s | est rms %s | taps %s | taps worst %s" % (
                st["segments"], st["seg_len"], st["passes"], st["converged"], st.get("exact_form"), st["acquisition"]["steps"],
                ["%.2g" % v for v in st["deviation_rms"]], ["%.2g" % v for v in st.get("deviation_taps", [])], ["%.2g" % v for v in st.get("deviation_taps_worst", [])]))
        for s_ in devs:
            print("     " + s_)
        sys.stdout.flush()
    del a, d
    _lib.call("qh_release_scratch")
