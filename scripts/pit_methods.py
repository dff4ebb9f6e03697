#!/usr/bin/env python3
"""Tier b against tier a on other method pairs / alphabets than the bench's (parallel-in-time robustness survey)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber

_lib.init(0)
CASES = [
    dict(name="16qam mcma->sbd 21 taps", M=16, nsym=2 ** 21, ntaps=21, methods=("mcma", "sbd"), mu=(1e-3, 2e-4), snr=22, lw=50e3, A=32),
    dict(name="16qam mcma->mddma 21 taps", M=16, nsym=2 ** 21, ntaps=21, methods=("mcma", "mddma"), mu=(1e-3, 2e-4), snr=22, lw=50e3, A=32),
    dict(name="64qam mcma->sbd 41 taps", M=64, nsym=2 ** 21, ntaps=41, methods=("mcma", "sbd"), mu=(3e-4, 1e-4), snr=28, lw=100., A=64),
    dict(name="64qam cma->rde 41 taps", M=64, nsym=2 ** 21, ntaps=41, methods=("cma", "rde"), mu=(2e-4, 2e-4), snr=28, lw=100., A=64),
    dict(name="qpsk cma 11 taps", M=4, nsym=2 ** 21, ntaps=11, methods=("cma",), mu=(1e-3,), snr=14, lw=100e3, A=32),
    dict(name="256qam cma->mrde 41 taps", M=256, nsym=2 ** 21, ntaps=41, methods=("cma", "mrde"), mu=(1e-4, 1e-4), snr=36, lw=100., A=64),
    dict(name="64qam cma->dd 17 taps", M=64, nsym=2 ** 21, ntaps=17, methods=("cma", "dd"), mu=(2e-4, 1e-4), snr=30, lw=100., A=64),
]
PIT2 = json.loads(os.environ.get("PIT2", "{}"))          # extra tier-b options of the LAST stage, e.g. PIT2='{"corr_beta": 0}'
ONLY = os.environ.get("ONLY")
if os.environ.get("LW"):
    for c in CASES:
        c["lw"] = float(os.environ["LW"])
for c in CASES:
    if ONLY and ONLY not in c["name"]:
        continue
    d = synth.make_capture_dev(c["M"], c["nsym"], nmodes=2, snr_db=c["snr"], theta=np.pi / 5.6, dgd=30e-12, linewidth=c["lw"], seed=1000)
    kw = dict(methods=c["methods"], Niter=(1,) * len(c["methods"]), Mtestangles=c["A"], Nbps=20, alphabet=d["alphabet_host"])
    res = {}
    for tier in ("a", "b"):
        try:
            pit = [dict() for _ in c["methods"]]
            for p_ in pit:
                p_.update(json.loads(os.environ.get("PITALL", "{}")))
            pit[-1].update(PIT2)
            pit[0].update(json.loads(os.environ.get("PIT1", "{}")))
            rx = ResidentReceiver(2, 2 * c["nsym"], 2, c["M"], c["ntaps"], c["mu"], tier=tier, pit=pit if tier == "b" else None, **kw)
            rx.E.copy_from(d["E"])
            rx.run(); _lib.sync()
            t0 = time.perf_counter(); rx.run(); _lib.sync(); el = time.perf_counter() - t0
            ser = ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)
            res[tier] = dict(ms=round(el * 1e3, 2), errors=[s["errors"] for s in ser], rep=rx.pit_reports(), eq=rx.eq.to_host(), w=rx.wxy.to_host())
            del rx
        except Exception as e:
            res[tier] = dict(error=repr(e)[:200])
    b = res["b"]
    dev = []
    if "eq" in res["a"] and "eq" in b:
        for m in range(2):
            g = 1j ** int(np.rint(np.angle(np.vdot(b["w"][m].ravel(), res["a"]["w"][m].ravel())) / (np.pi / 2)))
            dev.append(round(float(np.sqrt(np.mean(np.abs(res["a"]["eq"][m] - g * b["eq"][m]) ** 2))), 5))
    print("##", c["name"], "| a:", res["a"].get("ms"), res["a"].get("errors"), "| b:", b.get("ms"), b.get("errors"), b.get("error"),
          [(r["segments"], r["passes"], "exact_form" if r.get("exact_form") else r["converged"], [round(x, 5) for x in (r["deviation_rms"] if os.environ.get("FULL") else r["deviation_rms"][-3:])], round(r["gain"], 3)) for r in (b.get("rep") or [])], "| measured out rms dev", dev, flush=True)
