#!/usr/bin/env python3
"""Tier b with the adaptive step against the exact path: the reference script's recipe (Scripts/64_qam_equalisation.py: 64-QAM, 13 taps,
mu = 1.9e-3, mcma -> mddma, adaptive_stepsize=(True, True)), one stage at a time on the same capture.

    python scripts/adapt_exp.py [--log2 17] [--method mcma] [--mu 1.9e-3] [--seed 1000] [--pit key=value,...]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd._lib import DeviceArray
from qampy_amd.core.equalisation import equalisation as eq, hip_equalisation as hk

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=17)
ap.add_argument("--methods", default="mcma,mddma")
ap.add_argument("--mu", type=float, default=1.9e-3)
ap.add_argument("--ntaps", type=int, default=13)
ap.add_argument("--M", type=int, default=64)
ap.add_argument("--snr", type=float, default=25.)
ap.add_argument("--seed", type=int, default=1000)
ap.add_argument("--pit", default="")
ap.add_argument("--modes", default="", help="output modes to train (default: all, in turn)")
args = ap.parse_args()
_lib.init(0)
nsym = 2 ** args.log2
sig = synth.make_capture(args.M, nsym, nmodes=2, snr_db=args.snr, theta=np.pi / 3, dgd=30e-12, linewidth=0., seed=args.seed, dtype=np.complex64)
E = np.ascontiguousarray(np.asarray(sig))
tr = eq._cal_training_symbol_len(2, args.ntaps, E.shape[1])
pit = {}
for kv in [x for x in args.pit.split(",") if x]:
    k, v = kv.split("=")
    pit[k] = float(v) if "." in v or "e" in v else int(v)
dE = DeviceArray.from_host(E)
out = []
w_a = eq._init_taps(args.ntaps, 2, 2, np.complex64)
w_b = w_a.copy()
for stage, method in enumerate(args.methods.split(",")):
    sy = eq._reshape_symbols(sig.coded_symbols if method in ("sbd", "mddma", "dd") else None, method, args.M, np.complex64, 2)
    dsy = DeviceArray.from_host(np.ascontiguousarray(sy))
    res = {}
    for tier, w0 in (("a", w_a), ("b", w_b)):
        dw = DeviceArray.from_host(w0.copy())
        derr = DeviceArray((2, tr), np.complex64)
        dmu = DeviceArray.from_host(np.array([args.mu], np.float32))
        rep = hk.PitReportBuffer() if tier == "b" else None
        kw = dict(pit=dict(pit), report=rep) if tier == "b" else {}
        for rpt in range(2):                                     # second run timed
            dw.set(w0.copy()); dmu.set(np.array([args.mu], np.float32))
            _lib.sync(); t0 = time.perf_counter()
            hk.train_equaliser_dev(dE, tr, 1, 2, dmu, dw, [int(x) for x in args.modes.split(',')] if args.modes else None, True, dsy, method, derr, zero_err=True, **kw)
            _lib.sync(); dt = time.perf_counter() - t0
        res[tier] = dict(w=dw.to_host(), err=derr.to_host(), mu=float(dmu.to_host()[0]), ms=dt * 1e3, report=rep.read() if rep is not None else None)
    a, b = res["a"], res["b"]
    tap = [float(np.linalg.norm(a["w"][m] - b["w"][m]) / np.linalg.norm(a["w"][m])) for m in range(2)]
    er = [float(np.sqrt(np.mean(np.abs(a["err"][m] - b["err"][m]) ** 2))) for m in range(2)]
    r = b["report"]
    rec = dict(stage=method, nsym=nsym, ms_exact=round(a["ms"], 2), ms_tier_b=round(b["ms"], 2), mu_exact=a["mu"], mu_tier_b=b["mu"], mu_rel_dev=abs(a["mu"] - b["mu"]) / a["mu"],
               tap_rel_dev=tap, err_trace_rms_dev=er, finite=bool(np.all(np.isfinite(b["w"]))),
               last_mode_report=dict(segments=r["segments"], seg_len=r["seg_len"], passes=r["passes"], converged=r["converged"],
                                     defect=[float("%.3g" % d) for d in r["defect"]], result_change=[float("%.3g" % d) for d in r.get("result_change", [])], deviation_rms=[float("%.3g" % d) for d in r["deviation_rms"]],
                                     deviation_taps=[float("%.3g" % d) for d in r["deviation_taps"]]) if r else None)
    out.append(rec)
    print("##", json.dumps(rec), flush=True)
    w_a, w_b = a["w"], a["w"]                                    # both tiers start the next stage from the exact taps: stage-by-stage comparison
print(json.dumps(dict(what="adaptive step: tier b against the exact path", results=out)))
