#!/usr/bin/env python3
"""Parallel-in-time training (tier B+) against the exact path on the C3 capture: time, SER, tap / output deviation per (S, P)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber

wl = dict(M=64, nsym=int(os.environ.get("PIT_NSYM", 2 ** 22)), ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4))
_lib.init(0)
d = synth.make_capture_dev(wl["M"], wl["nsym"], nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=int(os.environ.get("PIT_SEED", 1000)))
kw = dict(methods=wl["methods"], Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"])


def run(S, P, prefix=(0, 1 << 17)):
    rx = ResidentReceiver(2, wl["nsym"] * 2, 2, wl["M"], wl["ntaps"], wl["mu"], segments=S, passes=P, prefix=prefix, **kw)
    rx.E.copy_from(d["E"])
    rx.run(); _lib.sync()
    t0 = time.perf_counter(); rx.run(); _lib.sync(); t = time.perf_counter() - t0
    rx.report_passes = True
    rx.run(); _lib.sync()
    ser = ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)
    return rx, t, ser


ref, t_ref, ser_ref = run(0, 0)
w_ref, out_ref, eq_ref = ref.wxy.to_host(), ref.out.to_host(), ref.eq.to_host()
e_ref = [e.to_host() for e in ref.err]
res = [dict(S=0, P=0, ms=round(t_ref * 1e3, 2), MSym_s=round(wl["nsym"] / t_ref / 1e6, 2), errors=[r["errors"] for r in ser_ref])]
for S, P, pre in [((0, 64), 1, (0, 1 << 17)), ((0, 64), 2, (0, 1 << 17)), ((0, 64), 3, (0, 1 << 17)), ((0, 64), 2, (0, 1 << 16)), ((0, 64), 2, (0, 0)),
                  ((0, 128), 2, (0, 1 << 17)), ((0, 32), 2, (0, 1 << 17)), ((64, 64), 3, (1 << 18, 1 << 17))]:
    rx, t, ser = run(S, P, pre)
    w, eq = rx.wxy.to_host(), rx.eq.to_host()
    e2 = rx.err[1].to_host()
    res.append(dict(S=S, P=P, prefix=pre, ms=round(t * 1e3, 2), MSym_s=round(wl["nsym"] / t / 1e6, 2), errors=[r["errors"] for r in ser],
                    tap_diff_max=float(np.max(np.abs(w - w_ref))), tap_diff_rel=float(np.linalg.norm(w - w_ref) / np.linalg.norm(w_ref)),
                    eq_rms_diff=float(np.sqrt(np.mean(np.abs(eq - eq_ref) ** 2))), eq_max_diff=float(np.max(np.abs(eq - eq_ref))),
                    err2_rms_diff=float(np.sqrt(np.mean(np.abs(e2 - e_ref[1]) ** 2))),
                    pass_change=[[float("%.3g" % v) for v in pc] for pc in rx.pass_change]))
    del rx
print(json.dumps(dict(what="C3 capture, exact vs parallel-in-time (S segments, P passes)", results=res)))
