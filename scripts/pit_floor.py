#!/usr/bin/env python3
"""Where does the tap deviation of tier b from the exact path floor, and why (round 5)?

  (a) the exact path in complex64 against the exact path in complex128 on the same capture: what single-precision rounding alone does to
      taps / equaliser output / error traces over a sweep of this length (no implementation in complex64 can be closer to the reference
      than this);
  (b) tier b against the exact path (both complex64) stage by stage at several tolerances, stage 2 also from IDENTICAL start taps, and the
      deviation split by tap direction: well excited (eigenvalue of the window covariance above 10 % of the largest) / weakly excited,
      and the part of it that is a common phase of the tap set.

    python scripts/pit_floor.py [c3|ns|c2]    TOLS=1e-3,1e-4,1e-5
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver

SHAPES = {
    "c3": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    "ns": dict(M=64, nsym=10 ** 7, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    "c2": dict(M=16, nsym=2 ** 20, ntaps=21, methods=("mcma",), mu=(1e-3,), A=32, snr=25, lw=50e3),
}
_lib.init(0)
key = (sys.argv[1:] or ["c3"])[0]
c = SHAPES[key]
nsym = int(os.environ.get("NSYM", c["nsym"]))
tols = [float(t) for t in os.environ.get("TOLS", "1e-3,1e-4,1e-5").split(",")]
d = synth.make_capture_dev(c["M"], nsym, nmodes=2, snr_db=c["snr"], theta=np.pi / 5.6, dgd=30e-12, linewidth=c["lw"], seed=1000)
E = d["E"].to_host()
ns = len(c["methods"])
kw = dict(methods=c["methods"], Niter=(1,) * ns, Mtestangles=None, alphabet=d["alphabet_host"])

# eigenbasis of the window covariance (host, double): rows of the 2 x ntaps stacked window
nt = c["ntaps"]
idx = (np.arange(8192) * (nsym // 8192))[:, None] * 2 + np.arange(nt)[None, :]
Xw = np.concatenate([E[0][idx], E[1][idx]], axis=1).astype(np.complex128)          # windows x (2 nt)
Rc = Xw.conj().T @ Xw / Xw.shape[0]
lam, V = np.linalg.eigh(Rc)
strong = lam > 0.1 * lam.max()
print("# %s nsym %d: eigenvalues %.3g .. %.3g, %d of %d directions well excited" % (key, nsym, lam.min(), lam.max(), strong.sum(), lam.size))


def split(wa, wb):
    """relative norm of wa - wb per output mode: total, well excited part, weakly excited part, after removing the best common phase"""
    out = []
    for m in range(wa.shape[0]):
        a, b = wa[m].ravel().astype(np.complex128), wb[m].ravel().astype(np.complex128)
        dv = V.conj().T @ (a - b)                 # (taps act as w.x: the update direction is conj(x), so project on the eigenvectors of Rc)
        na = np.linalg.norm(a)
        ph = np.angle(np.vdot(b, a))
        out.append("m%d tot %.2e strong %.2e weak %.2e | common phase %.2e rad, without it %.2e" % (
            m, np.linalg.norm(a - b) / na, np.linalg.norm(dv[strong]) / na, np.linalg.norm(dv[~strong]) / na, ph, np.linalg.norm(a - np.exp(1j * ph) * b) / na))
    return out


def stages(tier, dtype, pit=None, start2=None):
    """run stage by stage; returns [taps after stage s], [err traces], eq"""
    rx = ResidentReceiver(2, 2 * nsym, 2, c["M"], nt, c["mu"], tier=tier, pit=pit, dtype=dtype, **kw)
    rx.load(E.astype(dtype))
    taps, errs = [], []
    rx.reset(); rx.build_gram()
    for s in range(ns):
        if s == 1 and start2 is not None:
            rx.wxy.set(np.ascontiguousarray(start2.astype(dtype)))
        t0 = time.perf_counter()
        rx.train(s); _lib.sync()
        ms = (time.perf_counter() - t0) * 1e3
        taps.append(rx.wxy.to_host()); errs.append(rx.err[s].to_host())
    rx._apply(); _lib.sync()
    eq = rx.eq.to_host()
    rep = rx.pit_reports()
    if tier == "b" and getattr(rx, "_basis", None) is not None and not getattr(stages, "said", False):
        stages.said = True
        raw = rx._basis.to_host()
        n = 2 * nt
        lamd = raw[:8 * n].view(np.float64)
        Vd = raw[8 * n:8 * n + 8 * n * n].view(np.complex64).reshape(n, n).astype(np.complex128)
        print("  device basis: eigenvalues %.3g .. %.3g; |V^H V - I| max %.2e, fro %.2e; |V V^H - I| max %.2e" % (
            lamd.min(), lamd.max(), np.abs(Vd.conj().T @ Vd - np.eye(n)).max(), np.linalg.norm(Vd.conj().T @ Vd - np.eye(n)), np.abs(Vd @ Vd.conj().T - np.eye(n)).max()))
    del rx
    return taps, errs, eq, rep


def rms(x):
    return float(np.sqrt(np.mean(np.abs(x) ** 2)))


a64 = stages("a", np.complex64)
a128 = stages("a", np.complex128)
print("== (a) exact path, complex64 against complex128")
for s in range(ns):
    print("  stage %d taps: %s" % (s, "; ".join(split(a128[0][s], a64[0][s]))))
    print("  stage %d error trace rms dev: %s" % (s, ["%.2e" % rms(a128[1][s][m] - a64[1][s][m]) for m in range(2)]))
print("  equaliser output rel rms dev: %s" % ["%.2e" % (rms(a128[2][m] - a64[2][m]) / rms(a128[2][m])) for m in range(2)])
del a128
for tol in tols:
    b = stages("b", np.complex64, pit=dict(tol=tol))
    print("== (b) tier b tol %g against the exact path (complex64): passes %s" % (tol, [r["passes"] for r in b[3]]))
    for s in range(ns):
        print("  stage %d taps: %s" % (s, "; ".join(split(a64[0][s], b[0][s]))))
        print("  stage %d error trace rms dev: %s   est taps worst %s" % (s, ["%.2e" % rms(a64[1][s][m] - b[1][s][m]) for m in range(2)], ["%.2g" % v for v in b[3][s]["deviation_taps_worst"]]))
    print("  equaliser output rel rms dev: %s" % ["%.2e" % (rms(a64[2][m] - b[2][m]) / rms(a64[2][m])) for m in range(2)])
    if ns > 1:
        b2 = stages("b", np.complex64, pit=dict(tol=tol), start2=a64[0][0])
        print("  stage 1 from the EXACT stage-0 taps (passes %s): %s" % (b2[3][1]["passes"], "; ".join(split(a64[0][1], b2[0][1]))))
        print("      error trace rms dev %s, equaliser output %s" % (["%.2e" % rms(a64[1][1][m] - b2[1][1][m]) for m in range(2)],
                                                                      ["%.2e" % (rms(a64[2][m] - b2[2][m]) / rms(a64[2][m])) for m in range(2)]))
    sys.stdout.flush()
