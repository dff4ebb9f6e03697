"""Soak of the per-thread library state: (1) the adaptive-step recipe (mcma -> mddma, shared step) on a ReceiverGroup of 2 against one receiver,
(2) the mirrored host API (numpy in / out, tiers a and b) called from three host threads at once; every result bit for bit the single-threaded one."""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from qampy_amd import synth, _lib, equalisation
from qampy_amd.pipeline import ResidentReceiver, ReceiverGroup
bad = 0
# 1. adaptive recipe (mcma -> mddma, shared adaptive step) on a group of 2 against one receiver
nsym, M, ntaps, mu = 2 ** 17, 64, 13, (1.9e-3, 1.9e-3)
caps = [synth.make_capture_dev(M, nsym, nmodes=2, snr_db=25, theta=np.pi / 3, dgd=30e-12, linewidth=0., seed=sd) for sd in (1000, 1001)]
kw = dict(methods=("mcma", "mddma"), Niter=(1, 1), adaptive_stepsize=(True, True), Mtestangles=None, alphabet=caps[0]["alphabet_host"], tier="b")
rx = ResidentReceiver(2, 2 * nsym, 2, M, ntaps, mu, **kw)
ref = []
for c in caps:
    rx.E.copy_from(c["E"]); rx.run(); ref.append(rx.fetch())
g = ReceiverGroup(2, 2, 2 * nsym, 2, M, ntaps, mu, **kw)
for rnd in range(4):
    for j, r in enumerate(g.rx):
        r.E.copy_from(caps[(j + rnd) % 2]["E"])
    _lib.sync(); g.run(6)
    for j, r in enumerate(g.rx):
        res = r.fetch()
        ok = all(np.array_equal(res[k], ref[(j + rnd) % 2][k]) for k in ("wxy", "eq")) and all(np.array_equal(a, b) for a, b in zip(res["err"], ref[(j + rnd) % 2]["err"])) and res["mu"] == ref[(j + rnd) % 2]["mu"]
        if not ok: bad += 1; print("MISMATCH adaptive group", rnd, j)
g.close()
print("adaptive group done, bad", bad, [r["passes"] for r in rx.pit_reports()])
# 2. the mirrored host API (numpy in / out) from two threads at once
sigs = [synth.make_capture(64, 2 ** 16, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=sd, dtype=np.complex64) for sd in (1000, 1001, 1002, 1003)]
def one(sig, tier):
    out, w, e = equalisation.dual_mode_equalisation(sig, (1e-3, 5e-4), 41, methods=("cma", "mrde"), tier=tier)
    return np.asarray(out), w
want = {(i, t): one(s, t) for i, s in enumerate(sigs) for t in ("a", "b")}
res = {}
def work(ids, t):
    for rep in range(6):
        for i in ids:
            res[(i, t, rep)] = one(sigs[i], t)
for t in ("a", "b"):
    th = [threading.Thread(target=work, args=(ids, t)) for ids in ((0, 1), (2, 3), (1, 3))]
    for x in th: x.start()
    for x in th: x.join()
for (i, t, rep), (o, w) in res.items():
    if not (np.array_equal(o, want[(i, t)][0]) and np.array_equal(w, want[(i, t)][1])):
        bad += 1; print("MISMATCH host api", i, t, rep)
print("host API from 3 threads done,", len(res), "calls, bad", bad)
print("SOAK2", "FAILED" if bad else "OK")
