"""Soak of the overlapped / multi-receiver schedules: many consecutive passes over DIFFERENT captures, every result compared bit for bit with the
same receiver run one capture at a time.  Usage: python scripts/soak_overlap.py [passes] [log2 symbols] [c3 | c2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver, ReceiverGroup

npass = int(sys.argv[1]) if len(sys.argv) > 1 else 300
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
nsym, M, ntaps, mu = 2 ** log2n, 64, 41, (1e-3, 5e-4) if log2n < 20 else (2e-4, 2e-4)
seeds = (1000, 1003, 1007, 1011)
caps = [synth.make_capture_dev(M, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=sd) for sd in seeds]
kw = dict(methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=caps[0]["alphabet_host"])
if len(sys.argv) > 3 and sys.argv[3] == "c2":           # configs[1]: 16-QAM, 21-tap mcma, 32 test angles, 50 kHz linewidth
    M, ntaps, mu = 16, 21, (1e-3,)
    caps = [synth.make_capture_dev(M, nsym, nmodes=2, snr_db=25, theta=np.pi / 5.6, dgd=30e-12, linewidth=50e3, seed=sd) for sd in seeds]
    kw = dict(methods=("mcma",), Niter=(1,), Mtestangles=32, Nbps=20, alphabet=caps[0]["alphabet_host"])
KEYS = ("wxy", "eq", "out", "ph", "idx")
bad = 0
for tier in ("b", "a"):
    rx = ResidentReceiver(2, 2 * nsym, 2, M, ntaps, mu, tier=tier, **kw)
    ref = []
    for c in caps:
        rx.E.copy_from(c["E"]); rx.run(); ref.append(rx.fetch())
    t0 = time.time()
    n = npass if tier == "b" else max(8, npass // 10)
    prev = None
    for k in range(n):
        i = (k * 7 + k // 5) % len(caps)
        rx.E.copy_from(caps[i]["E"])
        rx.run(overlap=True)
        if prev is not None:                  # the search of the previous capture ran beside this training
            _lib.sync()
            got = {q: getattr(rx, q).to_host() for q in ("out", "ph", "idx")}
            if not all(np.array_equal(got[q], ref[prev][q]) for q in got):
                bad += 1; print("MISMATCH tier", tier, "pass", k - 1, "capture", prev, flush=True)
        prev = i
    res = rx.fetch()
    if not all(np.array_equal(res[q], ref[prev][q]) for q in KEYS):
        bad += 1; print("MISMATCH tier", tier, "last pass", flush=True)
    print("tier %s: %d overlapped passes over %d captures of 2^%d symbols, %.1f s, mismatches so far %d" % (tier, n, len(caps), log2n, time.time() - t0, bad), flush=True)
    if tier == "b":
        g = ReceiverGroup(3, 2, 2 * nsym, 2, M, ntaps, mu, tier="b", **kw)
        try:
            for rnd in range(max(2, npass // 30)):
                for j, r in enumerate(g.rx):
                    r.E.copy_from(caps[(j + rnd) % len(caps)]["E"])
                _lib.sync()
                g.run(3 * 5)
                for j, r in enumerate(g.rx):
                    res = r.fetch()
                    if not all(np.array_equal(res[q], ref[(j + rnd) % len(caps)][q]) for q in KEYS):
                        bad += 1; print("MISMATCH group round", rnd, "receiver", j, flush=True)
            print("group of 3: %d rounds of 15 passes, mismatches so far %d" % (max(2, npass // 30), bad), flush=True)
        finally:
            g.close()
    del rx
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
