#!/usr/bin/env python3
"""Randomised differential test of filter application, blind phase search, angle selection and decisions against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import _lib, theory
from qampy_amd.core import hip_dsp
from qampy_amd.core.equalisation import hip_equalisation as hk
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
_lib.init(0)
fails, t0, worst_bps = [], time.time(), 0.
for case in range(n_cases):
    ct = [np.complex64, np.complex128][int(rng.integers(0, 2))]
    rt = np.float32 if ct is np.complex64 else np.float64
    # ---- apply (complex and real taps)
    nmodes, os_, ntaps = int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 70))
    L = ntaps - 1 + int(rng.integers(0, 6000))          # shorter fields: (nsel, 0) here, a numpy error in the reference
    real = rng.random() < 0.25
    dt = rt if real else ct
    E = rng.normal(size=(nmodes, L)).astype(dt) if real else (rng.normal(size=(nmodes, L)) + 1j * rng.normal(size=(nmodes, L))).astype(ct)
    w = rng.normal(size=(nmodes, nmodes, ntaps)).astype(dt) if real else (rng.normal(size=(nmodes, nmodes, ntaps)) + 1j * rng.normal(size=(nmodes, nmodes, ntaps))).astype(ct)
    modes = rng.permutation(nmodes)[:int(rng.integers(1, nmodes + 1))].astype(np.int64) if rng.random() < 0.5 else None
    a, b = hk.apply_filter_to_signal(E, os_, w, modes), oracle.apply_filter_to_signal(E, os_, w, modes)
    tol = 2e-4 if ct is np.complex64 else 1e-11
    if a.shape != b.shape or not np.allclose(a, b, rtol=tol, atol=tol * ntaps * nmodes):
        fails.append(("apply", case, nmodes, os_, ntaps, L, real, str(dt)))
    # ---- bps + select_angles + make_decision
    M = int(rng.choice([4, 16, 32, 64, 128, 256]))
    alphabet = theory.coded_symbols_qam(M, dtype=ct)
    Lb, A, N = int(rng.integers(1, 5000)), int(rng.choice([4, 16, 20, 32, 64, 100])), int(rng.integers(1, 40))
    sig = (alphabet[rng.integers(0, M, Lb)] * np.exp(1j * 0.3) + 0.03 * (rng.normal(size=Lb) + 1j * rng.normal(size=Lb))).astype(ct)
    per_symbol = rng.random() < 0.3
    ang = np.ascontiguousarray(np.linspace(-np.pi / 4, np.pi / 4, A, endpoint=False).reshape(1, A) + (rng.normal(size=(Lb, 1)) * 0.01 if per_symbol else 0), dtype=rt)
    gi, oi = hip_dsp.bps(sig, ang, alphabet, N), oracle.bps(sig, ang, alphabet, N)
    mism = np.nonzero(gi != oi)[0]
    frac = mism.size / max(Lb, 1)
    near = (not mism.size) or np.max(np.minimum((gi[mism] - oi[mism]) % A, (oi[mism] - gi[mism]) % A)) <= max(2, A // 8)
    # near-ties: direct window sums here vs the reference's differences of running sums
    bad = mism.size > 2 and (frac > 2e-3 if ct is np.complex128 else (frac > 5e-2 or (frac > 1e-2 and not near and N >= 3)))
    # complex64: the oracle (like the reference) differences a float32 RUNNING sum over the whole capture, whose rounding error
    # exceeds the gaps between competing angles for short windows - any angle can win a near-tie there; complex128 mismatches
    # must be neighbouring angles
    far = ct is np.complex128 and mism.size and np.max(np.minimum((gi[mism] - oi[mism]) % A, (oi[mism] - gi[mism]) % A)) > max(2, A // 8)
    if gi.shape != oi.shape or bad or far:
        fails.append(("bps", case, M, Lb, A, N, per_symbol, str(ct), float(frac)))
    worst_bps = max(worst_bps, frac)
    sa, so = hip_dsp.select_angles(ang, gi), oracle.select_angles(ang, np.asarray(gi, dtype=np.int64))
    if not np.array_equal(sa, so):
        fails.append(("select_angles", case))
    d1, d2 = hk.make_decision(sig, alphabet), oracle.make_decision(sig, alphabet)
    dm = np.nonzero(d1[2] != d2[2])[0]
    # a different index is only acceptable for a tie within the rounding of the distance (hypot on the device vs libm)
    tie = all(abs(abs(sig[i] - alphabet[d1[2][i]]) - abs(sig[i] - alphabet[d2[2][i]])) <= 4 * np.finfo(rt).eps * abs(sig[i] - alphabet[d2[2][i]]) for i in dm)
    same = np.ones(Lb, bool); same[dm] = False
    if not (tie and dm.size <= 1 and np.array_equal(d1[0][same], d2[0][same]) and np.allclose(d1[1], d2[1], rtol=1e-5 if ct is np.complex64 else 1e-12)):
        fails.append(("make_decision", case, M, Lb, str(ct), int(dm.size)))
for f in fails[:20]:
    print("FAIL", f)
print("fuzz-dsp: %d cases, %d failures, worst bps mismatch fraction %.2e, %.1f s" % (n_cases, len(fails), worst_bps, time.time() - t0))
