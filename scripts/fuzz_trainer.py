#!/usr/bin/env python3
"""Randomised differential test of the trainers: random shapes, methods, step-size modes, sweeps and mode subsets, default kernel
choice (and a randomly forced form) against the oracle in complex128 (tight tolerance) - prints failures and a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import _lib as _qlib
from qampy_amd import _lib
from qampy_amd.core.equalisation import equalisation as eq, hip_equalisation as hk
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
_lib.init(0)
fails, t0 = [], time.time()
for case in range(n_cases):
    nmodes = int(rng.choice([1, 2, 2, 2, 3, 4]))
    os_ = int(rng.choice([1, 2, 2, 2, 3]))
    ntaps = int(rng.integers(1, 50))
    M = int(rng.choice([4, 16, 64]))
    method = str(rng.choice(["cma", "mcma", "cma2", "rde", "mrde", "sbd", "mddma", "dd", "sbd_data"]))
    if method in ("rde",) and M == 4:
        M = 16
    tr = int(rng.integers(1, int(os.environ.get("FUZZ_TR", 2500))))
    niter = int(rng.choice([1, 1, 2, 3]))
    adaptive = rng.choice([0, 0, 1, 2])
    form = str(rng.choice(["", "", "direct", "lookahead", "iterative"]))
    L = (tr - 1) * os_ + ntaps + int(rng.integers(0, 7))
    alphabet = eq._reshape_symbols(None, "sbd", M, np.complex128, 1)[0]
    tx = alphabet[rng.integers(0, M, (nmodes, L))]
    E = np.ascontiguousarray(tx + 0.05 * (rng.normal(size=tx.shape) + 1j * rng.normal(size=tx.shape)))
    w0 = eq._init_taps(ntaps, nmodes, nmodes, np.complex128) + 0.02 * (rng.normal(size=(nmodes, nmodes, ntaps)) + 1j * rng.normal(size=(nmodes, nmodes, ntaps)))
    if method == "sbd_data":
        sy = np.ascontiguousarray(tx[:, ::os_][:, :max(tr, 1)] if tx[:, ::os_].shape[1] >= tr else np.tile(tx[:, ::os_], (1, tr))[:, :tr])
    else:
        sy = eq._reshape_symbols(None, method, M, np.complex128, nmodes)
    k = int(rng.integers(1, nmodes + 1))
    modes = rng.permutation(nmodes)[:k].astype(np.int64) if rng.random() < 0.5 else None
    # step sizes well inside the stable region: close to the stability limit the recurrence amplifies rounding differences
    # between ANY two orders of summation (all forms, the oracle) exponentially and a tight comparison is meaningless
    mu = np.float64(rng.choice([1e-4, 5e-4, 2e-3]) / max(1., nmodes * ntaps / 10.) if method != "cma2" else 1e-5)
    sel = np.arange(nmodes) if modes is None else modes
    if adaptive == 2:
        wo, eo, muo = w0.copy(), np.zeros((nmodes, tr * niter), np.complex128), mu
        for m in sel:
            e1, wo, muo = oracle.train_equaliser(E, tr, niter, os_, mu, wo, np.array([m]), True, sy, method)
            eo[m] = e1[m]
    else:
        eo, wo, muo = oracle.train_equaliser(E, tr, niter, os_, mu, w0.copy(), modes, bool(adaptive), sy, method)
    _qlib.set_form("trainer", None)
    if form:
        _qlib.set_form("trainer", form)
    try:
        with np.errstate(all="ignore"):
            e, w, mu2 = hk.train_equaliser(E, tr, niter, os_, mu, w0.copy(), modes, [False, True, "per-mode"][adaptive], sy, method)
        fin = np.all(np.isfinite(wo))
        ok = (not fin and not np.all(np.isfinite(w))) or (np.allclose(w, wo, rtol=1e-8, atol=1e-10) and np.allclose(e, eo, rtol=1e-8, atol=1e-9)
                                                          and np.isclose(mu2, muo, rtol=1e-8))
    except Exception as ex:
        ok, w = False, ex
    if not ok:
        desc = dict(case=case, nmodes=nmodes, os=os_, ntaps=ntaps, M=M, method=method, tr=tr, niter=niter, adaptive=int(adaptive), form=form,
                    modes=None if modes is None else modes.tolist(), mu=float(mu))
        if not isinstance(w, Exception):
            desc["max_tap_diff"] = float(np.nanmax(np.abs(w - wo)))
            desc["max_err_diff"] = float(np.nanmax(np.abs(e - eo)))
            desc["mu"] = [float(mu2), float(muo)]
        else:
            desc["exception"] = repr(w)
        fails.append(desc)
        print("FAIL", desc, flush=True)
_qlib.set_form("trainer", None)
print("fuzz: %d cases, %d failures, %.1f s" % (n_cases, len(fails), time.time() - t0))
