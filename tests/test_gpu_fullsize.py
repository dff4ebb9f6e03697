"""
Tier b pinned to the REFERENCE at BASELINE.json's full sizes: configs[2] (C3: 64-QAM, 2^22 symbol periods, 41 taps,
cma -> mrde, 64-angle search), the north star's 10^7-symbol variant of it and configs[1] (C2: 16-QAM, 2^20, 21-tap mcma, 32 angles) through ``ResidentReceiver(tier="b")`` against the CPU
oracle (the restatement of the reference's loops that the golden vectors pin, reference-flag build, exact sequential recurrence) on the
same capture, with FLOAT tolerances - not only error counts:

    taps              relative norm per output mode          <= 3 tol
    equaliser output  relative rms per output mode           <= tol
    error traces      rms per stage and mode (signal units)  <= 3 tol
    symbol errors     after the phase search, per mode       within +-3 of the oracle's

tol = 1e-3 (the library default of tier b) AND tol = 1e-4 - SURVEY.md 8c's own bar for complex64 (rtol 1e-4 taps, atol 1e-4 output /
error traces), the tolerance bench.py's headline is held to since round 5.  The two-level statement for the RECOVERED output (after the
arg-min over the test angles) lives in bench.py's certificate; here the recovered signals are compared through their decisions.
"""
import functools
import numpy as np
import pytest

from oracle import oracle
from qampy_amd import synth, _lib
from qampy_amd.core import ber_functions as ber
from qampy_amd.core.equalisation import equalisation as host
from qampy_amd.pipeline import ResidentReceiver

pytestmark = pytest.mark.gpu

TOLS = [1e-3, 1e-4]
EW_RTOL = EW_ATOL = 1e-4          # tests/conftest.py's complex64 bar (np.testing.assert_allclose) - what the EXACT path is held to element by element


def _elementwise(ref, got):
    """assert_allclose's criterion as a measurement: share of elements with |got - ref| <= atol + rtol |ref|, largest deviation."""
    d = np.abs(got - ref)
    ok = d <= EW_ATOL + EW_RTOL * np.abs(ref)
    return float(np.mean(ok)), float(d.max())


def _record(row):
    """Measured element-wise figures, one JSON line per (shape, tolerance, quantity): gpurun_out/elementwise_fullsize.jsonl (copied to profiles/)."""
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "elementwise_fullsize.jsonl"), "a") as f:
        f.write(json.dumps(row) + "\n")
CASES = {
    "c3": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    "c2": dict(M=16, nsym=2 ** 20, ntaps=21, methods=("mcma",), mu=(1e-3,), A=32, snr=25, lw=50e3),
    # the north star's own size: 10^7 symbol periods of the C3 recipe (8-lane chains: 3584 / 4096 segments)
    "ns": dict(M=64, nsym=10 ** 7, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), A=64, snr=30, lw=100.),
    # SURVEY.md 8d's recipe at HALF its step sizes and 1 kHz - the largest steps at which the reference's own recurrence converges at 2^22
    # (profiles/r05_survey_recipe_exact_path.txt) - against the ORACLE, at both tolerances (round-5 verdict: it was held GPU against GPU at 1e-3 only)
    "sr": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(5e-4, 2.5e-4), A=64, snr=30, lw=1e3),
}


def _oracle_chain(c, E, coded):
    """The reference's order of evaluation on the host: every stage, the filter, the phase search, unwrap, de-rotation."""
    nm = E.shape[0]
    w = host._init_taps(c["ntaps"], nm, nm, np.complex64)
    tr = host._cal_training_symbol_len(2, c["ntaps"], E.shape[1])
    errs = []
    for s, m in enumerate(c["methods"]):
        sy = host._reshape_symbols(coded if m in host.DECISION_BASED else None, m, c["M"], np.complex64, nm)
        e, w, _ = oracle.train_equaliser(E, tr, 1, 2, np.float32(c["mu"][s]), w, None, False, sy, m, fast=True)
        errs.append(np.asarray(e))
    eq = oracle.apply_filter_to_signal(E, 2, w, fast=True)
    angles = np.linspace(-np.pi / 4, np.pi / 4, c["A"], endpoint=False, dtype=np.float32).reshape(1, -1)
    ph = np.array([oracle.select_angles(angles, oracle.bps(eq[m], angles, coded, 20, fast=True)) for m in range(nm)])
    ph[:, 20:-20] = np.unwrap(ph[:, 20:-20] * 4) / 4
    return np.asarray(w), errs, np.asarray(eq), (eq * np.exp(1j * ph)).astype(np.complex64)


@functools.lru_cache(maxsize=1)
def _capture_and_oracle(key):
    """The capture (device + host copy) and the oracle's results for it: once per shape, shared by the tolerances (maxsize 1: a 10^7-symbol
    capture with its traces is ~1.3 GB of host memory)."""
    c = CASES[key]
    d = synth.make_capture_dev(c["M"], c["nsym"], nmodes=2, snr_db=c["snr"], theta=np.pi / 5.6, dgd=30e-12, linewidth=c["lw"], seed=1000)
    E = d["E"].to_host()
    return d, E, _oracle_chain(c, E, d["alphabet_host"])


@pytest.mark.parametrize("tol", TOLS, ids=lambda t: "tol%g" % t)
@pytest.mark.parametrize("key", ["c2", "c3", "ns", "sr"])
def test_tier_b_at_full_size_against_the_oracle(key, tol):
    c = CASES[key]
    d, E, (wo, eo, qo, oo) = _capture_and_oracle(key)
    coded = d["alphabet_host"]
    rx = ResidentReceiver(2, E.shape[1], 2, c["M"], c["ntaps"], c["mu"], methods=c["methods"], Niter=(1,) * len(c["methods"]), Mtestangles=c["A"], Nbps=20,
                          alphabet=coded, tier="b", pit=dict(tol=tol))
    rx.E.copy_from(d["E"])
    rx.run()
    res = rx.fetch()
    reps = rx.pit_reports()
    assert all(r["converged"] for r in reps), reps
    assert abs(reps[-1]["tol"] - tol) < 1e-12 and all(abs(r["tol"] - ResidentReceiver.NONFINAL_TOL_FACTOR * tol) < 1e-12 for r in reps[:-1]), reps
    # what the headline is quoted on: the parallel-in-time solver certified every stage itself (no exact-form way out was needed)
    assert not any(r["exact_form"] for r in reps), reps
    for m in range(2):
        g = 1j ** int(np.rint(np.angle(np.vdot(res["wxy"][m].ravel(), wo[m].ravel())) / (np.pi / 2)))       # a common quarter turn is a symmetry
        assert g == 1, "tier b starts from the caller's taps: same quadrant as the sequential recurrence"
        tap = np.linalg.norm(wo[m] - res["wxy"][m]) / np.linalg.norm(wo[m])
        out = np.sqrt(np.mean(np.abs(qo[m] - res["eq"][m]) ** 2) / np.mean(np.abs(qo[m]) ** 2))
        assert tap <= 3 * tol, (key, m, "taps", tap)
        assert out <= tol, (key, m, "equaliser output", out)
        for s in range(len(c["methods"])):
            et = np.sqrt(np.mean(np.abs(eo[s][m] - res["err"][s][m]) ** 2))
            assert et <= 3 * tol, (key, m, "error trace of stage %d" % s, et)
        # ---- element by element, at the bar the exact path is held to (rtol = atol = 1e-4): taps and equaliser output in full at tol = 1e-4;
        # the error traces are measured and reported (an error function turns an output deviation into 1.3 - 1.5 x as much trace deviation)
        sh_t, mx_t = _elementwise(wo[m], res["wxy"][m])
        sh_q, mx_q = _elementwise(qo[m], res["eq"][m])
        tr_rows = [_elementwise(eo[s][m], res["err"][s][m]) for s in range(len(c["methods"]))]
        _record(dict(shape=key, tol=tol, mode=m, taps=dict(share=sh_t, max_abs=mx_t), equaliser_out=dict(share=sh_q, max_abs=mx_q),
                     err_traces=[dict(stage=c["methods"][s], share=a_, max_abs=b_) for s, (a_, b_) in enumerate(tr_rows)]))
        if tol <= 1e-4:
            assert sh_t == 1.0, (key, m, "taps element-wise", sh_t, mx_t)
            assert sh_q == 1.0, (key, m, "equaliser output element-wise", sh_q, mx_q)
    # decisions after carrier recovery: tier b on the device against the oracle's recovered signal through the same harness
    ser_b = ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)
    oo_dev = _lib.DeviceArray.from_host(np.ascontiguousarray(oo))
    ser_o = ber.cal_ser_dev(oo_dev, d["idx_tx"], rx.alphabet, 256, 8192, 2000)
    for m in range(2):
        assert abs(ser_b[m]["errors"] - ser_o[m]["errors"]) <= 3, (key, m, ser_b[m], ser_o[m])


def test_config1_exact_shape_against_the_oracle():
    """BASELINE.json configs[0] at its exact shape (QPSK, 1 polarisation, 2 samples per symbol, 2^16 symbols, 11-tap CMA, mu = 1e-3; the plumbing
    of Scripts/cma_equaliser.py; CPU-only by definition in BASELINE, run here through both tiers): taps, error trace and filter output of the exact path
    and of tier b against the oracle at the complex64 parity tolerance (tests/conftest.py), symbol errors identical."""
    nsym, nt = 2 ** 16, 11
    d = synth.make_capture_dev(4, nsym, nmodes=1, snr_db=14, theta=None, dgd=30e-12, linewidth=0., seed=1000)
    E = d["E"].to_host()
    w0 = host._init_taps(nt, 1, 1, np.complex64)
    tr = host._cal_training_symbol_len(2, nt, E.shape[1])
    sy = host._reshape_symbols(None, "cma", 4, np.complex64, 1)
    eo, wo, _ = oracle.train_equaliser(E, tr, 1, 2, np.float32(1e-3), w0.copy(), None, False, sy, "cma", fast=True)
    qo = np.asarray(oracle.apply_filter_to_signal(E, 2, wo, fast=True))
    for tier, tol in (("a", 1e-4), ("b", 1e-4)):
        rx = ResidentReceiver(1, E.shape[1], 2, 4, nt, (1e-3,), methods=("cma",), Niter=(1,), Mtestangles=None, alphabet=d["alphabet_host"], tier=tier,
                              pit=dict(tol=1e-4) if tier == "b" else None)
        rx.E.copy_from(d["E"])
        rx.run()
        res = rx.fetch()
        # cma is phase blind: tier b's result is the sequential recurrence's up to a common phase of the output mode (DESIGN.md 3.2.1 gauge)
        g = np.exp(1j * np.angle(np.vdot(res["wxy"][0].ravel(), np.asarray(wo)[0].ravel()))) if tier == "b" else 1.0
        if tier == "b":
            assert all(r["converged"] for r in rx.pit_reports())
            assert abs(np.angle(g)) < 5e-3
        tap = np.linalg.norm(np.asarray(wo)[0] - g * res["wxy"][0]) / np.linalg.norm(np.asarray(wo)[0])
        out = np.sqrt(np.mean(np.abs(qo[0] - g * res["eq"][0]) ** 2) / np.mean(np.abs(qo[0]) ** 2))
        et = np.sqrt(np.mean(np.abs(np.asarray(eo)[0] - g * res["err"][0][0]) ** 2))
        assert tap <= 3 * tol and out <= tol and et <= 3 * tol, (tier, tap, out, et)
        # decisions (QPSK without carrier recovery: modulo the quarter-turn ambiguity the harness resolves)
        ser = ber.cal_ser_dev(rx.eq, d["idx_tx"], _lib.DeviceArray.from_host(np.ascontiguousarray(d["alphabet_host"], dtype=np.complex64)), 256, 8192, 2000)
        ser_o = ber.cal_ser_dev(_lib.DeviceArray.from_host(np.ascontiguousarray(qo.astype(np.complex64))), d["idx_tx"],
                                _lib.DeviceArray.from_host(np.ascontiguousarray(d["alphabet_host"], dtype=np.complex64)), 256, 8192, 2000)
        assert abs(ser[0]["errors"] - ser_o[0]["errors"]) <= 3, (tier, ser, ser_o)


def _gpu_pair(M, nsym, mu, lw, tol, seed=1000):
    """tier a and tier b (ResidentReceiver) on one device-synthesised C3-shaped capture: results + symbol errors + tier b's reports."""
    d = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=lw, seed=seed)
    kw = dict(methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"])
    res = {}
    for tier in ("a", "b"):
        rx = ResidentReceiver(2, 2 * nsym, 2, M, 41, mu, tier=tier, pit=dict(tol=tol) if tier == "b" else None, **kw)
        rx.E.copy_from(d["E"])
        rx.run()
        r = rx.fetch()
        r["errors"] = [s["errors"] for s in ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, min(8192, nsym // 4), 2000)]
        r["rep"] = rx.pit_reports()
        res[tier] = r
        del rx
    return res["a"], res["b"]


def test_survey_recipe_step_sizes_at_full_size():
    """SURVEY.md 8d's LITERAL C3 recipe, mu = (1e-3, 5e-4), 5 kHz (round-4 verdict item 2).  What the exact path itself does with it is part of the test:
    it converges at 2^18 symbols and does NOT at 2^22 (at least one mode ends with more than 10 % symbol errors - profiles/r05_survey_recipe_exact_path.txt:
    every seed, every linewidth) - there tier b certifies the constant-modulus stage and hands the stage whose trajectory is not a contraction to the
    exact form, and the mode that did converge decodes identically.  At half the step sizes and 1 kHz the reference's recurrence converges at 2^22 and tier b
    certifies both stages within the float tolerances of the top of this file."""
    tol = 1e-3
    # (i) literal recipe where the reference converges
    a, b = _gpu_pair(64, 2 ** 18, (1e-3, 5e-4), 5e3, tol)
    assert a["errors"] == [0, 0] and b["errors"] == [0, 0] and all(r["converged"] and not r["exact_form"] for r in b["rep"]), (a["errors"], b["errors"], b["rep"])
    for m in range(2):
        assert np.linalg.norm(a["wxy"][m] - b["wxy"][m]) / np.linalg.norm(a["wxy"][m]) <= 3 * tol
        assert np.sqrt(np.mean(np.abs(a["eq"][m] - b["eq"][m]) ** 2) / np.mean(np.abs(a["eq"][m]) ** 2)) <= tol
    # (ii) literal recipe at full size: the reference's recurrence fails on a mode; tier b stays total
    a, b = _gpu_pair(64, 2 ** 22, (1e-3, 5e-4), 5e3, tol)
    bad = [e > 0.1 * 2 ** 22 for e in a["errors"]]
    assert any(bad), ("the exact path converged on the literal recipe at 2^22: update DESIGN.md 6 / bench.survey_recipe_block", a["errors"])
    assert b["rep"][0]["converged"] and not b["rep"][0]["exact_form"], b["rep"][0]
    assert b["rep"][1]["converged"] and b["rep"][1]["exact_form"], b["rep"][1]
    for m in range(2):
        if not bad[m]:
            assert abs(a["errors"][m] - b["errors"][m]) <= 3, (m, a["errors"], b["errors"])
    # (iii) half the step sizes, 1 kHz: converges, certified
    a, b = _gpu_pair(64, 2 ** 22, (5e-4, 2.5e-4), 1e3, tol)
    assert a["errors"] == [0, 0] and b["errors"] == [0, 0] and all(r["converged"] and not r["exact_form"] for r in b["rep"]), (a["errors"], b["errors"], b["rep"])
    for m in range(2):
        assert np.linalg.norm(a["wxy"][m] - b["wxy"][m]) / np.linalg.norm(a["wxy"][m]) <= 3 * tol
        assert np.sqrt(np.mean(np.abs(a["eq"][m] - b["eq"][m]) ** 2) / np.mean(np.abs(a["eq"][m]) ** 2)) <= tol
