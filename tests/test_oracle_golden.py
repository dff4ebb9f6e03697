"""
Pins the oracle (oracle/qampy_oracle.c) against the vectors captured from the imported reference.
CPU only.  The strict oracle build evaluates the same scalar expressions in the same order as the pure-Python run of
the reference, so agreement is far tighter than the tolerances used for the HIP path.
"""
import numpy as np
import pytest

from conftest import CT, RT, golden_cases
from oracle import oracle


def _close(a, b, dn, scale=1.0):
    if dn == "c128":
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12 * scale)
    else:
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("case", [c for c in golden_cases("train") if not c.get("real")], ids=lambda c: c["name"])
def test_train_equaliser(golden, case):
    g = golden["train"]
    n, dn = case["name"], case["dtype"]
    E = golden.input(case["input"], CT[dn])
    wx = g[n + "__wx0"].copy()
    err, wx, mu = oracle.train_equaliser(E, case["TrSyms"], case["Niter"], case["os"], RT[dn](case["mu"]), wx,
                                         np.array(case["modes"]), case["adaptive"], g[n + "__symbols"], case["method"])
    assert err.dtype == CT[dn] and err.shape == g[n + "__err"].shape
    _close(wx, g[n + "__wx"], dn)
    _close(err, g[n + "__err"], dn, scale=10)
    _close(mu, g[n + "__mu"], dn)


@pytest.mark.parametrize("case", golden_cases("train_cross"), ids=lambda c: c["name"])
def test_train_equaliser_cross_qam(golden, case):
    """32- / 128-QAM (cross constellations): decision-directed and partition-based error functions where no per-axis slicer applies
    (tests/golden/gen_golden_cross.py; pythran_equalisation.py:240-265 det_symbol, :4-9 partition_value)."""
    g = golden["train_cross"]
    n, dn = case["name"], case["dtype"]
    E = np.ascontiguousarray(g[case["input"] + "_E"].astype(CT[dn]))
    wx = g[n + "__wx0"].copy()
    err, wx, mu = oracle.train_equaliser(E, case["TrSyms"], case["Niter"], case["os"], RT[dn](case["mu"]), wx,
                                         np.array(case["modes"]), case["adaptive"], g[n + "__symbols"], case["method"])
    assert err.dtype == CT[dn] and err.shape == g[n + "__err"].shape
    _close(wx, g[n + "__wx"], dn)
    _close(err, g[n + "__err"], dn, scale=10)
    _close(mu, g[n + "__mu"], dn)


@pytest.mark.parametrize("case", [c for c in golden_cases("train") if c.get("real")], ids=lambda c: c["name"])
def test_train_equaliser_realvalued(golden, case):
    g = golden["train"]
    n, dn = case["name"], case["dtype"]
    Ec = golden.input(case["input"], CT[dn])
    E = np.ascontiguousarray(np.vstack([Ec.real, Ec.imag]))
    wx = g[n + "__wx0"].copy()
    err, wx, mu = oracle.train_equaliser_realvalued(E, case["TrSyms"], case["Niter"], 2, RT[dn](case["mu"]), wx,
                                                    np.array(case["modes"]), case["adaptive"], g[n + "__symbols"],
                                                    case["method"][:-5])
    _close(wx, g[n + "__wx"], dn)
    _close(err, g[n + "__err"], dn, scale=10)
    _close(mu, g[n + "__mu"], dn)


@pytest.mark.parametrize("case", [c for c in golden_cases("apply") if not c.get("realtaps")], ids=lambda c: c["name"])
def test_apply_filter(golden, case):
    g = golden["apply"]
    n, dn = case["name"], case["dtype"]
    E = golden.input(case["input"], CT[dn])
    out = oracle.apply_filter_to_signal(E, case["os"], g[n + "__wx"], case["modes"])
    assert out.shape == g[n + "__out"].shape
    _close(out, g[n + "__out"], dn)


@pytest.mark.parametrize("case", [c for c in golden_cases("bps") if "base" in c], ids=lambda c: c["name"])
def test_bps_index(golden, case):
    g = golden["bps"]
    dn = case["dtype"]
    E = g[case["base"] + "__E"].astype(CT[dn])
    angles = np.linspace(-np.pi / 4, np.pi / 4, case["A"], endpoint=False, dtype=RT[dn]).reshape(1, -1)
    alphabet = g[case["base"] + "__alphabet"].astype(CT[dn])
    for m in range(E.shape[0]):
        idx = oracle.bps(E[m], angles, alphabet, case["N"])
        ref = g[case["name"] + "__idx"][m]
        assert idx.dtype == np.int32
        # identical running float sums -> identical indices
        assert np.array_equal(idx, ref), "mode %d: %d mismatches" % (m, np.count_nonzero(idx != ref))
        ph = oracle.select_angles(angles, idx)
        assert np.array_equal(ph[:case["N"]], np.full(case["N"], angles[0, 0]))


def test_bps_per_symbol_grid(golden):
    g = golden["bps"]
    idx = oracle.bps(g["bps_grid__E"], g["bps_grid__angles"], g["bps_grid__alphabet"], 10)
    assert np.array_equal(idx, g["bps_grid__idx"])
    assert np.array_equal(oracle.select_angles(g["bps_grid__angles"], idx), g["bps_grid__sel"])
    assert np.array_equal(oracle.select_angles(g["bps_grid__angles"][:1].copy(), idx), g["bps_grid__sel1"])


@pytest.mark.parametrize("M", [4, 16, 32, 64, 128])
@pytest.mark.parametrize("dn", ["c128", "c64"])
def test_make_decision(golden, M, dn):
    g = golden["decision"]
    det, dist, idx = oracle.make_decision(g["md_M%d__E" % M].astype(CT[dn]), g["md_M%d__alphabet" % M].astype(CT[dn]))
    assert np.array_equal(idx, g["md_M%d_%s__idx" % (M, dn)])
    assert np.array_equal(det, g["md_M%d_%s__det" % (M, dn)])
    np.testing.assert_allclose(dist, g["md_M%d_%s__dist" % (M, dn)], rtol=1e-6 if dn == "c64" else 1e-14)


def test_unknown_method_raises():
    E = np.zeros((1, 64), np.complex64)
    with pytest.raises(ValueError):
        oracle.train_equaliser(E, 4, 1, 2, np.float32(1e-3), np.zeros((1, 1, 5), np.complex64), None, False,
                               np.ones((1, 1), np.complex64), "nonsense")
