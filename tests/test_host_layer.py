"""
Host logic of qampy_amd (constants, symbol reshaping, tap init, defaults, real<->complex packing, wrappers) against the
vectors captured from the reference.  CPU only: the three kernel-level modules are monkeypatched with the ORACLE so that
the Python host layer can be exercised without a GPU (the product itself has no such fallback).
"""
import numpy as np
import pytest

from conftest import CT, RT
from oracle import oracle
from oracle_kernels import OracleJobs
import qampy_amd
from qampy_amd import theory, synth
from qampy_amd.signals import SignalQAM
from qampy_amd.core.equalisation import equalisation as core_eq
from qampy_amd.core import phaserecovery as core_ph


@pytest.fixture
def oracle_kernels(monkeypatch):
    k = core_eq._kernels

    class OracleField:                      # stands in for hip_equalisation.ResidentField (the capture resident in HBM)
        def __init__(self, E, defer=False):
            self.E = E

        def finish(self):
            pass

        def train(self, *a):
            return oracle.train_equaliser(self.E, *a)

        def apply(self, os, wx, modes=None):
            return oracle.apply_filter_to_signal(self.E, os, np.ascontiguousarray(wx), modes)

    monkeypatch.setattr(k, "ResidentField", OracleField)
    monkeypatch.setattr(k, "ResidentJobs", OracleJobs)
    monkeypatch.setattr(k, "train_equaliser", oracle.train_equaliser)
    monkeypatch.setattr(k, "train_equaliser_realvalued", oracle.train_equaliser_realvalued)
    monkeypatch.setattr(k, "apply_filter_to_signal", oracle.apply_filter_to_signal)
    def oracle_recover(E, Mtestangles, symbols, N):          # what hip_dsp.bps_recover fuses on the device, from the oracle's parts
        rt = E.real.dtype.type
        angles = np.linspace(-np.pi / 4, np.pi / 4, Mtestangles, endpoint=False, dtype=rt).reshape(1, -1)
        ph = np.array([oracle.select_angles(angles, oracle.bps(np.ascontiguousarray(r), angles, symbols, N)) for r in E], dtype=rt)
        ph[:, N:-N] = np.unwrap(ph[:, N:-N] * 4) / 4
        return E * np.exp(1j * ph), ph

    monkeypatch.setattr(core_ph._dsp, "bps_recover", oracle_recover)
    monkeypatch.setattr(core_ph, "_bps_idx_hip", oracle.bps)
    monkeypatch.setattr(core_ph, "select_angles", oracle.select_angles)


# ------------------------------------------------------------------------------------------------ constants (row H)
@pytest.mark.parametrize("M", [4, 16, 32, 64, 128, 256])
def test_constants_bit_exact(golden, M):
    g = golden["constants"]
    assert np.array_equal(theory.cal_symbols_qam(M), g["M%d_symbols" % M])
    assert theory.cal_scaling_factor_qam(M) == g["M%d_scale" % M]
    assert np.array_equal(theory.gray_code_qam(M), g["M%d_graycode" % M])
    assert theory.cal_Rconstant(M) == g["M%d_R" % M]
    assert theory.cal_Rconstant_complex(M) == g["M%d_Rc" % M]
    assert np.array_equal(theory.generate_partition_codes_radius(M), g["M%d_rde" % M])
    assert np.array_equal(theory.generate_partition_codes_complex(M), g["M%d_mrde" % M])
    for dn, dt in CT.items():
        cs = theory.coded_symbols_qam(M, dt)
        assert cs.dtype == dt and np.array_equal(cs, g["M%d_coded_%s" % (M, dn)])
        for method in ("cma", "cma2", "sgncma", "mcma", "rde", "mrde", "sbd", "mddma", "dd", "sgncma_real", "cma_real",
                       "dd_real"):
            ref = g["M%d_eqsyms_%s_%s" % (M, method, dn)]
            got = core_eq.generate_symbols_for_eq(method, M, dt)
            assert got.dtype == ref.dtype and got.shape == ref.shape and np.array_equal(got, ref), (method, dn)


def test_known_answer_constants():
    # SURVEY.md §8(c.1)
    assert theory.cal_scaling_factor_qam(16) == 10 and theory.cal_scaling_factor_qam(64) == 42
    assert abs(theory.cal_Rconstant(16) - 1.32) < 1e-12
    assert abs(theory.cal_Rconstant_complex(64) - (0.880952380952381 + 0.880952380952381j)) < 1e-12
    assert theory.generate_partition_codes_radius(128).size == 2 * 17 - 1      # float-duplicate kept, like the reference
    assert theory.generate_partition_codes_radius(256).size == 2 * 34 - 1


def test_reshape_symbols(golden):
    g = golden["constants"]
    s16 = g["coded16_input"]
    for key in [k for k in g.files if k.startswith("reshape_")]:
        _, rest = key.split("_", 1)
        parts = rest.rsplit("_", 2)
        method, nm, src = parts[0], int(parts[1][1:]), parts[2]
        real = method.endswith("_real")
        if src == "none":
            dt = np.float64 if real else np.complex128
            got = core_eq._reshape_symbols(None, method, 16, dt, nm)
        else:
            dt = np.float32 if real else np.complex64
            got = core_eq._reshape_symbols(s16, method, 16, dt, nm)
        assert got.dtype == g[key].dtype and got.shape == g[key].shape and np.array_equal(got, g[key]), key


def test_reshape_symbols_errors():
    with pytest.raises(ValueError):
        core_eq._reshape_symbols(np.ones((3, 4), complex), "sbd", 16, np.complex128, 2)
    with pytest.raises(ValueError):
        core_eq.generate_symbols_for_eq("sbd_data", 16, np.complex128)
    with pytest.raises(ValueError):
        core_eq.generate_symbols_for_eq("bogus", 16, np.complex128)


def test_defaults():
    assert core_eq._cal_training_symbol_len(2, 41, 2 * 2 ** 22) == 4194259
    assert core_eq._cal_training_symbol_len(2, 21, 2 * 2 ** 20) == 1048551        # C2
    w = core_eq._init_taps(11, 2, 2, np.complex64)
    assert w.shape == (2, 2, 11) and w.dtype == np.complex64 and w[0, 0, 5] == 1 and w[1, 1, 5] == 1 and w.sum() == 2
    assert set(core_eq.TRAINING_FCTS) == set(core_eq.DECISION_BASED) | set(core_eq.NONDECISION_BASED)


# ------------------------------------------------------------------------------------------------ end to end (rows G, J)
def _sig(golden, name, dn):
    m = golden.cases["inputs"][name]
    return SignalQAM(golden.input(name, CT[dn]), m["M"], fb=m["fb"], fs=2 * m["fb"],
                     symbols=golden["inputs"][name + "_tx"].astype(CT[dn]))


def _close(a, b, dn):
    if dn == "c128":
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
    else:
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dn", ["c128", "c64"])
def test_e2e_wrappers_match_reference(golden, oracle_kernels, dn):
    g = golden["e2e"]
    s = _sig(golden, "q16_2m", dn)
    out, wxy, err = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=11, method="mcma", adaptive_stepsize=True, apply=True)
    assert type(out) is SignalQAM and out.fs == s.fb and out.dtype == CT[dn] and wxy.dtype == CT[dn]
    _close(np.asarray(out), g["e2e_eq_mcma_%s__out" % dn], dn)
    _close(wxy, g["e2e_eq_mcma_%s__wxy" % dn], dn)
    _close(err, g["e2e_eq_mcma_%s__err" % dn], dn)

    wxy2, err2 = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=9, Niter=2, method="cma", modes=[1])
    out2 = qampy_amd.equalisation.apply_filter(s, wxy2)
    _close(wxy2, g["e2e_eq_cma_m1_%s__wxy" % dn], dn)
    _close(err2, g["e2e_eq_cma_m1_%s__err" % dn], dn)
    assert np.all(err2[0] == 0)                       # rows of unselected modes stay zero
    _close(np.asarray(out2), g["e2e_eq_cma_m1_%s__out" % dn], dn)

    out3, wxy3, (e31, e32) = qampy_amd.equalisation.dual_mode_equalisation(
        s, (2e-3, 5e-4), 11, Niter=(3, 1), methods=("mcma", "sbd"), adaptive_stepsize=(True, True))
    assert type(out3) is SignalQAM
    _close(np.asarray(out3), g["e2e_dual_mcma_sbd_%s__out" % dn], dn)
    _close(wxy3, g["e2e_dual_mcma_sbd_%s__wxy" % dn], dn)
    _close(e31, g["e2e_dual_mcma_sbd_%s__err1" % dn], dn)
    _close(e32, g["e2e_dual_mcma_sbd_%s__err2" % dn], dn)

    s64 = _sig(golden, "q64_2m", dn)
    wxy4, (e41, e42) = qampy_amd.equalisation.dual_mode_equalisation(s64, (1e-3, 5e-4), 41, Niter=(2, 2),
                                                                    methods=("cma", "mrde"), apply=False)
    _close(wxy4, g["e2e_dual_cma_mrde_%s__wxy" % dn], dn)
    _close(e41, g["e2e_dual_cma_mrde_%s__err1" % dn], dn)
    _close(e42, g["e2e_dual_cma_mrde_%s__err2" % dn], dn)

    out5, wxy5, err5 = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=11, method="cma_real", apply=True)
    assert wxy5.dtype == RT[dn] and wxy5.shape == (4, 4, 11) and out5.dtype == CT[dn]
    _close(np.asarray(out5), g["e2e_eq_cma_real_%s__out" % dn], dn)
    _close(wxy5, g["e2e_eq_cma_real_%s__wxy" % dn], dn)
    _close(err5, g["e2e_eq_cma_real_%s__err" % dn], dn)

    E = golden.input("q16_2m", CT[dn])
    wxy6, err6 = core_eq.equalise_signal(E, 2, 5e-4, 16, wxy=g["e2e_eq_mcma_%s__wxy" % dn].copy(), method="dd")
    _close(wxy6, g["e2e_core_dd_%s__wxy" % dn], dn)
    _close(err6, g["e2e_core_dd_%s__err" % dn], dn)

    sda = _sig(golden, "q16_2m_da", dn)
    wxy7, err7 = qampy_amd.equalisation.equalise_signal(sda, 2e-3, Ntaps=11, method="sbd_data", TrSyms=1000)
    _close(wxy7, g["e2e_eq_sbd_data_%s__wxy" % dn], dn)
    _close(err7, g["e2e_eq_sbd_data_%s__err" % dn], dn)


def test_real_taps_apply(golden, oracle_kernels):
    g = golden["apply"]
    for dn in ("c128", "c64"):
        E = golden.input("q16_2m", CT[dn])
        out = core_eq.apply_filter(E, 2, g["ap_realtaps_%s__wx" % dn])
        assert out.dtype == CT[dn]
        _close(out, g["ap_realtaps_%s__out" % dn], dn)


@pytest.mark.parametrize("case", [c for c in __import__("conftest").golden_cases("bps") if "base" in c], ids=lambda c: c["name"])
def test_bps_host_layer(golden, oracle_kernels, case):
    g = golden["bps"]
    dn = case["dtype"]
    E = g[case["base"] + "__E"].astype(CT[dn])
    sig = SignalQAM(E, case["M"], coded_symbols=g[case["base"] + "__alphabet"].astype(CT[dn]))
    Eout, ph = qampy_amd.phaserec.bps(sig, case["A"], case["N"])
    assert type(Eout) is SignalQAM and Eout.dtype == CT[dn] and ph.dtype == RT[dn] and ph.shape == E.shape
    assert np.array_equal(ph, g[case["name"] + "__ph"])                      # identical indices -> identical numpy ops
    assert np.array_equal(np.asarray(Eout), g[case["name"] + "__Eout"])
    N = case["N"]
    assert np.all(ph[:, :N] == ph[0, 0]) and np.all(ph[:, -N:] == ph[0, 0])  # edges keep angles[0] = -pi/4
    if case["name"] + "__Eout1d" in g.files:
        e1, p1 = core_ph.bps(E[0], case["A"], sig.coded_symbols, N)
        assert e1.ndim == 1 and np.array_equal(e1, g[case["name"] + "__Eout1d"]) and np.array_equal(p1, g[case["name"] + "__ph1d"])


@pytest.mark.parametrize("case", __import__("conftest").golden_cases("twostage"), ids=lambda c: c["name"])
def test_bps_twostage_host_layer(golden, oracle_kernels, case):
    g = golden["twostage"]
    dn = case["dtype"]
    E = g[case["base"] + "__E"].astype(CT[dn])
    sig = SignalQAM(E, case["M"], coded_symbols=g[case["base"] + "__alphabet"].astype(CT[dn]))
    Eout, ph = qampy_amd.phaserec.bps_twostage(sig, case["A"], case["N"], B=case["B"])
    assert type(Eout) is SignalQAM and Eout.dtype == CT[dn] and ph.dtype == RT[dn]
    assert np.array_equal(ph, g[case["name"] + "__ph"]) and np.array_equal(np.asarray(Eout), g[case["name"] + "__Eout"])
    e1, p1 = core_ph.bps_twostage(E[0], case["A"], sig.coded_symbols, case["N"], B=case["B"])
    assert e1.ndim == 1 and np.array_equal(p1, g[case["name"] + "__ph1d"]) and np.array_equal(e1, g[case["name"] + "__Eout1d"])


def test_bps_rejects_unknown_backend():
    with pytest.raises(ValueError):
        core_ph.bps(np.zeros(8, np.complex64), 4, np.ones(4, np.complex64), 2, method="af")


# ------------------------------------------------------------------------------------------------ signal stand-in + generator
def test_signal_object_contract():
    s = synth.make_capture(16, 256, nmodes=2, snr_db=20, seed=3)
    assert s.shape == (2, 512) and s.os == 2 and s.M == 16 and s.dtype == np.complex64
    assert s.symbols.shape == (2, 256) and s.coded_symbols.shape == (16,)
    r = s.recreate_from_np_array(np.zeros((2, 100), np.complex64), fs=s.fb)
    assert type(r) is SignalQAM and r.os == 1 and r.fb == s.fb and r.M == 16
    assert type(s * 2) is SignalQAM and (s * 2).M == 16                     # metadata survives ndarray arithmetic
    assert abs(np.mean(np.abs(s.symbols) ** 2) - 1) < 0.1


def test_ser_counter_finds_rotation_lag_and_swap():
    rng = np.random.default_rng(0)
    alphabet = theory.coded_symbols_qam(16)
    tx = alphabet[rng.integers(0, 16, size=(2, 4000))]
    rx = np.vstack([np.roll(tx[1], 7) * 1j, np.roll(tx[0], -3) * -1])      # swapped, rotated, delayed
    rx[0, 100] += 1.0                                                       # one forced error
    nerr, ncmp, mode, rot, lag = synth.count_symbol_errors(rx[0], tx, alphabet)
    assert mode == 1 and lag == 7 and nerr == 1 and ncmp == 4000 - 7
    nerr, ncmp, mode, rot, lag = synth.count_symbol_errors(rx[1], tx, alphabet)
    assert mode == 0 and lag == -3 and nerr == 0


def test_adaptive_flag_and_given_symbols():
    from qampy_amd.core.equalisation import hip_equalisation as hk
    assert hk._adaptive_flag(False) == 0 and hk._adaptive_flag(True) == 1 and hk._adaptive_flag(np.bool_(True)) == 1
    assert hk._adaptive_flag("per-mode") == 2 and hk._adaptive_flag("per_mode") == 2 and hk._adaptive_flag("Private") == 2
    with pytest.raises(ValueError):
        hk._adaptive_flag("sometimes")
    with pytest.raises(ValueError):
        hk._adaptive_flag("per-mode", allow_per_mode=False)
    # the host generator takes given symbols (cross-checks of the device generator, pilot frames)
    a = synth.make_capture(16, 512, nmodes=2, seed=1, dtype=np.complex128)
    b = synth.make_capture(16, 512, nmodes=2, seed=99, dtype=np.complex128, symbols=a.symbols)
    assert np.allclose(np.asarray(a), np.asarray(b)) and np.array_equal(b.symbols, a.symbols)


def test_pilot_signal_geometry():
    from qampy_amd.signals import PilotSignal
    idx, idx_dat, idx_pil = PilotSignal._cal_pilot_idx(256, 32, 8)
    assert idx_pil[:33].all() and not idx_pil[33] and idx_pil[40] and idx_pil.sum() == 32 + (256 - 32) // 8 and (idx_dat ^ idx_pil).all()
    with pytest.raises(ValueError):
        PilotSignal._cal_pilot_idx(256, 32, 5)
    pilots = np.ones((2, int(idx_pil.sum())), np.complex128)
    sig = PilotSignal(np.zeros((2, 2 * 256 * 2), np.complex128), 16, 1e9, 2e9, 256, 32, 8, pilots)
    assert sig.os == 2 and sig.nframes == 2 and sig.pilot_seq.shape == (2, 32) and sig.ph_pilots.shape == (2, 28)
    one = sig.recreate_from_np_array(np.zeros((2, 256), np.complex128), fs=sig.fb)
    assert type(one) is PilotSignal and one.os == 1 and one.nframes == 1 and one.get_data().shape == (2, 196) and one.extract_pilots().shape == (2, 60)
    with pytest.raises(ValueError):
        PilotSignal(np.zeros((2, 10), np.complex128), 16, 1e9, 2e9, 256, 32, 8, pilots[:, :5])


def test_automatic_segment_grid_follows_the_machine_model():
    """`qh_pit_auto_segments` (no GPU needed): segments of >= 0.45/mu (cold) / 0.4/mu (warm) steps, one wave per SIMD - whole rounds
    of 896 (cold: one CU per shader engine left to the basis build) or 1024 waves of 4-chain (<= 4096 chains) or 8-chain waves -
    and never shorter than the target (profiles/r03_segment_grid.txt)."""
    from qampy_amd.core.equalisation import hip_equalisation as hk
    n = 2 ** 22 - 40
    assert hk.pit_auto_segments(n, 2e-4, 2, True) == 1792          # C3 cma: 896 waves x 4 chains / 2 modes
    assert hk.pit_auto_segments(n, 2e-4, 2, False) == 2047         # C3 mrde: 2048-step segments, 1024 waves
    assert hk.pit_auto_segments(10 ** 7, 2e-4, 2, True) == 3584    # 10^7 symbols: 8-chain waves, 896 of them
    assert hk.pit_auto_segments(10 ** 7, 2e-4, 2, False) == 4096
    assert hk.pit_auto_segments(2 ** 20, 1e-3, 2, True) == 1792    # C2
    for n, mu, nsel, cold in [(2 ** 17, 2e-4, 2, True), (2 ** 16, 1e-3, 1, True), (4 * 10 ** 7, 2e-4, 2, False), (2 ** 21, 5e-4, 2, False),
                              (3 * 10 ** 6, 1e-4, 3, True), (2 ** 22, 1e-3, 4, False)]:
        S = hk.pit_auto_segments(n, mu, nsel, cold)
        target = (0.45 if cold else 0.4) / mu
        assert 1 <= S <= 65536
        if S > 1:
            assert n // S >= int(target), (n, mu, nsel, cold, S)                  # never shorter than the target
            chains = S * nsel
            waves = -(-chains // (4 if chains <= 4096 else 8))
            cap = 896 if cold else 1024
            assert waves <= cap or waves % cap == 0, (S, waves)                     # one round, or whole rounds
    assert hk.pit_auto_segments(1000, 1e-3, 2, True) == 1                          # too short to cut


def test_receiver_group_deals_passes_round_robin_and_hands_exceptions_to_the_caller():
    """pipeline.ReceiverGroup without a GPU: the dispatch logic on stand-in receivers - `steps` passes dealt round robin, each receiver on its own
    thread, the pending work completed before run() returns, a worker's exception raised in the caller, threads released on close()."""
    import threading
    from qampy_amd.pipeline import ReceiverGroup

    class Fake:
        def __init__(self, tag):
            self.tag, self.calls, self.threads, self.flushed = tag, [], set(), 0

        def run(self, overlap=False, mark=None):
            self.calls.append(overlap)
            self.threads.add(threading.get_ident())
            if mark:
                mark("apply")

        def wait_post(self, mark=None):
            self.flushed += 1

        def load(self, E):
            self.E = E

    synced, released = [], []
    g = ReceiverGroup(3, "x", factory=Fake, sync=lambda: synced.append(threading.get_ident()), release=lambda: released.append(threading.get_ident()))
    try:
        g.load("capture")
        assert all(r.E == "capture" for r in g.rx)
        g.run(7)
        assert [len(r.calls) for r in g.rx] == [3, 2, 2] and all(all(r.calls) for r in g.rx)
        assert all(r.flushed == 1 for r in g.rx)
        tids = [next(iter(r.threads)) for r in g.rx]
        assert len(set(tids)) == 3 and threading.get_ident() not in tids          # one thread per receiver, none of them the caller's
        marks = []
        g.run(4, overlap=False, mark=lambda i, k: (lambda name: marks.append((i, k, name))))
        assert sorted(marks) == [(0, 0, "apply"), (0, 1, "apply"), (1, 0, "apply"), (2, 0, "apply")]
        assert g.map(lambda r: r.tag) == ["x", "x", "x"] and g.map(lambda r: len(r.calls), which=[2]) == [3]
        with pytest.raises(KeyError):
            g.map(lambda r: {}["missing"])
        assert g.map(lambda r: 1) == [1, 1, 1]                                      # the group survives a failed job
    finally:
        g.close()
    assert sorted(released) == sorted(tids) and not any(t.is_alive() for t in g._threads)
    with pytest.raises(ValueError):
        ReceiverGroup(0, factory=Fake)
