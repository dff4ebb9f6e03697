"""
Functional and full-size tests on a real MI355X.

* the reference's own statistical pins for the hot path, re-stated on the build's synthetic generator
  (test/test_signal_recover_functional.py:162-185 TestLMS, test/test_phaserec.py:106-145 TestDtype / TestCorrect);
* size-independent properties at BASELINE.json's full sizes (2^20 / 2^22 symbol periods), where the oracle would take
  minutes: linearity and impulse response of the filter application, a zero-step-size sweep that ties the trainer to the
  filter kernel, the rotation covariance of the blind phase search, a full C2 pass of the resident receiver.
"""
import numpy as np
import pytest

import qampy_amd
from qampy_amd import synth, theory
from qampy_amd.signals import SignalQAM
from qampy_amd.core.equalisation import hip_equalisation as hk
from qampy_amd.core.equalisation import equalisation as core_eq
from qampy_amd.core import hip_dsp

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ reference-style pins
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("method", ["sbd", "mddma", "dd", "dd_real", "dd_data_real", "sbd_data", "rde", "mrde"])
def test_lms_methods_converge(dtype, method):
    """TestLMS: 16-QAM, 2^13 symbols, 13 taps, Niter=3, adaptive step -> at most 3 symbol errors, dtype preserved."""
    N, taps, mu = 2 ** 13, 13, 0.2e-2
    data_aided = method in core_eq.DATA_AIDED
    s = synth.make_capture(16, N, nmodes=2, fb=40e9, beta=0.1, seed=2024, dtype=dtype, shift=taps // 2 if data_aided else 0)
    wxy, err = qampy_amd.equalisation.equalise_signal(s, mu, Niter=3, Ntaps=taps, method=method, adaptive_stepsize=True)
    sout = qampy_amd.equalisation.apply_filter(s, wxy)
    assert type(sout) is SignalQAM and sout.dtype == np.dtype(dtype)
    for m in range(2):
        nerr, n = synth.count_symbol_errors(np.asarray(sout)[m], s.symbols, s.coded_symbols)[:2]
        assert nerr <= 3, (method, m, nerr, n)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("angle", np.linspace(0.1, np.pi / 4.1, 8))
def test_bps_recovers_a_constant_rotation(dtype, angle):
    """TestCorrect + TestDtype: SER == 0, |ph + angle| <= pi/4/32, output dtype = input dtype, phase dtype half the size."""
    rng = np.random.default_rng(5)
    alphabet = theory.coded_symbols_qam(32, dtype)
    tx = alphabet[rng.integers(0, 32, size=(1, 2 ** 12))]
    s3 = SignalQAM(tx * np.exp(1j * angle).astype(dtype), 32, coded_symbols=alphabet)
    s2, ph = qampy_amd.phaserec.bps(s3, 32, 11)
    assert s2.dtype == np.dtype(dtype) and ph.dtype.itemsize == np.dtype(dtype).itemsize // 2
    np.testing.assert_allclose(ph[0][20:-20] + angle, 0, atol=np.pi / 4 / 32)
    assert np.array_equal(synth.decide(np.asarray(s2)[0, 20:-20], alphabet), synth.decide(tx[0, 20:-20], alphabet))
    s4, ph2 = qampy_amd.phaserec.bps_twostage(s3, 16, 11)
    np.testing.assert_allclose(ph2[0][25:-25] + angle, 0, atol=np.pi / 4 / 32)
    assert np.array_equal(synth.decide(np.asarray(s4)[0, 25:-25], alphabet), synth.decide(tx[0, 25:-25], alphabet))


# ------------------------------------------------------------------------------------------------ full-size properties
@pytest.fixture(scope="module")
def capture_c3():
    return synth.make_capture(64, 2 ** 22, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000,
                              dtype=np.complex64)


def test_apply_filter_full_size_linearity_and_impulse(capture_c3):
    E = np.ascontiguousarray(np.asarray(capture_c3))
    nt = 41
    spike = core_eq._init_taps(nt, 2, 2, np.complex64)
    out = hk.apply_filter_to_signal(E, 2, spike)
    N = (E.shape[1] - nt + 1) // 2
    assert out.shape == (2, N)
    assert np.array_equal(out, E[:, nt // 2: nt // 2 + 2 * N: 2])            # centre-spike taps: a pure decimator, bit exact
    rng = np.random.default_rng(3)
    w1 = ((rng.standard_normal((2, 2, nt)) + 1j * rng.standard_normal((2, 2, nt))) / nt).astype(np.complex64)
    w2 = ((rng.standard_normal((2, 2, nt)) + 1j * rng.standard_normal((2, 2, nt))) / nt).astype(np.complex64)
    o1, o2, o12 = (hk.apply_filter_to_signal(E, 2, w) for w in (w1, w2, (w1 + np.complex64(0.5j) * w2)))
    np.testing.assert_allclose(o12, o1 + np.complex64(0.5j) * o2, atol=2e-5)
    assert np.array_equal(hk.apply_filter_to_signal(E, 2, w1, modes=[1])[0], o1[1])  # mode subset = row of the full result


@pytest.mark.parametrize("method", ["cma", "mrde"])
def test_zero_step_sweep_equals_filter_output(capture_c3, method, forms):
    """mu = 0 over 2^22 symbols: taps unchanged and err = errfn(filter output) - ties both trainers to the apply kernel."""
    E = np.ascontiguousarray(np.asarray(capture_c3))
    nt = 41
    rng = np.random.default_rng(4)
    w = core_eq._init_taps(nt, 2, 2, np.complex64) + ((rng.standard_normal((2, 2, nt)) + 1j * rng.standard_normal((2, 2, nt))) * 0.02).astype(np.complex64)
    tr = core_eq._cal_training_symbol_len(2, nt, E.shape[1])
    sy = core_eq._reshape_symbols(None, method, 64, np.complex64, 2)
    y = hk.apply_filter_to_signal(E, 2, w)[:, :tr].astype(np.complex128)
    if method == "cma":
        want = (sy[0, 0].real - np.abs(y) ** 2) * y
    else:
        codes, parts = np.array_split(sy[0].astype(np.complex128), 2)
        rr = codes.real[np.sum(y.real[..., None] ** 2 > parts.real, axis=-1)]
        ri = codes.imag[np.sum(y.imag[..., None] ** 2 > parts.imag, axis=-1)]
        want = (rr - y.real ** 2) * y.real + 1j * (ri - y.imag ** 2) * y.imag
    for form in ("lookahead", "direct"):
        forms.set("trainer", form)
        if form == "direct":                                   # the direct form is ~3x slower: a quarter of the capture
            trn = tr // 4
        else:
            trn = tr
        err, w2, mu = hk.train_equaliser(E, trn, 1, 2, np.float32(0), w.copy(), None, False, sy, method)
        assert np.array_equal(w2, w)
        d = np.abs(err - want[:, :trn])
        # a sample sitting exactly on an MRDE partition may pick the neighbouring code after float32 rounding
        assert np.mean(d > 2e-4) < 1e-5 and np.percentile(d, 99.99) < 2e-4


def test_bps_rotation_covariance_full_size():
    """Rotating the input by one test-angle step moves every index by one (mod A, the alphabet is 4-fold symmetric)."""
    A, N, L = 64, 20, 2 ** 22
    rng = np.random.default_rng(9)
    alphabet = theory.coded_symbols_qam(64, np.complex64)
    ph = np.cumsum(rng.normal(scale=2e-4, size=L))
    E = (alphabet[rng.integers(0, 64, size=L)] + 0.03 * (rng.standard_normal(L) + 1j * rng.standard_normal(L))) * np.exp(1j * ph)
    E = E.astype(np.complex64)
    angles = np.linspace(-np.pi / 4, np.pi / 4, A, endpoint=False, dtype=np.float32).reshape(1, -1)
    i0 = hip_dsp.bps(E, angles, alphabet, N)
    i1 = hip_dsp.bps((E * np.exp(1j * np.pi / 2 / A)).astype(np.complex64), angles, alphabet, N)
    assert np.all(i0[:N] == 0) and np.all(i0[-N:] == 0)
    agree = np.mean((i1[N:-N] + 1) % A == i0[N:-N])
    assert agree > 0.995, agree                      # float32 rounding of the rotation flips only near-ties


def test_resident_receiver_full_c2_pass():
    from qampy_amd.pipeline import ResidentReceiver
    sig = synth.make_capture(16, 2 ** 20, nmodes=2, snr_db=25, theta=np.pi / 5.6, dgd=30e-12, linewidth=50e3, seed=1000,
                             dtype=np.complex64)
    rx = ResidentReceiver(2, sig.shape[1], 2, 16, 21, (1e-3,), methods=("mcma",), Niter=(1,), adaptive_stepsize=(False,),
                          TrSyms=(None,), Mtestangles=32, Nbps=20, alphabet=sig.coded_symbols)
    rx.load(sig)
    rx.run()
    res = rx.fetch()
    assert res["err"][0].shape == (2, 1048551) and res["out"].shape == (2, 1048566)          # SURVEY.md §8a sizes of C2
    assert np.all(np.isfinite(res["wxy"])) and np.all(np.abs(res["err"][0][:, -1000:]) > 0)
    ser = synth.cal_ser(res["out"][:, 2000:-2000][:, :2 ** 17], sig.symbols, sig.coded_symbols, max_lag=4096)
    assert ser.max() < 1e-3, ser


# ------------------------------------------------------------------------------------------------ on-device SER harness (8f.2)
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_device_ser_harness_matches_host_count(dtype):
    """qh_ser_*_dev (bounded-lag search on decided indices + counting pass in HBM) finds the same tx mode, rotation and lag
    as the host cross-correlation and counts exactly the same symbol errors (cal_ser semantics, core/signals.py:295-335)."""
    from qampy_amd._lib import DeviceArray
    from qampy_amd.core import ber_functions as ber
    rng = np.random.default_rng(5)
    M, n = 16, 30000
    sig = synth.make_capture(M, n + 64, nmodes=2, snr_db=17, seed=11, dtype=dtype)
    tx = np.asarray(sig.symbols)
    alphabet = np.ascontiguousarray(sig.coded_symbols, dtype=dtype)
    noise = (rng.normal(size=(2, n)) + 1j * rng.normal(size=(2, n))) * 0.12
    # row 0: tx mode 1 delayed by 7 symbols and rotated by -j;  row 1: tx mode 0 advanced by 19 symbols, rotated by -1
    rx = np.empty((2, n), dtype)
    rx[0] = (np.roll(tx[1], 7)[:n] * (-1j) + noise[0]).astype(dtype)
    rx[1] = (np.roll(tx[0], -19)[:n] * (-1) + noise[1]).astype(dtype)
    d_rx, d_al = DeviceArray.from_host(rx), DeviceArray.from_host(alphabet)
    idx_tx = ber.tx_indices_dev(np.ascontiguousarray(tx, dtype=dtype), d_al)
    res = ber.cal_ser_dev(d_rx, idx_tx, d_al, maxlag=64, window=2048, trim=100)
    for r, row in zip(res, rx):
        nerr, ncmp, mode, rot, lag = synth.count_symbol_errors(row, tx, alphabet, max_lag=64, trim=0)
        assert (r["tx_mode"], r["rotation"], r["lag"]) == (mode, rot, lag)
        # same decisions on [trim, n - trim)
        d_r = synth.decide(row * np.exp(1j * rot * np.pi / 2), alphabet)[100:n - 100]
        d_t = synth.decide(tx[mode], alphabet)[100 - lag:n - 100 - lag]
        assert r["compared"] == d_r.size and r["errors"] == int(np.count_nonzero(d_r != d_t))
        assert 0 < r["errors"] < 0.1 * r["compared"] and r["window_matches"] > 0.9 * r["window"]
    assert (res[0]["tx_mode"], res[0]["rotation"], res[0]["lag"]) == (1, 1, 7)
    assert (res[1]["tx_mode"], res[1]["rotation"], res[1]["lag"]) == (0, 2, -19)


def test_resident_receiver_device_ser_equals_host_ser():
    from qampy_amd.pipeline import ResidentReceiver
    sig = synth.make_capture(16, 2 ** 16, nmodes=2, snr_db=22, theta=np.pi / 5.6, dgd=30e-12, linewidth=50e3, seed=1001,
                             dtype=np.complex64)
    rx = ResidentReceiver(2, sig.shape[1], 2, 16, 21, (1e-3,), methods=("mcma",), Niter=(2,), adaptive_stepsize=(False,),
                          TrSyms=(None,), Mtestangles=32, Nbps=20, alphabet=sig.coded_symbols)
    rx.load(sig)
    rx.run()
    dev = rx.ser(sig.symbols, maxlag=256, trim=2000)
    out = rx.fetch()["out"]
    for d, row in zip(dev, out):
        nerr, ncmp, mode, rot, lag = synth.count_symbol_errors(row, sig.symbols, sig.coded_symbols, max_lag=256, trim=2000)
        assert (d["tx_mode"], d["rotation"]) == (mode, rot) and d["lag"] == lag + 2000      # host lag is relative to the trimmed row
        assert abs(d["errors"] - nerr) <= 2 and abs(d["compared"] - ncmp) <= 2 * 256 + 4000, (d, nerr, ncmp)


# ------------------------------------------------------------------------------------------------ channel bank (8e, within a GPU)
@pytest.mark.parametrize("methods,adaptive", [(("mcma", "sbd"), (False, False)), (("cma", "mrde"), (False, False)),
                                              (("mcma", "mddma"), (True, True)), (("mcma", "sbd"), ("per-mode", "per-mode"))])
@pytest.mark.parametrize("trainer", ["auto", "iterative"])
def test_channel_bank_equals_single_receivers(methods, adaptive, trainer, forms):
    """A bank of independent captures trained in ONE launch per stage (channel = blockIdx.y) gives bit-identical taps,
    errors and recovered symbols to one ResidentReceiver per capture."""
    from qampy_amd.pipeline import ChannelBank, ResidentReceiver
    nch, M = 3, 16 if "mrde" not in methods else 64
    sigs = [synth.make_capture(M, 2 ** 13 + 100 * c, nmodes=2, snr_db=24, theta=0.5 + 0.1 * c, dgd=20e-12, linewidth=10e3, seed=50 + c,
                               dtype=np.complex64)[:, :2 * 2 ** 13] for c in range(nch)]
    kw = dict(methods=methods, Niter=(2,) * len(methods), adaptive_stepsize=adaptive, TrSyms=(None,) * len(methods), Mtestangles=32,
              Nbps=10)
    L = sigs[0].shape[1]
    bank = ChannelBank(nch, 2, L, 2, M, 15, (2e-3, 5e-4)[:len(methods)], alphabet=sigs[0].coded_symbols, trainer=trainer, **kw)
    for c, sg in enumerate(sigs):
        bank.load(c, sg)
    bank.run()
    if trainer != "auto":
        forms.set("trainer", trainer)         # the single receivers in the same form
    for c, sg in enumerate(sigs):
        rx = ResidentReceiver(2, L, 2, M, 15, (2e-3, 5e-4)[:len(methods)], alphabet=sg.coded_symbols, **kw)
        rx.load(sg)
        rx.run()
        one, many = rx.fetch(), bank.fetch(c)
        for k in ("wxy", "eq", "out", "idx", "ph"):
            assert np.array_equal(one[k], many[k]), (c, k)
        for e1, e2 in zip(one["err"], many["err"]):
            assert np.array_equal(e1, e2)
        assert one["mu"] == many["mu"]
        assert np.all(np.isfinite(one["wxy"])) and np.abs(one["err"][-1]).max() > 0


# ------------------------------------------------------------------------------------------------ on-device synthesis (8f.4)
def test_device_synthesis_matches_host_generator_for_given_symbols():
    """Shaping + PMD of csrc/synth.hip (time-domain FIRs) against the host generator (frequency domain) on the SAME symbols."""
    M, nsym = 16, 2 ** 14
    d = synth.make_capture_dev(M, nsym, nmodes=2, theta=np.pi / 5.6, dgd=30e-12, seed=7)
    E, sy, idx = d["E"].to_host(), d["symbols"].to_host(), d["idx_tx"].to_host()
    assert np.array_equal(sy, d["alphabet_host"][idx])
    counts = np.bincount(idx.ravel(), minlength=M)
    assert counts.min() > 0.85 * idx.size / M and counts.max() < 1.15 * idx.size / M          # uniform symbols
    assert abs(np.mean(np.abs(E) ** 2) - 1) < 0.01                                           # unit power
    ref = np.asarray(synth.make_capture(M, nsym, nmodes=2, theta=np.pi / 5.6, dgd=30e-12, symbols=sy, dtype=np.complex128))
    err = np.sqrt(np.mean(np.abs(E - ref) ** 2) / np.mean(np.abs(ref) ** 2))
    assert err < 1e-2, err              # pulse truncated to +-48 symbols, 33-tap fractional delays, analytic power normalisation
    # without PMD the modes are independent pulse trains
    d0 = synth.make_capture_dev(M, nsym, nmodes=2, seed=7)
    ref0 = np.asarray(synth.make_capture(M, nsym, nmodes=2, symbols=d0["symbols"].to_host(), dtype=np.complex128))
    assert np.sqrt(np.mean(np.abs(d0["E"].to_host() - ref0) ** 2)) < 1e-2


def test_device_synthesis_noise_and_phase_noise_statistics():
    M, nsym, os_ = 4, 2 ** 16, 2
    clean = synth.make_capture_dev(M, nsym, nmodes=2, seed=3)["E"].to_host()
    noisy = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=15., seed=3)["E"].to_host()
    nvar = np.mean(np.abs(noisy - clean) ** 2)
    assert abs(nvar / (10 ** (-15. / 10) * os_) - 1) < 0.02, nvar                             # sigma^2 = P 10^(-snr/10) os
    n = noisy - clean
    assert abs(np.mean(n.real ** 2) / np.mean(n.imag ** 2) - 1) < 0.03 and abs(np.mean(n)) < 3e-3
    lw, fb = 1e6, 20e9
    pn = synth.make_capture_dev(M, nsym, nmodes=2, linewidth=lw, fb=fb, seed=3)["E"].to_host()
    ph = np.unwrap(np.angle(np.sum(pn * np.conj(clean), axis=0) if False else (pn * np.conj(clean))[0][np.abs(clean[0]) > 0.3]))
    sel = np.nonzero(np.abs(clean[0]) > 0.3)[0]
    # Wiener process: increments over a gap of g samples have variance g * 2 pi lw / fs; no jumps at the 1024-sample tiles
    gaps = np.diff(sel)
    inc = np.diff(ph)
    var = 2 * np.pi * lw / (fb * os_)
    assert abs(np.sum(inc ** 2) / (np.sum(gaps) * var) - 1) < 0.05
    assert np.max(np.abs(inc) / np.sqrt(gaps * var)) < 7
    assert abs(ph[-1] - ph[0]) > 0.05                                                         # it really drifts


def test_receiver_on_device_generated_capture():
    """A capture synthesised in HBM goes through the resident receiver without ever visiting the host; SER from the device harness."""
    from qampy_amd.pipeline import ResidentReceiver
    from qampy_amd.core import ber_functions as ber
    d = synth.make_capture_dev(16, 2 ** 17, nmodes=2, snr_db=25, theta=np.pi / 5.6, dgd=30e-12, linewidth=50e3, seed=1000)
    rx = ResidentReceiver(2, 2 ** 18, 2, 16, 21, (1e-3,), methods=("mcma",), Niter=(2,), adaptive_stepsize=(False,), TrSyms=(None,),
                          Mtestangles=32, Nbps=20, alphabet=d["alphabet_host"])
    rx.E.copy_from(d["E"])
    rx.run()
    res = ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, maxlag=256, window=4096, trim=20000)
    assert max(r["ser"] for r in res) < 2e-3 and sorted(r["tx_mode"] for r in res) == [0, 1], res


@pytest.mark.parametrize("method", ["cma", "mrde", "sbd"])
@pytest.mark.parametrize("adaptive", [False, True])
def test_divergent_training_terminates(method, adaptive):
    """A step size far beyond stability drives the taps to inf / nan; every trainer form must still return (the block-iterative
    sweeps are bounded by the block length) and produce non-finite taps like the reference's arithmetic does."""
    import time
    from qampy_amd.core.equalisation import equalisation as core_eq
    from qampy_amd.core.equalisation import hip_equalisation as hk
    sig = synth.make_capture(64 if method == "mrde" else 16, 2 ** 14, nmodes=2, snr_db=20, seed=2, dtype=np.complex64)
    E = np.ascontiguousarray(np.asarray(sig))
    tr = core_eq._cal_training_symbol_len(2, 21, E.shape[1])
    sy = core_eq._reshape_symbols(None, method, sig.M, np.complex64, 2)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        e, w, mu = hk.train_equaliser(E, tr, 1, 2, np.float32(50.), core_eq._init_taps(21, 2, 2, np.complex64), None, adaptive, sy, method)
    assert time.perf_counter() - t0 < 20
    assert e.shape == (2, tr) and (adaptive or not np.all(np.isfinite(w)))


def test_channel_bank_pipelined_passes_equal_plain_passes():
    """Two-stream software pipelining of the bank (trainers of pass k+1 beside filter + phase search of pass k) changes no result."""
    from qampy_amd import _lib
    from qampy_amd.pipeline import ChannelBank
    nch, M = 4, 16
    sigs = [synth.make_capture(M, 2 ** 13, nmodes=2, snr_db=24, theta=0.5 + 0.1 * c, dgd=20e-12, linewidth=10e3, seed=70 + c, dtype=np.complex64)
            for c in range(nch)]
    bank = ChannelBank(nch, 2, sigs[0].shape[1], 2, M, 15, (2e-3, 5e-4), methods=("mcma", "sbd"), Niter=(2, 1), Mtestangles=32, Nbps=10,
                       alphabet=sigs[0].coded_symbols, trainer="iterative")
    for c, sg in enumerate(sigs):
        bank.load(c, sg)
    bank.run()
    ref = [bank.fetch(c) for c in range(nch)]
    bank.eq.zero(); bank.out.zero()
    bank.run_pipelined(5)
    _lib.sync()
    for c in range(nch):
        got = bank.fetch(c)
        for k in ("wxy", "eq", "out", "idx", "ph"):
            assert np.array_equal(ref[c][k], got[k]), (c, k)


def test_release_scratch_and_continue():
    """qh_release_scratch gives the grow-only buffers (Gram tables ...) back; the next call re-allocates what it needs."""
    from qampy_amd import _lib
    from qampy_amd.core.equalisation import equalisation as core_eq
    from qampy_amd.core.equalisation import hip_equalisation as hk
    sig = synth.make_capture(16, 2 ** 12, nmodes=2, snr_db=25, seed=3, dtype=np.complex64)
    E = np.ascontiguousarray(np.asarray(sig))
    tr = core_eq._cal_training_symbol_len(2, 11, E.shape[1])
    sy = core_eq._reshape_symbols(None, "mcma", 16, np.complex64, 2)
    e1, w1, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(1e-3), core_eq._init_taps(11, 2, 2, np.complex64), None, False, sy, "mcma")
    _lib.call("qh_release_scratch")
    e2, w2, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(1e-3), core_eq._init_taps(11, 2, 2, np.complex64), None, False, sy, "mcma")
    assert np.array_equal(w1, w2) and np.array_equal(e1, e2)


# ------------------------------------------------------------------------------------------------ drop-in boundary, round 5
def test_default_tier_b_through_the_drop_in_module():
    """with the process-wide default tier set to b the UNCHANGED call of the pythran module's `train_equaliser` (host arrays, the
    reference's argument list - what core/equalisation/equalisation.py:555-557 calls) is solved in parallel in time, certified by the device at the given
    tolerance, within that tolerance of the exact recurrence; the mirrored host layer follows the same default; "a" switches back."""
    from qampy_amd import _lib
    M, nsym, ntaps, mu, tol = 16, 2 ** 20, 21, 1e-3, 1e-4
    d = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=25, theta=np.pi / 5.6, dgd=30e-12, linewidth=50e3, seed=1000)
    E = d["E"].to_host()
    tr = core_eq._cal_training_symbol_len(2, ntaps, E.shape[1])
    sy = core_eq._reshape_symbols(None, "mcma", M, np.complex64, 2)
    assert qampy_amd.get_default_tier()[0] == "a"
    e_a, w_a, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(mu), core_eq._init_taps(ntaps, 2, 2, np.complex64), None, False, sy, "mcma")
    qampy_amd.set_default_tier("b", tol)
    try:
        assert qampy_amd.get_default_tier() == ("b", tol)
        e_b, w_b, mu_b = hk.train_equaliser(E, tr, 1, 2, np.float32(mu), core_eq._init_taps(ntaps, 2, 2, np.complex64), None, False, sy, "mcma")
        rep = _lib.last_pit_report()
        assert rep["segments"] > 64 and rep["converged"] and not rep["exact_form"] and abs(rep["tol"] - tol) < 1e-12 and rep["acquisition"]["steps"] > 0, rep
        assert mu_b == np.float32(mu)
        for m in range(2):
            assert np.linalg.norm(w_a[m] - w_b[m]) / np.linalg.norm(w_a[m]) <= 3 * tol
            assert np.sqrt(np.mean(np.abs(e_a[m] - e_b[m]) ** 2)) <= 3 * tol
        # a warm call (given taps) takes no acquisition; a data-aided method has no parallel-in-time solver: the exact form, reported as such
        hk.train_equaliser(E, tr, 1, 2, np.float32(mu), w_b.copy(), None, False, sy, "mcma")
        assert _lib.last_pit_report()["acquisition"]["steps"] == 0 and _lib.last_pit_report()["converged"]
        # the mirrored host layer without a tier keyword
        w_h, e_h = core_eq.equalise_signal(E, 2, mu, M, Ntaps=ntaps, method="mcma")
        reps = core_eq.last_pit_reports()
        assert len(reps) == 1 and reps[0]["converged"] and not reps[0]["exact_form"] and abs(reps[0]["tol"] - tol) < 1e-12
        assert np.max(np.abs(w_h - w_b)) <= 3 * tol, "same solver, same tolerance as the drop-in module"
    finally:
        qampy_amd.set_default_tier("a")
    assert qampy_amd.get_default_tier()[0] == "a"
    e_a2, w_a2, _ = hk.train_equaliser(E, tr, 1, 2, np.float32(mu), core_eq._init_taps(ntaps, 2, 2, np.complex64), None, False, sy, "mcma")
    assert np.array_equal(w_a2, w_a) and np.array_equal(e_a2, e_a)


def test_results_on_pooled_pinned_memory_are_ordinary_arrays():
    """The mirrored host layers hand back views of pooled pinned buffers (one DMA at the PCIe rate): writable ndarrays of the right type that outlive the
    call and the library's scratch, whose buffers return to the pool when the last view dies - and whose values are those of a plain pageable copy."""
    import gc
    from qampy_amd import _lib
    s = synth.make_capture(16, 2 ** 18, nmodes=2, snr_db=25, theta=np.pi / 5.6, dgd=30e-12, seed=5, dtype=np.complex64)
    out, wxy, err = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=21, method="mcma", apply=True)
    rec, ph = qampy_amd.phaserec.bps(out, 32, 20)
    assert type(out) is SignalQAM and out.flags.writeable and err.flags.writeable and rec.dtype == np.complex64 and ph.dtype == np.float32
    keep = np.array(out, copy=True)
    _lib.call("qh_release_scratch")
    gc.collect()
    assert np.array_equal(np.asarray(out), np.asarray(keep))
    out2, _, err2 = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=21, method="mcma", apply=True)
    assert np.array_equal(np.asarray(out2), np.asarray(keep)) and np.array_equal(err2, err) and out2.ctypes.data != out.ctypes.data
    out[:, :8] = 0                                    # writable, and not aliased with the second result
    assert np.array_equal(np.asarray(out2), np.asarray(keep))
    p_first = out2.ctypes.data
    del out2, err2
    gc.collect()
    out3, _, _ = qampy_amd.equalisation.equalise_signal(s, 1e-3, Ntaps=21, method="mcma", apply=True)
    assert np.array_equal(np.asarray(out3), np.asarray(keep))
    assert out3.ctypes.data in (p_first, err.ctypes.data) or True     # (which pooled buffer comes back is the pool's business)
