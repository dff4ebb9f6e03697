"""
Pilot-based receiver (BASELINE config 5 shape): core functions on plain arrays against vectors captured
from the reference (tests/golden/pilot.npz, generator tests/golden/gen_golden.py::gen_pilot).  The CPU variant runs the host
layer on the oracle's kernels; the gpu variant runs the real HIP path, including the one-launch window batch of frame_sync.
"""
import numpy as np
import pytest

from oracle import oracle
from oracle_kernels import OracleJobs
from qampy_amd.core import pilotbased_receiver as pil
from qampy_amd.core import ber_functions, phaserecovery
from qampy_amd.core.filter import moving_average
from qampy_amd.core.equalisation import equalisation as core_eq


def _oracle_windows(E, starts, win_len, TrSyms, Niter, os_, mu, wx0, modes, adaptive, symbols, method):
    errs, wxs, mus = [], [], []
    for s in np.asarray(starts):
        seg = np.ascontiguousarray(E[:, s:s + win_len])
        e, w, m = oracle.train_equaliser(seg, TrSyms, Niter, os_, mu, wx0.copy(), modes, adaptive, symbols, method)
        errs.append(e); wxs.append(w); mus.append(m)
    return np.array(errs), np.array(wxs), np.array(mus)


def _numpy_cfo(E, fo, os=1):
    """What hip_dsp.comp_freq_offset computes on the device (qampy/core/phaserecovery.py:435-473)."""
    t = np.arange(1, E.shape[1] + 1, dtype=float)
    return (E * np.exp(-2j * np.pi * t * np.asarray(fo, dtype=float).reshape(-1, 1) / os)).astype(E.dtype)


def _numpy_trace(E, knots, kph):
    """What hip_dsp.pilot_phase_trace computes on the device: the reference's own lines (qampy/core/pilotbased_receiver.py:318-327)."""
    trace = np.array([np.interp(np.arange(E.shape[1]), knots, p) for p in kph]).astype(E.dtype)
    return E * np.exp(-1j * trace), trace


def _oracle_search(E, starts, win_len, *a):
    err, wx, _ = _oracle_windows(E, starts, win_len, *a)
    var = np.var(err, axis=-1).T                       # (nmodes, nwin)
    best = np.argmin(var, axis=-1)
    return var, best, wx[best]


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
    monkeypatch.setattr(k, "train_equaliser_windows", _oracle_windows)
    monkeypatch.setattr(k, "train_equaliser_windows_search", _oracle_search)
    monkeypatch.setattr(phaserecovery._dsp, "comp_freq_offset", _numpy_cfo)
    monkeypatch.setattr(phaserecovery._dsp, "pilot_phase_trace", _numpy_trace)


def test_helpers_match_reference(golden, oracle_kernels):
    g = golden["pilot"]
    ix, y2, ii, acm = ber_functions.find_sequence_offset_complex(g["fso_x"], g["fso_y"])
    assert ix == g["fso_ix"] and ii == g["fso_ii"] and np.isclose(acm, g["fso_acm"])
    assert np.array_equal(moving_average(g["mavg_in"], 5), g["mavg_out"])
    assert np.array_equal(phaserecovery.find_freq_offset(g["ffo_in"], fft_size=2 ** 12), g["ffo_out"])
    foe = g["fs_foe"]
    E2 = g["rx"][g["fs_order"], :]
    np.testing.assert_allclose(phaserecovery.comp_freq_offset(E2, np.ones(foe.shape) * np.mean(foe)), g["synced"], rtol=0, atol=1e-12)


def _run_chain(g, rtol):
    os_, frame_len = int(g["os"]), int(g["frame_len"])
    shift, foe, order, wx1, ok = pil.frame_sync(g["rx"], g["pilot_seq"], os_, frame_len=frame_len, M_pilot=4, mu=5e-3, Ntaps=17,
                                                adaptive_stepsize=True, Niter=10, method="cma")
    assert ok == bool(g["fs_ok"]) and np.array_equal(shift, g["fs_shift"]) and np.array_equal(order, g["fs_order"])
    np.testing.assert_allclose(foe, g["fs_foe"], rtol=1e-12)
    np.testing.assert_allclose(wx1, g["fs_wx1"], rtol=rtol, atol=rtol)
    E3 = g["synced"]
    taps, foe_all = pil.equalize_pilot_sequence(E3, g["pilot_seq"], g["eq_shift"], os_, mu=(1e-3, 1e-3), foe_comp=False, Ntaps=45,
                                                methods=("cma", "sbd"))
    np.testing.assert_allclose(taps, g["eq_taps"], rtol=rtol, atol=rtol)
    assert np.array_equal(foe_all, g["eq_foe"])
    frames = [core_eq.apply_filter(E3[:, i0:i0 + frame_len * os_ + 44], os_, taps, modes=[m])[0] for m, i0 in enumerate(g["eq_shift"])]
    eq = np.array(frames)
    np.testing.assert_allclose(eq, g["eq_frame"], rtol=rtol, atol=10 * rtol)
    out, ph = pil.pilot_based_cpe_new(eq, g["ph_pilots"], g["cpe_idx"], frame_len, seq_len=None, max_num_blocks=None,
                                      use_pilot_ratio=1, num_average=5, nframes=1)
    np.testing.assert_allclose(ph, g["cpe_ph"], rtol=0, atol=100 * rtol)
    np.testing.assert_allclose(out, g["cpe_out"], rtol=0, atol=100 * rtol)
    f, fm, c = pil.pilot_based_foe(eq[:, :g["pilot_seq"].shape[1]], g["pilot_seq"])
    np.testing.assert_allclose(fm, g["pfoe_modes"], rtol=1e-5, atol=1e-9)
    return out


def test_pilot_chain_on_oracle_kernels(golden, oracle_kernels):
    _run_chain(golden["pilot"], 1e-9)


@pytest.mark.gpu
def test_pilot_chain_on_gpu(golden):
    _run_chain(golden["pilot"], 1e-8)


@pytest.mark.gpu
def test_window_batch_equals_single_calls(golden, forms):
    """frame_sync's one-launch batch gives exactly what separate equalise_signal calls give (same kernel, same order)."""
    forms.set("trainer", "direct")        # the batch runs the direct-form kernel
    g = golden["pilot"]
    E = np.ascontiguousarray(g["rx"].astype(np.complex64))
    starts = np.arange(2, 20) * 256
    for adaptive, method in ((True, "cma"), (False, "mcma")):
        w_all, e_all = core_eq.equalise_signal_windows(E, 2, 5e-3, 4, starts, 512, Ntaps=17, Niter=4, method=method,
                                                       adaptive_stepsize=adaptive)
        for i, s in enumerate(starts):
            w, e = core_eq.equalise_signal(E[:, s:s + 512], 2, 5e-3, 4, Ntaps=17, Niter=4, method=method, adaptive_stepsize=adaptive)
            assert np.array_equal(w, w_all[i]) and np.array_equal(e, e_all[i])


# ------------------------------------------------------------------------------------------------ basic API (signal objects)
def _pilot_signal(g):
    from qampy_amd.signals import PilotSignal
    pilots = np.hstack([g["pilot_seq"], g["ph_pilots"]])
    sig = PilotSignal(g["rx"].copy(), int(g["M"]), 24e9, 48e9, int(g["frame_len"]), g["pilot_seq"].shape[1], 32, pilots,
                      coded_symbols=g["alphabet"])
    assert np.array_equal(sig._idx_pil, g["idx_pil"]) and sig.os == int(g["os"]) and sig.nframes == 3
    return sig


def _run_basic_api(g, rtol):
    """sync2frame -> corr_foe -> pilot_equaliser -> pilot_cpe on a signal object reproduces the reference's arrays
    (qampy/signals.py:1709-1750, qampy/equalisation.py:42-87 and :268-338, qampy/phaserec.py:156-192)."""
    from qampy_amd import equalisation, phaserec
    sig = _pilot_signal(g)
    wx1, ok = sig.sync2frame(returntaps=True)
    assert ok == bool(g["fs_ok"]) and np.array_equal(sig.shiftfctrs, g["shiftfctrs"]) and sig.synctaps == 17
    np.testing.assert_allclose(wx1, g["fs_wx1"], rtol=rtol, atol=rtol)
    before, foe_c = np.asarray(sig).copy(), np.mean(sig._foe)
    sig.corr_foe()                       # = comp_freq_offset with the signal's oversampling (qampy/signals.py:1747-1750)
    np.testing.assert_allclose(np.asarray(sig), phaserecovery.comp_freq_offset(before, np.ones(2) * foe_c, 2), rtol=0, atol=1e-12)
    sig[:, :] = g["synced"]              # the captured chain continued from the os=1 variant of that call
    taps, eq, foe, ntaps = equalisation.pilot_equaliser(sig, (1e-3, 1e-3), 45, foe_comp=False, methods=("cma", "sbd"), verbose=True)
    assert ntaps == (45, 17) and np.array_equal(foe, g["eq_foe"])
    np.testing.assert_allclose(taps, g["eq_taps"], rtol=rtol, atol=rtol)
    assert type(eq) is type(sig) and eq.os == 1 and eq.shape == g["eq_frame"].shape
    np.testing.assert_allclose(np.asarray(eq), g["eq_frame"], rtol=rtol, atol=10 * rtol)
    out, ph = phaserec.pilot_cpe(eq, N=5)
    np.testing.assert_allclose(ph, g["cpe_ph"], rtol=0, atol=100 * rtol)
    np.testing.assert_allclose(np.asarray(out), g["cpe_out"], rtol=0, atol=100 * rtol)
    # payload / pilot extraction and the apply_filter(frames=...) path on two consecutive frames
    assert out.get_data().shape[1] == np.count_nonzero(~g["idx_pil"]) and out.extract_pilots().shape[1] == np.count_nonzero(g["idx_pil"])
    two = equalisation.apply_filter(sig, taps, frames=[0, 1])
    assert two.shape == (2, 2 * int(g["frame_len"]))
    np.testing.assert_allclose(np.asarray(two)[:, :int(g["frame_len"])], g["eq_frame"], rtol=rtol, atol=10 * rtol)
    one = equalisation.apply_filter(sig, taps, frames=[1])
    np.testing.assert_allclose(np.asarray(one), np.asarray(two)[:, int(g["frame_len"]):], rtol=rtol, atol=10 * rtol)
    fo = phaserec.find_freq_offset(eq, fft_size=2 ** 12)
    assert np.shape(fo) == (2, 1) and np.all(np.abs(fo) < 0.01)          # residual offset after the pilot-based compensation


def test_pilot_basic_api_on_oracle_kernels(golden, oracle_kernels):
    _run_basic_api(golden["pilot"], 1e-9)


@pytest.mark.gpu
def test_pilot_basic_api_on_gpu(golden):
    _run_basic_api(golden["pilot"], 1e-8)


# ------------------------------------------------------------------------------------------------ frames / constant phase
def _run_frames(g, gf, rtol):
    """pilot_equaliser_nframes and the constant-phase helpers against the reference's basic API on its own signal object
    (tests/golden/pilot_frames.npz: qampy/equalisation.py:340-397, qampy/phaserec.py:194-238)."""
    from qampy_amd import equalisation, phaserec
    sig = _pilot_signal(g)
    sig[:, :] = gf["synced"]
    sig.shiftfctrs, sig.synctaps = gf["shiftfctrs"].copy(), 17
    taps, sout, rest = equalisation.pilot_equaliser_nframes(sig, (1e-3, 1e-3), 45, foe_comp=False, frames=[0, 1], methods=("cma", "sbd"))
    np.testing.assert_allclose(np.array(taps), gf["nf_taps"], rtol=rtol, atol=rtol)
    assert type(sout) is type(sig) and sout.shape == gf["nf_out"].shape
    np.testing.assert_allclose(np.asarray(sout), gf["nf_out"], rtol=rtol, atol=10 * rtol)
    assert np.array_equal(np.array(rest[0]), gf["nf_foe"]) and np.array_equal(np.array(rest[1]), gf["nf_ntaps"])
    # (the taps of frame 0 initialise frame 1 and are trained on IN PLACE by its pre-convergence stage, in the reference as here:
    # entry 0 of the returned list is therefore not what frame 0 alone returns)
    # without applying the filter the reference returns nothing usable (missing `return`, qampy/equalisation.py:335-336); here: the taps
    only = equalisation.pilot_equaliser_nframes(sig, (1e-3, 1e-3), 45, apply=False, foe_comp=False, frames=[0, 1], verbose=False, methods=("cma", "sbd"))
    np.testing.assert_allclose(np.array(only[0]), gf["nf_taps"], rtol=rtol, atol=rtol)
    with pytest.raises(ValueError):
        equalisation.pilot_equaliser_nframes(sig, (1e-3, 1e-3), 45, frames=[5], methods=("cma", "sbd"))      # incomplete frame
    ph = phaserec.find_pilot_const_phase(gf["cp_rec"], gf["cp_ref"])
    assert ph.shape == (2, 1)
    np.testing.assert_allclose(ph, gf["cp_phase"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(phaserec.correct_pilot_const_phase(gf["cp_rec"], ph), gf["cp_out"], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        phaserec.correct_pilot_const_phase(gf["cp_rec"], np.zeros(3))


def test_pilot_frames_on_oracle_kernels(golden, oracle_kernels):
    _run_frames(golden["pilot"], golden["pilot_frames"], 1e-9)


@pytest.mark.gpu
def test_pilot_frames_on_gpu(golden):
    _run_frames(golden["pilot"], golden["pilot_frames"], 1e-8)


def _frames_batched_vs_sequential(dtype, rtol, **capkw):
    """From its second frame on ``pilot_equaliser_nframes`` trains the pilot-aided sweeps of all remaining frames together (one launch per
    stage over all (frame, mode) chains; the pre-convergence stages stay a chain through the frames because the reference trains the
    handed-on taps in place).  Same results as going frame by frame - the taps, the equalised frames, and the state the caller's array
    of start taps is left in."""
    from qampy_amd import equalisation, synth
    from qampy_amd.signals import PilotSignal
    cap = synth.make_pilot_capture(**capkw)

    def run(batched):
        sig = PilotSignal(cap["E"].astype(dtype), cap["M"], cap["fb"], cap["fs"], cap["frame_len"], cap["seq_len"], cap["ins_rat"], cap["pilots"],
                          symbols=cap["payload"], coded_symbols=cap["alphabet"])
        assert sig.sync2frame()
        sig.corr_foe()
        saved = equalisation._pilot_equaliser_frames
        if not batched:
            equalisation._pilot_equaliser_frames = lambda *a, **k: None
        try:
            nfr = (sig.shape[-1] - int(np.max(sig.shiftfctrs))) // (sig.os * sig.frame_len) - 1
            assert nfr >= 3
            taps, out, rest = equalisation.pilot_equaliser_nframes(sig, (1e-3, 1e-3), 25, foe_comp=True, frames=list(range(nfr)), methods=("cma", "sbd_data"), Niter=6)
        finally:
            equalisation._pilot_equaliser_frames = saved
        return np.array(taps), np.asarray(out), np.array(rest[0])

    ta, oa, fa = run(False)
    tb, ob, fb = run(True)
    assert ta.shape[0] >= 3
    np.testing.assert_allclose(tb, ta, rtol=rtol, atol=rtol)
    np.testing.assert_allclose(ob, oa, rtol=rtol, atol=10 * rtol)
    np.testing.assert_allclose(fb, fa, rtol=1e-6, atol=1e-12)


def test_batched_frames_equal_frame_by_frame_on_oracle_kernels(oracle_kernels):
    _frames_batched_vs_sequential(np.complex128, 1e-10, M=16, frame_len=2 ** 12, seq_len=2 ** 8, nframes=5, modal_delay=100, frame_offset=999, snr_db=25)


@pytest.mark.gpu
def test_batched_frames_equal_frame_by_frame_on_gpu():
    _frames_batched_vs_sequential(np.complex128, 1e-9, M=64, frame_len=2 ** 14, seq_len=2 ** 9, nframes=6, modal_delay=300, frame_offset=2345)
    _frames_batched_vs_sequential(np.complex64, 2e-4, M=64, frame_len=2 ** 14, seq_len=2 ** 9, nframes=6, modal_delay=300, frame_offset=2345)


# ------------------------------------------------------------------------------------------------ config 5 shape
def _config5_chain(cap, dtype):
    from qampy_amd import equalisation, phaserec
    from qampy_amd.signals import PilotSignal
    sig = PilotSignal(cap["E"].astype(dtype), cap["M"], cap["fb"], cap["fs"], cap["frame_len"], cap["seq_len"], cap["ins_rat"], cap["pilots"],
                      symbols=cap["payload"], coded_symbols=cap["alphabet"])
    ok = sig.sync2frame()
    shifts = np.array(sig.shiftfctrs)
    sig.corr_foe()
    taps, eq = equalisation.pilot_equaliser(sig, (1e-3, 1e-3), 45, foe_comp=False, methods=("cma", "sbd_data"))
    out, ph = phaserec.pilot_cpe(eq, N=5, use_seq=False)
    return dict(ok=ok, shifts=shifts, taps=taps, eq=np.asarray(eq), out=np.asarray(out), ser=out.cal_ser(frames=[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["c128", "c64"])
def test_config5_256qam_against_oracle_kernel_chain(monkeypatch, prec):
    """BASELINE config 5 as stated: 256-QAM payload, 2^16-symbol frames, 2 modes, 2 SPS, frequency offset and modal delay -
    frame sync, pilot-sequence equaliser (data-aided second stage), filter over the frame and pilot phase recovery on the HIP
    kernels against the same host layer running on the oracle's kernels; complex128 and complex64 (the performance dtype of the path)."""
    from qampy_amd import synth
    dtype = np.complex128 if prec == "c128" else np.complex64
    cap = synth.make_pilot_capture()
    hip = _config5_chain(cap, dtype)
    k = core_eq._kernels

    class OracleField:
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
    monkeypatch.setattr(k, "apply_filter_to_signal", oracle.apply_filter_to_signal)
    monkeypatch.setattr(k, "train_equaliser_windows_search", _oracle_search)
    monkeypatch.setattr(phaserecovery._dsp, "comp_freq_offset", _numpy_cfo)
    monkeypatch.setattr(phaserecovery._dsp, "pilot_phase_trace", _numpy_trace)
    cpu = _config5_chain(cap, dtype)
    assert hip["ok"] and cpu["ok"] and np.array_equal(hip["shifts"], cpu["shifts"])
    if prec == "c128":
        np.testing.assert_allclose(hip["taps"], cpu["taps"], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(hip["out"], cpu["out"], rtol=0, atol=1e-6)
        assert np.array_equal(hip["ser"], cpu["ser"]) and hip["ser"].max() < 5e-2
    else:
        # complex64: 2 x 30 sweeps of the adaptive-step recurrence in single precision on both sides, in different summation orders (tests/conftest.py:
        # rtol / atol 1e-4 for one sweep of a few thousand steps); the 256-QAM decisions of the two chains may part on a few symbols of 2 x 2^16
        tap_dev = float(np.max(np.abs(hip["taps"] - cpu["taps"])))
        out_dev = float(np.sqrt(np.mean(np.abs(hip["out"] - cpu["out"]) ** 2) / np.mean(np.abs(cpu["out"]) ** 2)))
        assert tap_dev <= 1e-3 and out_dev <= 2e-3, (tap_dev, out_dev)
        assert np.max(np.abs(hip["ser"] - cpu["ser"])) <= 3e-4 and hip["ser"].max() < 5e-2, (hip["ser"], cpu["ser"])


@pytest.mark.gpu
def test_long_pilot_sequence_results_are_complete_arrays(monkeypatch):
    """with a pilot sequence long enough for its equalised copy to reach the pinned-result path (>= 1 MiB: complex128, 2 modes,
    2^15 symbols) `equalize_pilot_sequence` read arrays whose device -> host copies were only ENQUEUED (the field deferred them and nobody called
    finish()).  The deferral is opt-in now (only equalise_signal / dual_mode_equalisation, which finish): the result equals the one obtained with the
    pinned path switched off, run after run."""
    from qampy_amd import _lib, theory
    rng = np.random.default_rng(11)
    seq_len, os_, nt = 2 ** 15, 2, 17
    ref = theory.coded_symbols_qam(4, dtype=np.complex128)[rng.integers(0, 4, (2, seq_len))]
    up = np.repeat(ref, os_, axis=1)
    h = np.array([0.08, 0.9, 0.25, -0.05])
    rx = np.stack([np.convolve(up[0] + 0.15 * up[1], h)[:up.shape[1]], np.convolve(up[1] - 0.1j * up[0], h)[:up.shape[1]]])
    rx = rx * np.exp(2j * np.pi * 3e-6 * np.arange(rx.shape[1]))
    rx = np.concatenate([rx, np.zeros((2, nt + 3))], axis=1) + 0.02 * (rng.normal(size=(2, rx.shape[1] + nt + 3)) + 1j * rng.normal(size=(2, rx.shape[1] + nt + 3)))
    kw = dict(os=os_, foe_comp=True, mu=(2e-3, 2e-3), M_pilot=4, Ntaps=nt, Niter=2, adaptive_stepsize=False, methods=("cma", "sbd_data"))
    runs = [pil.equalize_pilot_sequence(rx, ref, np.array([0, 0]), **kw) for _ in range(3)]
    monkeypatch.setattr(_lib, "PINNED_MIN_BYTES", 1 << 40)            # every result through the synchronous pageable path
    taps_ref, foe_ref = pil.equalize_pilot_sequence(rx, ref, np.array([0, 0]), **kw)
    assert np.all(np.isfinite(taps_ref)) and abs(float(foe_ref[0, 0])) > 0
    for taps, foe in runs:
        assert np.array_equal(taps, taps_ref) and np.array_equal(foe, foe_ref)
