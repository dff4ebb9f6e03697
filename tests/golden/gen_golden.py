#!/usr/bin/env python3
"""
Generate the golden input/output vectors of the hot path by IMPORTING THE REFERENCE (this container only).

Run from the repo root:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 -O tests/golden/gen_golden.py

``-O`` is mandatory: qampy/core/pythran_dsp.py:69 asserts ``p == 0 or p == L`` although it is always called with a
(1, A) grid; the compiled reference drops asserts (-DNDEBUG), pure Python needs -O to do the same.
pythran is not installed here, so the two "pythran" modules run as the plain Python they are - sequential semantics,
identical to the compiled reference with OMP_NUM_THREADS=1.

Only *data* is written (inputs and the reference's outputs) into tests/golden/*.npz + cases.json.  Inputs are
synthesised with qampy_amd.synth (the build's own generator) so that this script touches the reference only for the
functions under test.
"""
import json
import os
import sys
import time

import numpy as np

assert not __debug__, "run with python3 -O (see docstring)"

from qampy.core.equalisation import pythran_equalisation as ref_eq          # noqa: E402
from qampy.core.equalisation import equalisation as ref_core_eq             # noqa: E402
from qampy.core import pythran_dsp as ref_dsp                               # noqa: E402
from qampy.core import phaserecovery as ref_core_ph                         # noqa: E402
from qampy import equalisation as ref_basic_eq                              # noqa: E402
from qampy import phaserec as ref_basic_ph                                  # noqa: E402
from qampy import theory as ref_theory                                      # noqa: E402
from qampy import signals as ref_signals                                    # noqa: E402

from qampy_amd import synth                                                 # noqa: E402
from qampy_amd.signals import SignalQAM                                     # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CT = {"c64": np.complex64, "c128": np.complex128}
RT = {"c64": np.float32, "c128": np.float64}


def save(name, arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (name, len(arrays), os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------ constants (row H)
def gen_constants():
    arr = {}
    for M in (4, 16, 32, 64, 128, 256):
        arr["M%d_symbols" % M] = ref_theory.cal_symbols_qam(M)
        arr["M%d_scale" % M] = np.float64(ref_theory.cal_scaling_factor_qam(M))
        arr["M%d_graycode" % M] = ref_theory.gray_code_qam(M)
        arr["M%d_R" % M] = np.float64(ref_core_eq._cal_Rconstant(M))
        arr["M%d_Rc" % M] = np.complex128(ref_core_eq._cal_Rconstant_complex(M))
        arr["M%d_rde" % M] = ref_core_eq.generate_partition_codes_radius(M)
        arr["M%d_mrde" % M] = ref_core_eq.generate_partition_codes_complex(M)
        for dn, dt in CT.items():
            arr["M%d_coded_%s" % (M, dn)] = ref_signals.SignalQAMGrayCoded(M, 8, dtype=dt).coded_symbols
            for method in ("cma", "cma2", "sgncma", "mcma", "rde", "mrde", "sbd", "mddma", "dd", "sgncma_real",
                           "cma_real", "dd_real"):
                arr["M%d_eqsyms_%s_%s" % (M, method, dn)] = ref_core_eq.generate_symbols_for_eq(method, M, dt)
    # _reshape_symbols behaviour (equalisation.py:568-594) for representative shapes
    s16 = ref_signals.SignalQAMGrayCoded(16, 8).coded_symbols
    rs = {}
    for nm in (1, 2, 3):
        for method in ("cma", "mcma", "rde", "mrde", "sbd", "dd", "mddma"):
            rs["reshape_%s_n%d_none" % (method, nm)] = ref_core_eq._reshape_symbols(None, method, 16, np.complex128, nm)
            rs["reshape_%s_n%d_coded" % (method, nm)] = ref_core_eq._reshape_symbols(s16, method, 16, np.complex64, nm)
    for nm in (2, 4):
        for method in ("cma_real", "dd_real", "sgncma_real"):
            rs["reshape_%s_n%d_none" % (method, nm)] = ref_core_eq._reshape_symbols(None, method, 16, np.float64, nm)
        rs["reshape_dd_real_n%d_coded" % nm] = ref_core_eq._reshape_symbols(s16, "dd_real", 16, np.float32, nm)
    arr.update(rs)
    arr["coded16_input"] = s16
    save("constants.npz", arr)


# ------------------------------------------------------------------------------------------------ inputs
def gen_inputs():
    """Impaired 2 SPS captures (stored as complex128; the complex64 cases use .astype(complex64))."""
    inp = {}
    meta = {}

    def add(name, **kw):
        sig = synth.make_capture(dtype=np.complex128, **kw)
        inp[name + "_E"] = np.asarray(sig)
        inp[name + "_tx"] = np.asarray(sig.symbols)
        meta[name] = dict(kw, fb=kw.get("fb", 20e9))

    add("q4_1m", M=4, nsym=1400, nmodes=1, snr_db=16, seed=11)
    add("q16_2m", M=16, nsym=1400, nmodes=2, snr_db=24, theta=np.pi / 5.6, dgd=30e-12, seed=12)
    add("q64_2m", M=64, nsym=1400, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, seed=13)
    add("q16_3m", M=16, nsym=900, nmodes=3, snr_db=24, seed=14)
    # data-aided: symbol i must sit under the centre tap of window i -> delay the waveform by ntaps//2 samples
    add("q16_2m_da", M=16, nsym=1400, nmodes=2, snr_db=24, theta=np.pi / 7, dgd=20e-12, seed=15, shift=5)
    save("inputs.npz", inp)
    return inp, meta


def preconverge(E, M, ntaps, os_=2):
    """Taps after two MCMA sweeps: a sane starting point for the decision-directed cases."""
    wxy, _ = ref_core_eq.equalise_signal(E, os_, 2e-3, M, Ntaps=ntaps, Niter=3, method="mcma")
    return wxy


# ------------------------------------------------------------------------------------------------ train (rows A, B, C)
def gen_train(inp):
    arr = {}
    cases = []
    tap_cache = {}

    def run(name, inname, M, method, dn, ntaps, adaptive, Niter=1, modes=None, mu=1e-3, TrSyms=None, conv=False,
            symbols=None, os_=2):
        ct, rt = CT[dn], RT[dn]
        E = np.ascontiguousarray(inp[inname + "_E"].astype(ct))
        nmodes = E.shape[0]
        if conv:
            key = (inname, ntaps)
            if key not in tap_cache:
                tap_cache[key] = preconverge(inp[inname + "_E"], M, ntaps)
            wx0 = tap_cache[key].astype(ct)
        else:
            wx0 = ref_core_eq._init_taps(ntaps, nmodes, nmodes, ct)
        if TrSyms is None:
            TrSyms = ref_core_eq._cal_training_symbol_len(os_, ntaps, E.shape[1])
        syms = ref_core_eq._reshape_symbols(symbols, method, M, ct, nmodes)
        md = np.arange(nmodes) if modes is None else np.atleast_1d(modes)
        wx = wx0.copy()
        err, wx, mu_out = ref_eq.train_equaliser(E, TrSyms, Niter, os_, rt(mu), wx, md, adaptive, syms.copy(), method)
        assert err.dtype == ct and wx.dtype == ct, (err.dtype, wx.dtype)
        arr[name + "__wx0"] = wx0
        arr[name + "__symbols"] = syms
        arr[name + "__err"] = err
        arr[name + "__wx"] = wx
        arr[name + "__mu"] = np.asarray(mu_out)
        cases.append(dict(name=name, input=inname, M=M, method=method, dtype=dn, ntaps=ntaps, adaptive=bool(adaptive),
                          Niter=Niter, modes=[int(m) for m in md], mu=mu, TrSyms=int(TrSyms), os=os_))

    t0 = time.time()
    for dn in ("c128", "c64"):
        for adaptive in (False, True):
            a = "a1" if adaptive else "a0"
            for method in ("cma", "sgncma", "mcma", "rde", "mrde"):
                run("tr_%s_%s_%s" % (method, dn, a), "q16_2m", 16, method, dn, 11, adaptive)
            # cma2 (complex X**2) is not phase blind: it only stays bounded from converged taps with a small step
            run("tr_cma2_%s_%s" % (dn, a), "q16_2m", 16, "cma2", dn, 11, adaptive, conv=True, mu=1e-4, TrSyms=400)
            for method in ("sbd", "mddma", "dd"):
                run("tr_%s_%s_%s" % (method, dn, a), "q16_2m", 16, method, dn, 11, adaptive, conv=True, mu=5e-4)
            tx = inp["q16_2m_da_tx"]
            run("tr_sbd_data_%s_%s" % (dn, a), "q16_2m_da", 16, "sbd_data", dn, 11, adaptive, symbols=tx, mu=2e-3)
        # shape / sweep variants (non-adaptive and adaptive mixed)
        run("tr_cma_q4_1m_%s" % dn, "q4_1m", 4, "cma", dn, 11, False)
        run("tr_mcma_q4_1m_ad_%s" % dn, "q4_1m", 4, "mcma", dn, 7, True, Niter=2)
        run("tr_mcma_t21_it3_%s" % dn, "q16_2m", 16, "mcma", dn, 21, False, Niter=3)
        run("tr_cma_t41_%s" % dn, "q64_2m", 64, "cma", dn, 41, False, Niter=2)
        run("tr_mrde_t41_%s" % dn, "q64_2m", 64, "mrde", dn, 41, False, conv=True, mu=5e-4)
        run("tr_rde_q64_%s" % dn, "q64_2m", 64, "rde", dn, 21, True, conv=True, mu=5e-4)
        run("tr_sbd_q64_%s" % dn, "q64_2m", 64, "sbd", dn, 21, True, conv=True, mu=5e-4)
        run("tr_mcma_mode1_%s" % dn, "q16_2m", 16, "mcma", dn, 11, False, modes=[1])
        run("tr_mcma_mode10_ad_%s" % dn, "q16_2m", 16, "mcma", dn, 11, True, modes=[1, 0])
        run("tr_mcma_3m_%s" % dn, "q16_3m", 16, "mcma", dn, 9, False, modes=[0, 2])
        run("tr_cma_trsyms_%s" % dn, "q16_2m", 16, "cma", dn, 13, True, TrSyms=500, Niter=2)
    print("train complex cases: %.1f s" % (time.time() - t0))

    # real-valued trainer (row C)
    def run_real(name, inname, M, method, dn, ntaps, adaptive, Niter=1, modes=None, mu=1e-3, symbols=None, conv=False):
        ct, rt = CT[dn], RT[dn]
        Ec = inp[inname + "_E"].astype(ct)
        E = ref_core_eq._convert_sig_to_real(Ec)
        nmodes = E.shape[0]
        if conv:
            wc = preconverge(inp[inname + "_E"], M, ntaps)
            # real-valued image of complex taps: out_r = wr*xr - wi*xi ; out_i = wi*xr + wr*xi
            n = nmodes // 2
            wx0 = np.zeros((nmodes, nmodes, ntaps), dtype=rt)
            wx0[:n, :n] = wc.real
            wx0[:n, n:] = -wc.imag
            wx0[n:, :n] = wc.imag
            wx0[n:, n:] = wc.real
        else:
            wx0 = ref_core_eq._init_taps(ntaps, nmodes, nmodes, rt)
        TrSyms = ref_core_eq._cal_training_symbol_len(2, ntaps, E.shape[1])
        syms = ref_core_eq._reshape_symbols(symbols, method, M, rt, nmodes)
        md = np.arange(nmodes) if modes is None else np.hstack([np.atleast_1d(modes), np.atleast_1d(modes) + nmodes // 2])
        wx = wx0.copy()
        err, wx, mu_out = ref_eq.train_equaliser_realvalued(E, TrSyms, Niter, 2, rt(mu), wx, md, adaptive, syms.copy(),
                                                            method[:-5])
        arr[name + "__wx0"] = wx0
        arr[name + "__symbols"] = syms
        arr[name + "__err"] = err
        arr[name + "__wx"] = wx
        arr[name + "__mu"] = np.asarray(mu_out)
        cases.append(dict(name=name, input=inname, M=M, method=method, dtype=dn, ntaps=ntaps, adaptive=bool(adaptive),
                          Niter=Niter, modes=[int(m) for m in md], mu=mu, TrSyms=int(TrSyms), os=2, real=True))

    for dn in ("c128", "c64"):
        run_real("trr_cma_%s" % dn, "q16_2m", 16, "cma_real", dn, 11, False)
        run_real("trr_cma_ad_%s" % dn, "q16_2m", 16, "cma_real", dn, 11, True, Niter=2)
        run_real("trr_sgncma_%s" % dn, "q16_2m", 16, "sgncma_real", dn, 11, False, mu=1e-4)
        run_real("trr_dd_%s" % dn, "q16_2m", 16, "dd_real", dn, 11, True, conv=True, mu=5e-4)
        run_real("trr_dd_data_%s" % dn, "q16_2m_da", 16, "dd_data_real", dn, 11, False, symbols=inp["q16_2m_da_tx"],
                 mu=2e-3)
        run_real("trr_cma_mode1_%s" % dn, "q16_2m", 16, "cma_real", dn, 9, False, modes=[1])
    save("train.npz", arr)
    return cases


# ------------------------------------------------------------------------------------------------ apply (row D)
def gen_apply(inp):
    arr = {}
    cases = []
    rng = np.random.default_rng(5)
    for dn in ("c128", "c64"):
        ct, rt = CT[dn], RT[dn]
        for inname, ntaps, os_, modes in (("q16_2m", 11, 2, None), ("q64_2m", 41, 2, None), ("q16_2m", 21, 2, [1]),
                                          ("q16_3m", 8, 2, [2, 0]), ("q4_1m", 5, 1, None), ("q16_2m", 13, 3, [1, 0])):
            E = np.ascontiguousarray(inp[inname + "_E"].astype(ct))
            nm = E.shape[0]
            wx = (rng.standard_normal((nm, nm, ntaps)) + 1j * rng.standard_normal((nm, nm, ntaps))).astype(ct) / ntaps
            md = None if modes is None else np.asarray(modes)
            out = ref_eq.apply_filter_to_signal(E, os_, wx, md)
            name = "ap_%s_t%d_os%d_%s_%s" % (inname, ntaps, os_, "all" if modes is None else "".join(map(str, modes)), dn)
            arr[name + "__wx"] = wx
            arr[name + "__out"] = out
            cases.append(dict(name=name, input=inname, ntaps=ntaps, os=os_, modes=modes, dtype=dn))
        # real-valued overload through the host layer's packing (equalisation.py:178-184)
        E = np.ascontiguousarray(inp["q16_2m_E"].astype(ct))
        wr = (rng.standard_normal((4, 4, 9)) / 9).astype(rt)
        out = ref_core_eq.apply_filter(E, 2, wr)
        name = "ap_realtaps_%s" % dn
        arr[name + "__wx"] = wr
        arr[name + "__out"] = out
        cases.append(dict(name=name, input="q16_2m", ntaps=9, os=2, modes=None, dtype=dn, realtaps=True))
    save("apply.npz", arr)
    return cases


# ------------------------------------------------------------------------------------------------ bps (rows E, F, G)
def gen_bps():
    arr = {}
    cases = []
    rng = np.random.default_rng(77)
    t0 = time.time()
    for (M, A, N, L) in ((4, 16, 10, 1500), (16, 32, 20, 1500), (32, 32, 10, 1200), (64, 64, 20, 1000), (64, 16, 10, 800),
                         (16, 64, 10, 1000), (16, 12, 7, 700)):
        alphabet = ref_signals.SignalQAMGrayCoded(M, 8).coded_symbols
        nm = 2
        tx = alphabet[rng.integers(0, M, size=(nm, L))]
        ph = np.cumsum(rng.normal(scale=np.sqrt(2 * np.pi * 100e3 / 20e9), size=(nm, L)), axis=1) + 0.3
        snr = {4: 14, 16: 22, 32: 25, 64: 28}[M]
        noise = (rng.standard_normal((nm, L)) + 1j * rng.standard_normal((nm, L))) * 10 ** (-snr / 20) / np.sqrt(2)
        E128 = (tx + noise) * np.exp(1j * ph)
        base = "bps_M%d_A%d_N%d" % (M, A, N)
        arr[base + "__E"] = E128
        for dn in ("c128", "c64"):
            ct, rt = CT[dn], RT[dn]
            E = E128.astype(ct)
            sig = SignalQAM(E, M, coded_symbols=alphabet.astype(ct))
            angles = np.linspace(-np.pi / 4, np.pi / 4, A, endpoint=False, dtype=rt).reshape(1, -1)
            idx = np.array([ref_dsp.bps(E[i], angles, alphabet.astype(ct), N) for i in range(nm)])
            Eout, phout = ref_basic_ph.bps(sig, A, N)
            assert type(Eout) is SignalQAM and phout.dtype == rt, (type(Eout), phout.dtype)
            arr["%s_%s__idx" % (base, dn)] = idx
            arr["%s_%s__Eout" % (base, dn)] = np.asarray(Eout)
            arr["%s_%s__ph" % (base, dn)] = np.asarray(phout)
            # 1-D input flavour of the core API (phaserecovery.py:156-159)
            if M == 16 and A == 32:
                e1, p1 = ref_core_ph.bps(E[0], A, alphabet.astype(ct), N)
                arr["%s_%s__Eout1d" % (base, dn)] = np.asarray(e1)
                arr["%s_%s__ph1d" % (base, dn)] = np.asarray(p1)
            cases.append(dict(name="%s_%s" % (base, dn), base=base, M=M, A=A, N=N, L=L, dtype=dn))
        arr[base + "__alphabet"] = alphabet
    # per-symbol angle grid (p == L branch, pythran_dsp.py:76-79) + select_angles with a 2-D grid
    M, A, N, L = 16, 8, 10, 600
    alphabet = ref_signals.SignalQAMGrayCoded(M, 8).coded_symbols
    E = (alphabet[rng.integers(0, M, size=L)] + 0.05 * (rng.standard_normal(L) + 1j * rng.standard_normal(L))) * np.exp(0.2j)
    grid = (np.linspace(-0.3, 0.3, A)[None, :] + 0.05 * rng.standard_normal((L, 1)))
    idx = ref_dsp.bps(E, grid, alphabet, N)
    arr["bps_grid__E"] = E
    arr["bps_grid__angles"] = grid
    arr["bps_grid__alphabet"] = alphabet
    arr["bps_grid__idx"] = idx
    arr["bps_grid__sel"] = ref_dsp.select_angles(grid, idx)
    arr["bps_grid__sel1"] = ref_dsp.select_angles(grid[:1].copy(), idx.astype(int))
    cases.append(dict(name="bps_grid", M=M, A=A, N=N, L=L, dtype="c128"))
    print("bps cases: %.1f s" % (time.time() - t0))
    save("bps.npz", arr)
    return cases


# ------------------------------------------------------------------------------------------------ bps_twostage (row f.1)
def gen_twostage():
    """Two-stage BPS (qampy/core/phaserecovery.py:222-288) through the basic wrapper qampy/phaserec.py:24-60."""
    arr = {}
    cases = []
    rng = np.random.default_rng(4242)
    t0 = time.time()
    for (M, A, N, B, L) in ((16, 16, 10, 4, 900), (64, 32, 15, 4, 700), (4, 8, 8, 6, 600)):
        alphabet = ref_signals.SignalQAMGrayCoded(M, 8).coded_symbols
        nm = 2
        tx = alphabet[rng.integers(0, M, size=(nm, L))]
        ph = np.cumsum(rng.normal(scale=np.sqrt(2 * np.pi * 50e3 / 20e9), size=(nm, L)), axis=1) - 0.25
        snr = {4: 14, 16: 22, 64: 28}[M]
        noise = (rng.standard_normal((nm, L)) + 1j * rng.standard_normal((nm, L))) * 10 ** (-snr / 20) / np.sqrt(2)
        E128 = (tx + noise) * np.exp(1j * ph)
        base = "ts_M%d_A%d_N%d_B%d" % (M, A, N, B)
        arr[base + "__E"] = E128
        arr[base + "__alphabet"] = alphabet
        for dn in ("c128", "c64"):
            ct = CT[dn]
            sig = SignalQAM(E128.astype(ct), M, coded_symbols=alphabet.astype(ct))
            Eout, phout = ref_basic_ph.bps_twostage(sig, A, N, B=B)
            assert type(Eout) is SignalQAM, type(Eout)
            arr["%s_%s__Eout" % (base, dn)] = np.asarray(Eout)
            arr["%s_%s__ph" % (base, dn)] = np.asarray(phout)
            e1, p1 = ref_core_ph.bps_twostage(E128[0].astype(ct), A, alphabet.astype(ct), N, B=B)
            arr["%s_%s__Eout1d" % (base, dn)] = np.asarray(e1)
            arr["%s_%s__ph1d" % (base, dn)] = np.asarray(p1)
            cases.append(dict(name="%s_%s" % (base, dn), base=base, M=M, A=A, N=N, B=B, L=L, dtype=dn))
    print("twostage cases: %.1f s" % (time.time() - t0))
    save("twostage.npz", arr)
    return cases


# ------------------------------------------------------------------------------------------------ pilot receiver (row f.3)
def gen_pilot():
    """
    Core pilot-based receiver (qampy/core/pilotbased_receiver.py) on plain arrays: frame_sync, coarse FOE compensation,
    equalize_pilot_sequence, filter application to one frame, pilot_based_cpe_new.  The reference's signal classes and
    impairment chain are used here only to PRODUCE the test capture; every array the functions see is stored.
    """
    from qampy.core import pilotbased_receiver as ref_pil
    from qampy.core import ber_functions as ref_ber
    from qampy.core import filter as ref_filter
    from qampy import impairments as ref_imp
    arr = {}
    np.random.seed(20240928)
    t0 = time.time()
    frame_len, seq_len, ins_rat, M = 2 ** 12, 2 ** 8, 32, 64
    sig = ref_signals.SignalWithPilots(M, frame_len, seq_len, ins_rat, nmodes=2, Mpilots=4, nframes=3, fb=24e9)
    sig2 = sig.resample(sig.fb * 2, beta=0.1)
    rx = ref_imp.simulate_transmission(sig2, snr=27, dgd=10e-12, freq_off=40e6, lwdth=50e3, roll_frame_sync=True,
                                       modal_delay=(700, 500))
    os_ = int(rx.os)
    pilot_seq = np.asarray(sig.pilot_seq)
    ph_pilots = np.asarray(sig.ph_pilots)
    idx_pil = np.asarray(sig._idx_pil)
    E = np.array(rx, dtype=np.complex128, copy=True)
    arr.update(rx=E, pilot_seq=pilot_seq, ph_pilots=ph_pilots, idx_pil=idx_pil, frame_len=np.int64(frame_len), os=np.int64(os_),
               tx_frame=np.asarray(sig)[:, :frame_len], alphabet=np.asarray(sig.symbols.coded_symbols), M=np.int64(M))
    # --- helpers
    arr["fso_x"] = pilot_seq[0]
    arr["fso_y"] = np.roll(pilot_seq[0], 17)[:200] * 1j
    ix, y2, ii, acm = ref_ber.find_sequence_offset_complex(arr["fso_x"], arr["fso_y"])
    arr.update(fso_ix=np.int64(ix), fso_ii=np.int64(ii), fso_acm=np.float64(acm))
    arr["mavg_in"] = np.random.randn(2, 50)
    arr["mavg_out"] = ref_filter.moving_average(arr["mavg_in"], 5)
    # --- frame sync (sync2frame defaults, qampy/signals.py:1725-1733)
    shift, foe, order, wx1, ok = ref_pil.frame_sync(E, pilot_seq, os_, frame_len=frame_len, M_pilot=4, mu=5e-3, Ntaps=17,
                                                    adaptive_stepsize=True, Niter=10, method="cma")
    arr.update(fs_shift=shift.copy(), fs_foe=np.asarray(foe), fs_order=order.copy(), fs_wx1=wx1, fs_ok=np.bool_(ok))
    # --- what sync2frame / corr_foe do with it
    E2 = E[order, :]
    shift2 = shift.copy()
    shift2[shift2 < 0] += frame_len * os_
    shiftfctrs = shift2[order]
    foe_off = np.ones(np.asarray(foe).shape) * np.mean(foe)
    E3 = ref_core_ph.comp_freq_offset(E2, foe_off)
    arr.update(synced=E3, shiftfctrs=shiftfctrs)
    arr["ffo_in"] = E3[:, :4096:2]
    arr["ffo_out"] = ref_core_ph.find_freq_offset(arr["ffo_in"], fft_size=2 ** 12)
    # --- pilot equaliser (qampy/equalisation.py:268-338 with Ntaps=45, synctaps=17, frame 0)
    Ntaps = 45
    eq_shift = shiftfctrs - (Ntaps - 17) // 2
    taps, foe_all = ref_pil.equalize_pilot_sequence(E3, pilot_seq, eq_shift, os_, mu=(1e-3, 1e-3), foe_comp=False, Ntaps=Ntaps,
                                                    methods=("cma", "sbd"))
    arr.update(eq_shift=eq_shift, eq_taps=taps, eq_foe=foe_all)
    frames = []
    for m in range(2):                       # _apply_to_pilotsignal with per-mode shift factors (qampy/equalisation.py:42-87)
        i0 = eq_shift[m]
        frames.append(ref_core_eq.apply_filter(E3[:, i0:i0 + frame_len * os_ + Ntaps - 1], os_, taps, modes=[m])[0])
    eq = np.array(frames)
    arr["eq_frame"] = eq
    # --- pilot CPE (qampy/phaserec.py:156-192 with use_seq=False, N=5)
    idx = np.nonzero(idx_pil)[0][seq_len:]
    out, ph = ref_pil.pilot_based_cpe_new(eq, ph_pilots, idx, frame_len, seq_len=None, max_num_blocks=None, use_pilot_ratio=1,
                                          num_average=5, nframes=1)
    arr.update(cpe_idx=idx, cpe_out=out, cpe_ph=ph)
    foe_p, foe_pm, cond = ref_pil.pilot_based_foe(eq[:, :seq_len], pilot_seq)
    arr.update(pfoe=np.float64(foe_p), pfoe_modes=foe_pm, pfoe_cond=cond)
    print("pilot cases: %.1f s, sync ok=%s shift=%s order=%s" % (time.time() - t0, ok, shift, order))
    save("pilot.npz", arr)


def gen_pilot_frames():
    """
    Frame-by-frame pilot equalisation and the constant-phase helpers through the reference's BASIC API on its own signal
    object (qampy/equalisation.py:340-397 pilot_equaliser_nframes -> :268-338 pilot_equaliser -> :42-87 _apply_to_pilotsignal;
    qampy/phaserec.py:194-238 find_pilot_const_phase / correct_pilot_const_phase), on the SAME seeded capture as gen_pilot
    (checked against pilot.npz), so that the tests can feed the stored arrays to the build's PilotSignal.
    """
    from qampy import impairments as ref_imp
    np.random.seed(20240928)
    frame_len, seq_len, ins_rat, M = 2 ** 12, 2 ** 8, 32, 64
    sig = ref_signals.SignalWithPilots(M, frame_len, seq_len, ins_rat, nmodes=2, Mpilots=4, nframes=3, fb=24e9)
    sig2 = sig.resample(sig.fb * 2, beta=0.1)
    rx = ref_imp.simulate_transmission(sig2, snr=27, dgd=10e-12, freq_off=40e6, lwdth=50e3, roll_frame_sync=True, modal_delay=(700, 500))
    stored = np.load(os.path.join(OUT, "pilot.npz"))
    assert np.array_equal(np.asarray(rx), stored["rx"]), "the seeded capture differs from the one in pilot.npz"
    rx.sync2frame(Ntaps=17)
    rx.corr_foe()
    arr = dict(synced=np.array(rx, copy=True), shiftfctrs=np.asarray(rx.shiftfctrs).copy())
    taps, sout, rest = ref_basic_eq.pilot_equaliser_nframes(rx, (1e-3, 1e-3), 45, foe_comp=False, frames=[0, 1], methods=("cma", "sbd"))
    arr.update(nf_taps=np.array(taps), nf_out=np.asarray(sout).copy(), nf_foe=np.array(rest[0]), nf_ntaps=np.array(rest[1]))
    # (apply=False cannot be captured: the reference's pilot_equaliser forgets the `return` of its verbose no-apply branch, :335-336,
    # and its frame loop indexes a bare taps array otherwise)
    # constant pilot phase on the pilot sequence of the first equalised frame
    rec = np.asarray(sout)[:, :seq_len]
    ref = np.asarray(sig.pilot_seq)
    ph = ref_basic_ph.find_pilot_const_phase(rec, ref)
    arr.update(cp_rec=rec.copy(), cp_ref=ref.copy(), cp_phase=ph, cp_out=np.asarray(ref_basic_ph.correct_pilot_const_phase(rec, ph)))
    save("pilot_frames.npz", arr)


# ------------------------------------------------------------------------------------------------ make_decision (row I)
def gen_decision():
    arr = {}
    rng = np.random.default_rng(9)
    for M in (4, 16, 32, 64, 128):
        alphabet = ref_signals.SignalQAMGrayCoded(M, 8).coded_symbols
        L = 3000
        E = alphabet[rng.integers(0, M, size=L)] + 0.15 * (rng.standard_normal(L) + 1j * rng.standard_normal(L))
        E[:M] = alphabet                      # exact hits
        E[M] = 0                              # equidistant point: pins the first-minimum tie rule
        for dn in ("c128", "c64"):
            ct = CT[dn]
            det, dist, idx = ref_eq.make_decision(E.astype(ct), alphabet.astype(ct))
            arr["md_M%d_%s__det" % (M, dn)] = det
            arr["md_M%d_%s__dist" % (M, dn)] = dist
            arr["md_M%d_%s__idx" % (M, dn)] = idx
        arr["md_M%d__E" % M] = E
        arr["md_M%d__alphabet" % M] = alphabet
    save("decision.npz", arr)


# ------------------------------------------------------------------------------------------------ host layer end to end
def gen_e2e(inp, meta):
    """Core and basic API calls (rows G, J): equalise_signal / dual_mode_equalisation / apply_filter on a signal object."""
    arr = {}
    cases = []

    def sigobj(inname, dn):
        m = meta[inname]
        E = inp[inname + "_E"].astype(CT[dn])
        return SignalQAM(E, m["M"], fb=m["fb"], fs=2 * m["fb"], symbols=inp[inname + "_tx"].astype(CT[dn]))

    t0 = time.time()
    for dn in ("c128", "c64"):
        # basic API, single stage, apply=True, default TrSyms
        s = sigobj("q16_2m", dn)
        out, wxy, err = ref_basic_eq.equalise_signal(s, 1e-3, Ntaps=11, method="mcma", adaptive_stepsize=True, apply=True)
        assert type(out) is SignalQAM and out.fs == s.fb
        arr["e2e_eq_mcma_%s__out" % dn] = np.asarray(out)
        arr["e2e_eq_mcma_%s__wxy" % dn] = wxy
        arr["e2e_eq_mcma_%s__err" % dn] = err
        # basic API, apply=False then apply_filter
        wxy2, err2 = ref_basic_eq.equalise_signal(s, 1e-3, Ntaps=9, Niter=2, method="cma", modes=[1])
        out2 = ref_basic_eq.apply_filter(s, wxy2)
        arr["e2e_eq_cma_m1_%s__wxy" % dn] = wxy2
        arr["e2e_eq_cma_m1_%s__err" % dn] = err2
        arr["e2e_eq_cma_m1_%s__out" % dn] = np.asarray(out2)
        # dual mode, decision-directed second stage, Gray-ordered alphabet from the signal object
        out3, wxy3, (e31, e32) = ref_basic_eq.dual_mode_equalisation(s, (2e-3, 5e-4), 11, Niter=(3, 1), methods=("mcma", "sbd"),
                                                                      adaptive_stepsize=(True, True))
        arr["e2e_dual_mcma_sbd_%s__out" % dn] = np.asarray(out3)
        arr["e2e_dual_mcma_sbd_%s__wxy" % dn] = wxy3
        arr["e2e_dual_mcma_sbd_%s__err1" % dn] = e31
        arr["e2e_dual_mcma_sbd_%s__err2" % dn] = e32
        # C3-shaped: CMA -> MRDE, 41 taps, 64-QAM, non-adaptive, apply=False
        s64 = sigobj("q64_2m", dn)
        wxy4, (e41, e42) = ref_basic_eq.dual_mode_equalisation(s64, (1e-3, 5e-4), 41, Niter=(2, 2), methods=("cma", "mrde"),
                                                                apply=False)
        arr["e2e_dual_cma_mrde_%s__wxy" % dn] = wxy4
        arr["e2e_dual_cma_mrde_%s__err1" % dn] = e41
        arr["e2e_dual_cma_mrde_%s__err2" % dn] = e42
        # real-valued method through the host layer
        out5, wxy5, err5 = ref_basic_eq.equalise_signal(s, 1e-3, Ntaps=11, method="cma_real", apply=True)
        arr["e2e_eq_cma_real_%s__out" % dn] = np.asarray(out5)
        arr["e2e_eq_cma_real_%s__wxy" % dn] = wxy5
        arr["e2e_eq_cma_real_%s__err" % dn] = err5
        # core API with plain arrays, generated alphabet (symbols=None -> raster order)
        E = inp["q16_2m_E"].astype(CT[dn])
        wxy6, err6 = ref_core_eq.equalise_signal(E, 2, 5e-4, 16, wxy=wxy.copy(), method="dd", adaptive_stepsize=False)
        arr["e2e_core_dd_%s__wxy" % dn] = wxy6
        arr["e2e_core_dd_%s__err" % dn] = err6
        # data-aided through the basic API (symbols taken from sig.symbols)
        sda = sigobj("q16_2m_da", dn)
        wxy7, err7 = ref_basic_eq.equalise_signal(sda, 2e-3, Ntaps=11, method="sbd_data", TrSyms=1000)
        arr["e2e_eq_sbd_data_%s__wxy" % dn] = wxy7
        arr["e2e_eq_sbd_data_%s__err" % dn] = err7
        cases.append(dn)
    print("e2e cases: %.1f s" % (time.time() - t0))
    save("e2e.npz", arr)
    return cases


# ------------------------------------------------------------------------------------------------ harness (rows f2, f4)
def gen_harness():
    """
    Vectors that pin the measurement harness to the reference:
      * SER: SignalQAMGrayCoded.cal_ser (qampy/signals.py:295-335 -> _sync_and_adjust :246-267 -> core/ber_functions.py:108-160
        sync_and_adjust / find_sequence_offset_complex, + make_decision) on received versions of known symbols (noise, quarter
        turns, delays, swapped modes);
      * synthesis: pulse shaping to 2 samples/symbol (core/resample.py:73-126 rrcos_resample through Signal.resample),
        first-order PMD (core/impairments.py:94-131 apply_PMD_to_field), the AWGN scaling of change_snr (:188-233) and the
        Wiener phase-noise variance of phase_noise (:133-160), all on given symbols / seeded generators.
    """
    from qampy.core import impairments as ref_imp
    arr = {}
    # ---- SER
    M, N = 64, 4096
    s = ref_signals.SignalQAMGrayCoded(M, N, nmodes=2, fb=20e9, seed=[11, 12], dtype=np.complex128)
    tx = np.asarray(s).copy()
    arr["ser_tx"] = tx
    arr["ser_alphabet"] = np.asarray(s.coded_symbols)
    rng = np.random.default_rng(5)
    cases = []
    for name, rots, lags, swap, snr in (("clean", (0, 0), (0, 0), False, 40.), ("noisy", (0, 0), (0, 0), False, 19.),
                                        ("rot_lag", (1, 3), (37, -12), False, 19.), ("swapped", (2, 1), (5, 200), True, 21.)):
        rx = np.array([np.roll(tx[m] * 1j ** rots[m], lags[m]) for m in range(2)])
        rx = rx + 10 ** (-snr / 20) * (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape)) / np.sqrt(2)
        if swap:
            rx = rx[::-1].copy()
        ser, errs, tx_sync = s.cal_ser(signal_rx=s.recreate_from_np_array(rx), verbose=True)
        arr["ser_%s_rx" % name] = rx
        arr["ser_%s_ser" % name] = np.asarray(ser)
        arr["ser_%s_errmask" % name] = np.asarray(errs) != 0            # per received symbol, in rx order
        arr["ser_%s_txsync" % name] = np.asarray(tx_sync)
        cases.append(dict(name=name, rots=list(rots), lags=list(lags), swap=swap, snr=snr))
    # ---- synthesis
    sy = ref_signals.SignalQAMGrayCoded(16, 4096, nmodes=2, fb=20e9, seed=[3, 5], dtype=np.complex128)
    arr["syn_symbols"] = np.asarray(sy).copy()
    up = sy.resample(2 * sy.fb, beta=0.1, renormalise=True)
    arr["syn_shaped"] = np.asarray(up).copy()
    arr["syn_pmd"] = np.asarray(ref_imp.apply_PMD_to_field(np.asarray(up), np.pi / 5.6, 30e-12, up.fs))
    np.random.seed(7)
    big = ref_signals.SignalQAMGrayCoded(16, 2 ** 15, nmodes=2, fb=20e9, seed=[1, 2], dtype=np.complex128).resample(40e9, beta=0.1, renormalise=True)
    noisy = ref_imp.change_snr(big, 20., big.fb, big.fs)
    arr["syn_noise_std"] = np.float64(np.std(np.asarray(noisy) - np.asarray(big)))
    arr["syn_noise_power_in"] = np.float64(np.mean(np.abs(np.asarray(big)) ** 2))
    np.random.seed(8)
    ph = ref_imp.phase_noise((2, 2 ** 16), 100e3, 40e9)
    arr["syn_pn_step_var"] = np.float64(np.var(np.diff(ph, axis=1)))
    arr["syn_pn_params"] = np.array([100e3, 40e9])
    save("harness.npz", arr)
    return cases


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "pilot_frames":
        gen_pilot_frames()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "harness":          # only this group (the others are unchanged)
        path = os.path.join(OUT, "cases.json")
        cases = json.load(open(path))
        cases["harness"] = gen_harness()
        with open(path, "w") as f:
            json.dump(cases, f, indent=1, default=float)
        return
    gen_constants()
    inp, meta = gen_inputs()
    cases = {"inputs": meta}
    cases["train"] = gen_train(inp)
    cases["apply"] = gen_apply(inp)
    cases["bps"] = gen_bps()
    cases["twostage"] = gen_twostage()
    gen_pilot()
    gen_pilot_frames()
    gen_decision()
    cases["e2e"] = gen_e2e(inp, meta)
    cases["harness"] = gen_harness()
    cases["versions"] = dict(numpy=np.__version__, python=sys.version.split()[0], reference="QAMpy v0.5.1 (/root/reference)")
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1, default=float)
    print("done")


if __name__ == "__main__":
    main()
