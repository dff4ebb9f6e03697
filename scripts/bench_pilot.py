#!/usr/bin/env python3
"""Pilot-based receiver (BASELINE config 5 shape: 256-QAM payload, QPSK pilots, 2 modes): frame sync + pilot-sequence equaliser +
filter over a frame + pilot phase recovery through the basic API (host arrays in / out, PCIe included), HIP kernels vs the CPU
port (the same host layer on the oracle's kernels).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, theory, _lib, equalisation, phaserec
from qampy_amd.signals import PilotSignal
from qampy_amd.core.equalisation import equalisation as core_eq
from oracle import oracle

M, frame_len, seq_len, ins_rat, nframes, os_, fb = 256, 2 ** 16, 2 ** 10, 32, 3, 2, 24e9
rng = np.random.default_rng(5)
idx, idx_dat, idx_pil = PilotSignal._cal_pilot_idx(frame_len, seq_len, ins_rat)
npil, ndat = int(idx_pil.sum()), int(idx_dat.sum())
pil_alpha, dat_alpha = theory.coded_symbols_qam(4, dtype=np.complex128), theory.coded_symbols_qam(M, dtype=np.complex128)
pilots = pil_alpha[rng.integers(0, 4, (2, npil))]
payload = dat_alpha[rng.integers(0, M, (2, ndat))]
frame = np.empty((2, frame_len), np.complex128)
frame[:, idx_pil] = pilots
frame[:, idx_dat] = payload
tx = np.tile(frame, nframes)
tx[1] = np.roll(tx[1], 700)                                              # modal delay: the second mode's frames start 700 symbols later
cap = synth.make_capture(M, tx.shape[1], nmodes=2, os=os_, snr_db=35, theta=np.pi / 5.6, dgd=10e-12, linewidth=10e3, fb=fb, beta=0.1, seed=9,
                         dtype=np.complex128, symbols=tx)
E = np.roll(np.asarray(cap), 2 * 12345, axis=1)                        # the frame starts somewhere inside the capture
n = np.arange(E.shape[1])
E = E * np.exp(2j * np.pi * 40e6 / (fb * os_) * n)                      # 40 MHz frequency offset


def chain(label):
    sig = PilotSignal(E.copy(), M, fb, fb * os_, frame_len, seq_len, ins_rat, pilots, symbols=payload, coded_symbols=dat_alpha)
    t = [time.perf_counter()]
    ok = sig.sync2frame()
    t.append(time.perf_counter())
    sig.corr_foe()
    t.append(time.perf_counter())
    taps, eq = equalisation.pilot_equaliser(sig, (1e-3, 1e-3), 45, foe_comp=False, methods=("cma", "sbd_data"))
    t.append(time.perf_counter())
    out, ph = phaserec.pilot_cpe(eq, N=5, use_seq=False)
    t.append(time.perf_counter())
    ser = out.cal_ser(frames=[0])
    d = np.diff(t)
    return dict(path=label, sync_ok=bool(ok), ms=dict(frame_sync=round(d[0] * 1e3, 2), foe=round(d[1] * 1e3, 2), pilot_equaliser_and_apply=round(d[2] * 1e3, 2),
                                                       pilot_cpe=round(d[3] * 1e3, 2), total=round((t[-1] - t[0]) * 1e3, 2)),
                payload_ser=[float(s) for s in ser], taps=taps)


_lib.init(0)
chain("warm-up")
gpu = chain("hip")
oracle.build(fast_native=True)
k = core_eq._kernels
saved = (k.train_equaliser, k.apply_filter_to_signal, k.train_equaliser_windows)


def _windows(Ein, starts, win_len, TrSyms, Niter, os2, mu, wx0, modes, adaptive, symbols, method):
    res = [oracle.train_equaliser(np.ascontiguousarray(Ein[:, s:s + win_len]), TrSyms, Niter, os2, mu, wx0.copy(), modes, adaptive, symbols, method, fast=True)
           for s in np.asarray(starts)]
    return np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res])


k.train_equaliser = lambda *a: oracle.train_equaliser(*a, fast=True)
k.apply_filter_to_signal = lambda *a, **kw: oracle.apply_filter_to_signal(*a, fast=True, **kw)
k.train_equaliser_windows = _windows
try:
    cpu = chain("cpu port (%d threads)" % os.cpu_count())
finally:
    k.train_equaliser, k.apply_filter_to_signal, k.train_equaliser_windows = saved
tap_diff = float(np.max(np.abs(gpu.pop("taps") - cpu.pop("taps"))))
nsym = frame_len
print(json.dumps(dict(what="pilot receiver, %d-QAM payload, frame %d symbols (pilot sequence %d, 1 phase pilot per %d), %d frames captured, 1 frame "
                           "recovered, 2 modes, 2 SPS; host arrays in/out" % (M, frame_len, seq_len, ins_rat, nframes),
                      hip=gpu, cpu=cpu, max_abs_tap_diff=tap_diff, frame_MSym_per_s=dict(hip=round(nsym / gpu["ms"]["total"] / 1e3, 3),
                                                                                          cpu=round(nsym / cpu["ms"]["total"] / 1e3, 3)))))
