#!/usr/bin/env python3
"""Frame-synchronisation search (BASELINE config 5 shape): all CMA search windows in one launch vs the CPU port's loop."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.core.equalisation import equalisation as eq
from oracle import oracle

_lib.init(0)
frame_len, seq_len, os_, ntaps, niter = 2 ** 16, 2 ** 10, 2, 17, 10
sig = synth.make_capture(4, frame_len + 4 * seq_len, nmodes=2, snr_db=20, theta=0.6, dgd=10e-12, seed=3, dtype=np.complex64)
E = np.ascontiguousarray(np.asarray(sig))
window = seq_len * os_
step = window // 2
starts = np.arange(2, (frame_len * os_) // step + 1) * step
eq.equalise_signal_windows(E, os_, 5e-3, 4, starts, window, Ntaps=ntaps, Niter=niter, method="cma", adaptive_stepsize=True)
t0 = time.perf_counter()
w, e = eq.equalise_signal_windows(E, os_, 5e-3, 4, starts, window, Ntaps=ntaps, Niter=niter, method="cma", adaptive_stepsize=True)
t_gpu = time.perf_counter() - t0
oracle.build(fast_native=True)
tr = eq._cal_training_symbol_len(os_, ntaps, window)
sy = eq._reshape_symbols(None, "cma", 4, np.complex64, 2)
t0 = time.perf_counter()
for s in starts:
    oracle.train_equaliser(np.ascontiguousarray(E[:, s:s + window]), tr, niter, os_, np.float32(5e-3), eq._init_taps(ntaps, 2, 2, np.complex64),
                           None, True, sy, "cma", fast=True)
t_cpu = time.perf_counter() - t0
print(json.dumps(dict(what="frame_sync equaliser search, %d windows x %d steps x %d sweeps x 2 modes, CMA %d taps, adaptive" % (
    starts.size, tr, niter, ntaps), gpu_ms_incl_pcie=round(t_gpu * 1e3, 2), cpu_port_ms=round(t_cpu * 1e3, 2), speedup=round(t_cpu / t_gpu, 1))))
