#!/usr/bin/env python3
"""Experiment: does a cleaner (less noisy) set of stage-1 taps as seeds lower the first boundary defect of stage 2?
Emulation: after stage 1 (tier b), one more exact cma sweep with mu/k over the first 2^18 symbol periods refines rx.wxy."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from qampy_amd import synth, _lib
from qampy_amd._lib import DeviceArray
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core.equalisation import hip_equalisation as hk

cfg = dict(bench.WORKLOADS["c3"])
nsym = cfg["nsym"]
_lib.init(0)
for seed in (1000, 1001):
    d = synth.make_capture_dev(cfg["M"], nsym, nmodes=2, snr_db=cfg["snr_db"], theta=np.pi / 5.6, dgd=30e-12, linewidth=cfg["linewidth"], seed=seed)
    kw = dict(methods=cfg["methods"], Niter=cfg["niter"], Mtestangles=cfg["A"], Nbps=cfg["Nbps"], alphabet=d["alphabet_host"])
    for div in (0, 4, 16):
        rx = ResidentReceiver(2, nsym * 2, 2, cfg["M"], cfg["ntaps"], cfg["mu"], tier="b", **kw)
        rx.E.copy_from(d["E"])
        rx.reset(); rx.build_gram(); rx.train(0)
        if div:
            n = 1 << 18
            dmu = DeviceArray.from_host(np.array([cfg["mu"][0] / div], np.float32))
            derr = DeviceArray((2, n), np.complex64)
            hk.train_equaliser_dev(rx.E, n, 1, 2, dmu, rx.wxy, rx.modes, False, rx.symbols[0], cfg["methods"][0], derr)
        rx.train(1); rx.apply(); rx.recover(); _lib.sync()
        rep = rx.pit_reports()
        print(seed, "mu/%d" % div if div else "plain", [(r["passes"], [round(x, 4) for x in r["defect"]]) for r in rep], flush=True)
        del rx
