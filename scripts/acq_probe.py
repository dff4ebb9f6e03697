import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
_lib.init(0)
CASES = [dict(name="c3", M=64, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), snr=30, lw=100.),
         dict(name="256qam", M=256, ntaps=41, methods=("cma", "mrde"), mu=(1e-4, 1e-4), snr=36, lw=100.),
         dict(name="c2", M=16, ntaps=21, methods=("mcma",), mu=(1e-3,), snr=25, lw=50e3),
         dict(name="qpsk", M=4, ntaps=11, methods=("cma",), mu=(1e-3,), snr=14, lw=100e3),
         dict(name="64mcma", M=64, ntaps=41, methods=("mcma", "sbd"), mu=(3e-4, 1e-4), snr=28, lw=100.)]
for c in CASES:
    for seed in (1000, 1001):
        d = synth.make_capture_dev(c["M"], 2 ** 20, nmodes=2, snr_db=c["snr"], theta=np.pi / 5.6, dgd=30e-12, linewidth=c["lw"], seed=seed)
        pit = [dict() for _ in c["methods"]]
        pit[0].update(acq_chunk=1024, acq_plateau=0.999, acq_max=16384)
        rx = ResidentReceiver(2, 2 ** 21, 2, c["M"], c["ntaps"], c["mu"], tier="b", pit=pit, methods=c["methods"], Niter=(1,) * len(c["methods"]), Mtestangles=32, Nbps=20, alphabet=d["alphabet_host"])
        rx.E.copy_from(d["E"]); rx.run(); _lib.sync()
        r = rx.pit_reports()[0]
        e = r["acquisition"]["mean_sq_err"]
        print("##", c["name"], seed, r["acquisition"]["steps"], "mu_acq %.2e" % r["acquisition"]["mu"], [round(x / e[-1], 2) for x in e], flush=True)
        del rx
