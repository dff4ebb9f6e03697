#!/usr/bin/env python3
"""One capture over several ranks (qampy_amd.distributed) against the single-process tier-b run of the same capture.
Run under a launcher (RANK / WORLD_SIZE / MASTER_*); every rank may use the same GPU (tests: 2 ranks, socket collectives, one MI355X -
RCCL refuses two ranks on one device; with one GPU per rank pass --backend rccl).

    python -c "from qampy_amd import comm; comm.launch('scripts/split_check.py', ['--same-gpu'], 2)"
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="tcp", choices=["tcp", "rccl", "auto"])
ap.add_argument("--nsym", type=int, default=2 ** 20)
ap.add_argument("--same-gpu", action="store_true", help="all ranks on device 0 (single-GPU box)")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", 0))
from qampy_amd import _lib, synth
from qampy_amd.comm import Comm
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.distributed import SplitCaptureReceiver
from qampy_amd.core import ber_functions as ber

_lib.init(dev)
cm = Comm(device=dev, backend=args.backend)
M, ntaps, mu, nsym = 64, 41, (2e-4, 2e-4), args.nsym
d = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000)
kw = dict(methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"])
res = {}
for name, cls in (("single", ResidentReceiver), ("split", SplitCaptureReceiver)):
    rx = cls(2, 2 * nsym, 2, M, ntaps, mu, **(dict(kw, tier="b") if name == "single" else dict(kw, comm=cm)))
    rx.E.copy_from(d["E"])
    rx.run(); _lib.sync()
    cm.barrier()
    t0 = time.perf_counter()
    rx.run(); _lib.sync()
    cm.barrier()
    el = time.perf_counter() - t0
    r = rx.fetch()
    res[name] = dict(w=r["wxy"], out=r["out"], ms=el * 1e3, rep=rx.pit_reports(),
                     err=[s["errors"] for s in ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)],
                     exch=(getattr(rx, "exchanges", 0), getattr(rx, "exchanged_bytes", 0)))
    del rx
dw = float(np.max(np.abs(res["single"]["w"] - res["split"]["w"])))
do = float(np.sqrt(np.mean(np.abs(res["single"]["out"] - res["split"]["out"]) ** 2)))
same_passes = [a["passes"] for a in res["single"]["rep"]] == [a["passes"] for a in res["split"]["rep"]]
ok = dw < 2e-3 and do < 2e-3 and same_passes and res["single"]["err"] == res["split"]["err"] and all(a["converged"] for a in res["split"]["rep"])
# every rank must hold the same taps
t = res["split"]["w"].view(np.float32).ravel().astype(np.float64)
ident = bool(np.array_equal(cm.allreduce(t, "min"), cm.allreduce(t, "max")))
if rank == 0:
    print(json.dumps(dict(check="split capture vs single process", world=world, backend=cm.backend, ok=bool(ok and ident), ranks_identical=ident,
                          max_tap_diff=dw, out_rms_diff=do, passes=[a["passes"] for a in res["split"]["rep"]], errors=res["split"]["err"],
                          ms_single=round(res["single"]["ms"], 3), ms_split=round(res["split"]["ms"], 3), exchanges=res["split"]["exch"][0],
                          exchanged_MB=round(res["split"]["exch"][1] / 2 ** 20, 1))))
    print("SPLIT_CHECK_OK" if (ok and ident) else "SPLIT_CHECK_FAILED")
cm.close()
