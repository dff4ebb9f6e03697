#!/usr/bin/env python3
"""One capture over several ranks (qampy_amd.distributed) against the single-process tier-b run of the same capture.
Run under torch.distributed.run; every rank may use the same GPU (tests: 2 ranks, gloo, one MI355X).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/split_check.py [--backend gloo|nccl] [--nsym N]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--nsym", type=int, default=2 ** 20)
ap.add_argument("--same-gpu", action="store_true", help="all ranks on device 0 (single-GPU box)")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(dev)
dist.init_process_group(args.backend, rank=rank, world_size=world)

from qampy_amd import _lib, synth
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.distributed import SplitCaptureReceiver
from qampy_amd.core import ber_functions as ber

_lib.init(dev)
M, ntaps, mu, nsym = 64, 41, (2e-4, 2e-4), args.nsym
d = synth.make_capture_dev(M, nsym, nmodes=2, snr_db=30, theta=np.pi / 5.6, dgd=30e-12, linewidth=100., seed=1000)
kw = dict(methods=("cma", "mrde"), Niter=(1, 1), Mtestangles=64, Nbps=20, alphabet=d["alphabet_host"])
res = {}
for name, cls in (("single", ResidentReceiver), ("split", SplitCaptureReceiver)):
    rx = cls(2, 2 * nsym, 2, M, ntaps, mu, **(dict(kw, tier="b") if name == "single" else kw))
    rx.E.copy_from(d["E"])
    rx.run(); _lib.sync()
    dist.barrier()
    t0 = time.perf_counter()
    rx.run(); _lib.sync()
    dist.barrier()
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
t = torch.tensor(res["split"]["w"].view(np.float32).ravel().astype(np.float64))
lo, hi = t.clone(), t.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
ident = bool(torch.equal(lo, hi))
if rank == 0:
    print(json.dumps(dict(check="split capture vs single process", world=world, backend=args.backend, ok=bool(ok and ident), ranks_identical=ident,
                          max_tap_diff=dw, out_rms_diff=do, passes=[a["passes"] for a in res["split"]["rep"]], errors=res["split"]["err"],
                          ms_single=round(res["single"]["ms"], 3), ms_split=round(res["split"]["ms"], 3), exchanges=res["split"]["exch"][0],
                          exchanged_MB=round(res["split"]["exch"][1] / 2 ** 20, 1))))
    print("SPLIT_CHECK_OK" if (ok and ident) else "SPLIT_CHECK_FAILED")
dist.destroy_process_group()
