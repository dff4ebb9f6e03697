"""n receivers in flight (ReceiverGroup), optional pass baton: steady state for a rocprofv3 kernel trace.  Usage: group_trace.py n baton captures_per_receiver"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
from qampy_amd.pipeline import ReceiverGroup
n, baton, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = bench.WORKLOADS["c3"]
sig = bench.make_input(cfg, cfg["nsym"], 1000)
assert baton == 0, "the pass baton was measured and dropped (profiles/r06_pass_baton.txt)"
g = ReceiverGroup(n, sig.shape[0], sig.shape[1], 2, cfg["M"], cfg["ntaps"], cfg["mu"], methods=cfg["methods"], Niter=cfg["niter"], adaptive_stepsize=cfg["adaptive"],
                  TrSyms=(None, None), Mtestangles=cfg["A"], Nbps=cfg["Nbps"], dtype=np.complex64, alphabet=sig.coded_symbols, tier="b", pit=dict(tol=1e-4))
for r in g.rx:
    r.post_parts = parts
g.load(sig)
g.run(n * K, prefetch=True)
g.close()
