"""Phase search alone at a BASELINE shape: ms per launch of the search (+ unwrap / de-rotation) on a resident equalised capture.  Usage: bps_time.py [c3|ns|c2] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
key = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = bench.WORKLOADS[key]
sig = bench.make_input(cfg, cfg["nsym"], 1000)
rx = bench.make_receiver(cfg, sig, tier="b", pit=dict(tol=1e-4))
rx.load(sig)
rx.run()
_lib.sync()
ev = [_lib.Event() for _ in range(reps + 1)]
rx.recover(); _lib.sync()
ev[0].record()
for i in range(reps):
    rx.recover()
    ev[i + 1].record()
_lib.sync()
ms = [ev[i + 1].elapsed_ms(ev[i]) for i in range(reps)]
print("%s: bps_recover (search + unwrap + de-rotation, %d modes x %d symbols x %d angles): min %.3f ms, median %.3f ms" % (key, rx.modes.size, rx.N, cfg["A"], min(ms), float(np.median(ms))))
idx = rx.idx.to_host()
import hashlib
print("idx sha256", hashlib.sha256(idx.tobytes()).hexdigest()[:16], "mean idx", float(idx.mean()))
