#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06k; mkdir -p $R
python -m pytest tests/test_gpu_pit.py tests/test_gpu_functional.py -q -x -k "overlap or parts or prefetch or group or receiver or pipelin" > $R/tests.txt 2>&1; tail -3 $R/tests.txt
python - <<EOF
import sys, time; sys.path.insert(0, ".")
import numpy as np, bench
from qampy_amd import _lib
cfg = bench.WORKLOADS["c3"]; nsym = cfg["nsym"]
sig = bench.make_input(cfg, nsym, 1000)
for aside in (False, True, False, True):
    rx = bench.make_receiver(cfg, sig, tier="b", pit=dict(tol=1e-4)); rx.load(sig); rx.filter_aside = aside
    for _ in range(3): rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync(); t0 = time.perf_counter()
    for _ in range(40): rx.run(overlap=True, prefetch=True)
    rx.wait_post(); _lib.sync(); el = time.perf_counter() - t0
    r = rx.fetch()
    import hashlib
    print("filter_aside", aside, "%.1f MSym/s %.3f ms" % (40 * nsym / el / 1e6, el / 40 * 1e3), hashlib.sha256(r["out"].tobytes()).hexdigest()[:12], hashlib.sha256(r["eq"].tobytes()).hexdigest()[:12], flush=True)
    del rx
EOF
