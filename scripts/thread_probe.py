"""Why is ONE receiver on a ReceiverGroup worker thread slower than the same receiver on the main thread?  (round 6 diagnosis)"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from qampy_amd import _lib
from qampy_amd.pipeline import ReceiverGroup, ResidentReceiver

cfg = bench.WORKLOADS["c3"]; nsym = cfg["nsym"]; K = 40
sig = bench.make_input(cfg, nsym, 1000, host=False)
pit = dict(tol=1e-4)

def loop(rx, k, prefetch):
    for _ in range(k):
        rx.run(overlap=True, prefetch=prefetch)
    rx.wait_post()
    _lib.sync()

for prefetch in (False, True):
    rx = bench.make_receiver(cfg, sig, tier="b", pit=pit); rx.load(sig)
    loop(rx, 3, prefetch); t0 = time.perf_counter(); loop(rx, K, prefetch); el = time.perf_counter() - t0
    print("main thread, prefetch %d: %.1f MSym/s (%.3f ms)" % (prefetch, K * nsym / el / 1e6, el / K * 1e3), flush=True)
    del rx
    # a plain worker thread, receiver built on the main thread
    rx = bench.make_receiver(cfg, sig, tier="b", pit=pit); rx.load(sig); _lib.sync()
    res = {}
    def work():
        if hasattr(rx, "_owner_thread"):
            rx._owner_thread = threading.get_ident()
        loop(rx, 3, prefetch); t0 = time.perf_counter(); loop(rx, K, prefetch); res["el"] = time.perf_counter() - t0
        _lib.call("qh_thread_release")
    t = threading.Thread(target=work); t.start(); t.join()
    print("worker thread, prefetch %d: %.1f MSym/s (%.3f ms)" % (prefetch, K * nsym / res["el"] / 1e6, res["el"] / K * 1e3), flush=True)
    del rx
