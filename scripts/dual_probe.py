"""Two captures in flight on one GPU (two processes, one receiver each): aggregate throughput against one process.
Usage: python scripts/dual_probe.py [nproc] [workload] [steps]"""
import os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(here, ".."))
    import bench
    from qampy_amd import _lib
    key, K, start = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    cfg = bench.WORKLOADS[key]
    sig = bench.make_input(cfg, cfg["nsym"], 1000, host=False)
    rx = bench.make_receiver(cfg, sig, tier="b", pit={})
    rx.load(sig)
    for _ in range(3):
        rx.run(overlap=True)
    rx.wait_post(); _lib.sync()
    while time.time() < start:
        pass
    t0 = time.time()
    for _ in range(K):
        rx.run(overlap=True)
    rx.wait_post(); _lib.sync()
    t1 = time.time()
    reps = rx.pit_reports()
    print("child %.3f %.3f ms/step passes %s certified %s" % (t0 - start, (t1 - t0) / K * 1e3, [r["passes"] for r in reps], all(r["converged"] == 1 for r in reps)), flush=True)
    print("SPAN %.6f %.6f" % (t0, t1), flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
key = sys.argv[2] if len(sys.argv) > 2 else "c3"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = time.time() + 25.0
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", key, str(K), repr(start)], stdout=subprocess.PIPE, text=True) for _ in range(n)]
outs = [p.communicate()[0] for p in ps]
spans = []
for o in outs:
    for l in o.splitlines():
        if l.startswith("SPAN"):
            spans.append(tuple(float(x) for x in l.split()[1:]))
        else:
            print(l)
nsym = {"c3": 2 ** 22, "c2": 2 ** 20, "ns": 10 ** 7}[key]
t0, t1 = min(a for a, _ in spans), max(b for _, b in spans)
print("%d process(es), %s: %d captures in %.1f ms -> %.1f MSym/s aggregate, %.3f ms per capture" % (n, key, n * K, (t1 - t0) * 1e3, n * K * nsym / (t1 - t0) / 1e6, (t1 - t0) / (n * K) * 1e3))
