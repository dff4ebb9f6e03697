set -x
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
for w in c3 ns c2 c5; do timeout 900 python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; tail -c 1500 gpurun_out/bench_$w.json; tail -3 gpurun_out/bench_$w.err; done
