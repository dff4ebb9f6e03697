#!/bin/bash
# One GPU session: tests, bench, rocprofv3 kernel trace.  Run through gpurun from the repo root.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "bench rc=$?"
cat gpurun_out/bench_c3.json; tail -5 gpurun_out/bench_c3.err
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
cat gpurun_out/bench_c2.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_c3 | head -20
