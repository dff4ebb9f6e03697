#!/bin/bash
# One GPU session: smoke, tests, bench (C3 default + C2), rocprofv3 kernel trace.  Run through gpurun from the repo root:
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh <tag>'
cd $GRAFT_REPO_ROOT
TAG=${1:-run}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "bench rc=$?"
cat gpurun_out/${TAG}_bench_c3.json
timeout 600 python bench.py --workload c2 --no-cpu-baseline --bank 0 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
cat gpurun_out/${TAG}_bench_c2.json
timeout 600 python bench.py --workload c1 --bank 0 > gpurun_out/${TAG}_bench_c1.json 2> gpurun_out/${TAG}_bench_c1.err
timeout 600 python scripts/bench_methods.py > gpurun_out/${TAG}_methods.json 2> /dev/null
timeout 600 python scripts/bench_framesync.py > gpurun_out/${TAG}_framesync.json 2> /dev/null
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --bank 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py gpurun_out/${TAG}_prof/c3_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
cat gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
