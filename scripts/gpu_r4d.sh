#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4d; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
timeout 600 python bench.py --bank 0 --no-cpu-baseline --no-extra-shapes > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
timeout 600 python scripts/sens_probe.py > $R/sens.txt 2>&1
timeout 900 python scripts/pit_exp.py --workload c3 --reps 3 --variants default,g:3584:2047:0.001:0.001,g:2688:2047:0.001:0.001,g:1792:4094:0.001:0.001,g:1792:3072:0.001:0.001,g:1344:1536:0.001:0.001 2>&1 | grep "^##" > $R/grid.txt
ls -la $R
