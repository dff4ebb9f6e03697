#!/bin/bash
# quick check: tier-b tests, default bench line without the CPU legs, method survey
cd $GRAFT_REPO_ROOT
R=gpurun_out/quick; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -3 $R/gpu_tests.txt
timeout 600 python bench.py --bank 0 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
FULL=1 timeout 900 python scripts/pit_methods.py > $R/pit_methods.txt 2>&1
timeout 600 python scripts/pit_survey.py > $R/pit_survey.txt 2>&1
