#!/bin/bash
# iteration check: tier-b tests, then the default bench line without the CPU legs
cd $GRAFT_REPO_ROOT
R=gpurun_out/iter; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py -q -m gpu -x > $R/gpu_tests.txt 2>&1; tail -3 $R/gpu_tests.txt
timeout 900 python bench.py --bank 0 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1; head -60 $R/bench_c3.txt
