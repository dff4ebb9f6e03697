#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4f; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py tests/test_abi.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
echo "### defaults (way outs on)" > $R/pit_model.txt
FULL=1 timeout 900 python scripts/pit_methods.py >> $R/pit_model.txt 2>&1
timeout 600 python bench.py --bank 0 --no-cpu-baseline --no-extra-shapes > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 0 --steps 3 > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_timeline.py $DB > $R/c3_timeline.txt 2>&1
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt 2>&1
rm -rf $R/kt
ls -la $R
