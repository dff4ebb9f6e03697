#!/bin/bash
# Round 4, first GPU batch: suite with the exact-form way out, default bench line, method survey, diagnosis of the mcma stall (corr_beta)
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4a; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
timeout 600 python bench.py > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
FULL=1 timeout 600 python scripts/pit_methods.py > $R/pit_methods.txt 2>&1
for b in 0.5 1.5 3; do
  echo "### corr_beta $b (mcma stage), no exact redo" >> $R/pit_beta.txt
  FULL=1 PITALL='{"exact_redo_off":1}' PIT1="{\"corr_beta\": $b}" ONLY="mcma" timeout 600 python scripts/pit_methods.py >> $R/pit_beta.txt 2>&1
done
echo "### default, no exact redo" >> $R/pit_beta.txt
FULL=1 PITALL='{"exact_redo_off":1}' timeout 600 python scripts/pit_methods.py >> $R/pit_beta.txt 2>&1
ls -la $R
