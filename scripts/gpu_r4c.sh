#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4c; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
echo "### measured model, no exact redo" > $R/pit_model.txt
FULL=1 PITALL='{"exact_redo_off":1}' timeout 600 python scripts/pit_methods.py >> $R/pit_model.txt 2>&1
timeout 600 python bench.py --bank 0 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
timeout 300 python scripts/pit_survey.py > $R/pit_survey.txt 2>&1
# what holds the sbd stage's estimate up
QAMPY_HIP_PIT_DUMP=/tmp/dump FULL=1 PITALL='{"exact_redo_off":1}' ONLY="64qam mcma->sbd" timeout 600 python scripts/pit_methods.py > $R/dump_run.txt 2>&1
python scripts/pit_dump_analyse.py /tmp/dump > $R/dump_analysis.txt 2>&1
ls -la $R
