#!/bin/bash
# Round 4, batch 2: measured coarse model (signal-subspace block) against round 3's model (correction = 2)
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4b; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py tests/test_gpu_split.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
echo "### measured model, no exact redo" > $R/pit_model.txt
FULL=1 PITALL='{"exact_redo_off":1}' timeout 600 python scripts/pit_methods.py >> $R/pit_model.txt 2>&1
echo "### round-3 model (correction = 2), no exact redo" >> $R/pit_model.txt
FULL=1 PITALL='{"exact_redo_off":1, "correction":2}' timeout 600 python scripts/pit_methods.py >> $R/pit_model.txt 2>&1
timeout 600 python bench.py --bank 0 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
timeout 300 python scripts/pit_survey.py > $R/pit_survey.txt 2>&1
ls -la $R
