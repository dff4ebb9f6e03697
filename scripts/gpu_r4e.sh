#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4e; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
QAMPY_HIP_PIT_DUMP=/tmp/dump FULL=1 PITALL='{"exact_redo_off":1, "max_passes":8}' ONLY="64qam mcma->sbd" timeout 600 python scripts/pit_methods.py > $R/dump_run.txt 2>&1
python scripts/pit_dump_analyse.py /tmp/dump 4 > $R/dump_analysis.txt 2>&1
rm -f /tmp/dump*
QAMPY_HIP_PIT_DUMP=/tmp/dumq FULL=1 PITALL='{"exact_redo_off":1, "max_passes":8, "phase_seed":0}' ONLY="64qam mcma->sbd" timeout 600 python scripts/pit_methods.py > $R/dump_run_noseed.txt 2>&1
python scripts/pit_dump_analyse.py /tmp/dumq 4 > $R/dump_analysis_noseed.txt 2>&1
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 0 --steps 3 > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_timeline.py $DB > $R/c3_timeline.txt 2>&1
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt 2>&1
rm -rf $R/kt
ls -la $R
