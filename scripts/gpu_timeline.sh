#!/bin/bash
# kernel timeline + statistics of the default bench command (tier b only, few steps) -> gpurun_out/tl/
cd $GRAFT_REPO_ROOT
R=gpurun_out/tl; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 0 --steps 5 "$@" > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt
python scripts/rocpd_timeline.py $DB > $R/c3_timeline.txt
rm -rf $R/kt
