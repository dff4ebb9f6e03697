#!/bin/bash
# kernel trace of a short default bench run -> per-kernel statistics + timeline of the last capture (gpurun_out/<dir>/)
cd $GRAFT_REPO_ROOT
R=gpurun_out/${1:-tl}; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 0 --steps 5 ${@:2} > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt
python scripts/rocpd_timeline.py $DB > $R/c3_timeline.txt
find $R/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/c3_rocprofv3_kernel_stats.csv
rm -rf $R/kt
tail -2 $R/kt.log; wc -l $R/c3_timeline.txt
