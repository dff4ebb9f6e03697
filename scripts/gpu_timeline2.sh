#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/${1:-tl2}; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/scripts/prefetch_trace.py c3 6 > $GRAFT_REPO_ROOT/$R/run.txt 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_timeline.py $DB pit_setup_kernel 7 > $R/timeline.txt
rm -rf $R/kt
cat $R/run.txt; wc -l $R/timeline.txt
