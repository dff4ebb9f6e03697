#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/adp; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o ad -- python $GRAFT_REPO_ROOT/scripts/adapt_exp.py --log2 20 --methods mcma > $GRAFT_REPO_ROOT/$R/o.txt 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/stats.txt
python scripts/rocpd_timeline.py $DB > $R/timeline.txt
rm -rf $R/kt
