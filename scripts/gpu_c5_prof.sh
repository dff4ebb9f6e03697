#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/c5p; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$R/b.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/stats.txt
python scripts/rocpd_timeline.py $DB frame_sync_none 1 > $R/timeline.txt 2>/dev/null
rm -rf $R/kt
