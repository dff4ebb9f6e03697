#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/${1:-gt}; mkdir -p $R
export TMPDIR=/tmp
for cfgs in "1 0 12 0" "2 0 10 0" "2 1 10 0" "2 1 10 1" "3 0 8 1"; do
  tag=$(echo $cfgs | tr ' ' '_')
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/kt_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/group_trace.py $cfgs > $GRAFT_REPO_ROOT/$R/run_$tag.txt 2> $GRAFT_REPO_ROOT/$R/kt_$tag.log )
  DB=$(find $R/kt_$tag -name "*results.db" | head -1)
  python scripts/rocpd_passes.py $DB 8 > $R/passes_$tag.txt 2>&1
  python scripts/rocpd_timeline.py $DB pit_setup_kernel 5 | head -150 > $R/timeline_$tag.txt 2>&1
  rm -rf $R/kt_$tag
  echo "== $cfgs"; tail -1 $R/passes_$tag.txt
done
