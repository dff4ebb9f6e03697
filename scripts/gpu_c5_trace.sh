#!/bin/bash
# config 5 (pilot chain through the basic API): host profile + kernel statistics of two steps -> gpurun_out/r05/c5_*
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
export TMPDIR=/tmp
python scripts/c5_profile.py > $R/c5_host_profile.txt 2>&1
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/ktc5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$R/c5_kt_bench.json 2> $GRAFT_REPO_ROOT/$R/c5_kt.log )
DB=$(find $R/ktc5 -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/c5_kernel_stats.txt
rm -rf $R/ktc5
head -60 $R/c5_kernel_stats.txt
