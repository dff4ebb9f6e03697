#!/bin/bash
# experiment: segment grids with more chains than SIMDs (needs a build with -DQH_SEG_DUAL to let two waves share a SIMD)
cd $GRAFT_REPO_ROOT
R=gpurun_out/dual; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp QAMPY_HIP_SEG_LANES=16 QAMPY_HIP_PIT_TIMING=all
timeout 900 python scripts/pit_exp.py --workload c3 --variants "$1" --reps 3 2> $R/err.txt | grep "^##" > $R/out.txt
cat $R/out.txt | cut -c1-1500
tail -3 $R/err.txt
