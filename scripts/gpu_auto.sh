#!/bin/bash
# automatic segment grid: all three shapes, two seeds
cd $GRAFT_REPO_ROOT
R=gpurun_out/auto; rm -rf $R; mkdir -p $R
timeout 900 python scripts/pit_exp.py --seeds 1000,1001 --variants default 2>&1 | grep "^##" > $R/c3.txt
timeout 900 python scripts/pit_exp.py --workload ns --seeds 1000 --variants default 2>&1 | grep "^##" > $R/ns.txt
timeout 900 python scripts/pit_exp.py --workload c2 --seeds 1000,1001 --variants default 2>&1 | grep "^##" > $R/c2.txt
