#!/bin/bash
# sweep started without the acquisition run: accuracy against the exact path, all shapes
cd $GRAFT_REPO_ROOT
R=gpurun_out/noacq; rm -rf $R; mkdir -p $R
timeout 1200 python scripts/pit_exp.py --seeds 1000,1001,1002 --variants default,noacq 2>&1 | grep "^##" | sed "s/^##/## c3/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --snr 24 --nsym 2097152 --seeds 1001 --variants default,noacq 2>&1 | grep "^##" | sed "s/^##/## c3snr24/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --workload ns --seeds 1000 --variants default,noacq 2>&1 | grep "^##" | sed "s/^##/## ns/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --workload c2 --seeds 1000,1001 --variants default,noacq 2>&1 | grep "^##" | sed "s/^##/## c2/" >> $R/out.txt
