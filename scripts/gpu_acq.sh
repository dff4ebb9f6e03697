#!/bin/bash
# length of the acquisition run against passes / time
cd $GRAFT_REPO_ROOT
R=gpurun_out/acq; rm -rf $R; mkdir -p $R
V=default,a:1024:1024,a:2048:2048,a:1536:3072,a:2048:4096,a:3072:6144
timeout 1500 python scripts/pit_exp.py --seeds 1000,1001,1002 --no-exact --variants $V 2>&1 | grep "^##" | sed "s/^##/## c3/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --snr 24 --nsym 2097152 --seeds 1001 --no-exact --variants $V 2>&1 | grep "^##" | sed "s/^##/## c3snr24/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --workload ns --seeds 1000 --no-exact --variants $V 2>&1 | grep "^##" | sed "s/^##/## ns/" >> $R/out.txt
timeout 1200 python scripts/pit_exp.py --workload c2 --seeds 1000,1001 --no-exact --variants $V 2>&1 | grep "^##" | sed "s/^##/## c2/" >> $R/out.txt
