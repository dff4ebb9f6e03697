#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/ad2; rm -rf $R; mkdir -p $R
run() { echo "== $*" >> $R/out.txt; env $1 timeout 300 python scripts/adapt_exp.py --log2 20 --methods mcma --modes 0 --pit max_passes=16 2>&1 | grep "^##" | cut -c1-2600 >> $R/out.txt; }
run QAMPY_HIP_PIT_ADAPT_NEWTON=1
run QAMPY_HIP_PIT_ADAPT_NEWTON=0
echo "== full" >> $R/out.txt; timeout 300 python scripts/adapt_exp.py --log2 20 2>&1 | grep "^##" | cut -c1-2600 >> $R/out.txt
