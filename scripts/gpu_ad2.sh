#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/ad2; rm -rf $R; mkdir -p $R
export QAMPY_HIP_PIT_NOSTALL=1 QAMPY_HIP_PIT_ADAPT_NOFALLBACK=1
for l in 17 20; do echo "== log2 $l" >> $R/out.txt; timeout 300 python scripts/adapt_exp.py --log2 $l --methods mcma --modes 0 --pit max_passes=24 2>&1 | grep "^##" | cut -c1-3600 >> $R/out.txt; done
