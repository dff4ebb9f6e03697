#!/bin/bash
# eigen-solver sweeps against passes / time: all three shapes, two seeds
cd $GRAFT_REPO_ROOT
R=gpurun_out/sw; rm -rf $R; mkdir -p $R
for n in ${SWEEPS:-5 4 3}; do
  export QAMPY_HIP_PIT_EIGSWEEPS=$n
  timeout 900 python scripts/pit_exp.py --seeds 1000,1001,1002 --variants default --no-exact 2>&1 | grep "^##" | sed "s/^##/## sweeps=$n c3/" >> $R/out.txt
  timeout 900 python scripts/pit_exp.py --workload ns --seeds 1000 --variants default --no-exact 2>&1 | grep "^##" | sed "s/^##/## sweeps=$n ns/" >> $R/out.txt
  timeout 900 python scripts/pit_exp.py --workload c2 --seeds 1000,1001 --variants default --no-exact 2>&1 | grep "^##" | sed "s/^##/## sweeps=$n c2/" >> $R/out.txt
done
