#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/ad2; rm -rf $R; mkdir -p $R
for sd in 1000 1001 1002; do for l in 17 20 22; do echo "== seed $sd log2 $l full recipe" >> $R/out.txt; timeout 300 python scripts/adapt_exp.py --log2 $l --seed $sd 2>&1 | grep "^##" | cut -c1-3600 >> $R/out.txt; done; done
timeout 1500 python -m pytest tests/test_gpu_pit.py -m gpu -x -q > $R/pit.log 2>&1; tail -n 3 $R/pit.log
