#!/bin/bash
# tier-b regression set + all shapes against the exact path
cd $GRAFT_REPO_ROOT
R=gpurun_out/t; rm -rf $R; mkdir -p $R
timeout 1500 python -m pytest tests/test_gpu_pit.py tests/test_gpu_split.py -m gpu -x -q > $R/pit.log 2>&1
tail -n 5 $R/pit.log
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/bench.json 2> $R/bench.err
