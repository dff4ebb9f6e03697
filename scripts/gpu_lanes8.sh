#!/bin/bash
# 8-lane hand-scheduled blocks (round 6): correctness (tier-b tests), NS timing, C3 with 8 lanes per chain and twice the segments
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06j; mkdir -p $R
python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py -q -x > $R/tests.txt 2>&1; tail -5 $R/tests.txt
run() { tag=$1; shift; python bench.py --no-extra-shapes --bank 0 --no-cpu-baseline --exact-steps 1 --warmup 3 --steps 16 --detail-out $R/detail_$tag.json "$@" > $R/line_$tag.json 2> $R/err_$tag.txt
  python -c "
import json;d=json.load(open('$R/line_$tag.json'));dd=json.load(open('$R/detail_$tag.json'));print('$tag', d['value'], d['ms_per_step'], d['tier_b']['certified'], d['tier_b']['passes'], [ (s['S'],s['pass_ms']) for s in dd['tier_b']['stages']], d['one_capture_at_a_time'])"; tail -1 $R/err_$tag.txt; }
run ns --workload ns
run c3
QAMPY_HIP_SEG_LANES=8 run c3_l8
