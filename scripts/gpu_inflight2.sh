#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06i; mkdir -p $R
run() { tag=$1; shift; python bench.py --no-extra-shapes --bank 0 --no-cpu-baseline --exact-steps 0 --warmup 3 --detail-out $R/detail_$tag.json "$@" > $R/line_$tag.json 2> $R/err_$tag.txt
  python -c "
import json;d=json.load(open('$R/line_$tag.json'));print('$tag', d['value'], d['ms_per_step'], d['tier_b']['certified'], d['roofline']['launch_ms'])"; tail -1 $R/err_$tag.txt; }
run f3_p1_s24 --in-flight 3 --pool 1 --steps 24
run f3_p1_s120 --in-flight 3 --pool 1 --steps 120
run f3_p8_s120 --in-flight 3 --pool 8 --steps 120
run f1_p8_s120 --in-flight 1 --pool 8 --steps 120
run f1_p1_s120 --in-flight 1 --pool 1 --steps 120
python scripts/share_probe.py c3 40 1e-4 1 3 2>&1 | grep -v "each on"
