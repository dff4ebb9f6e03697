#!/bin/bash
# bench.py with 1..4 captures in flight (pool of different captures), then the phase-search profile.  Output: gpurun_out/r06h/
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06h; mkdir -p $R
for n in 1 2 3 4; do
  python bench.py --in-flight $n --no-extra-shapes --bank 0 --no-cpu-baseline --steps 24 --warmup 3 --detail-out $R/detail_$n.json > $R/line_$n.json 2> $R/err_$n.txt
  python -c "
import json;d=json.load(open('$R/line_$n.json'));print($n, d['value'], d['ms_per_step'], d['tier_b']['certified'], d['roofline']['launch_ms'], d.get('note'))"
  tail -2 $R/err_$n.txt
done
bash scripts/gpu_bps_prof.sh r06g > /dev/null 2>&1; cat gpurun_out/r06g/run.txt; head -8 gpurun_out/r06g/kernel_stats.txt; grep bps_stream gpurun_out/r06g/pmc.txt
