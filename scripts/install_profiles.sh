#!/bin/bash
# copy the round's profile set from gpurun_out/prof_r (scripts/gpu_profiles.sh + a final `bench.py` run) into profiles/
set -e
R=gpurun_out/prof_r
cp $R/pmc_traffic_c3.json profiles/pmc_traffic_c3.json
cp $R/c3_kernel_stats.txt profiles/r02_c3_kernel_stats.txt
cp $R/c3_timeline.txt profiles/r02_c3_timeline.txt
cp $R/pmc_summary.txt profiles/r02_pmc_counters_c3.txt
tail -1 $R/bench_c3_final.json > profiles/r02_bench_c3.json
for w in ns c2 c5; do tail -1 $R/bench_$w.json > profiles/r02_bench_$w.json; done
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open('profiles/r02_bench_c3.json'))
print("c3", d["value"], d["ms_per_step"], "traffic", d["roofline"]["traffic"], "sha", json.load(open('profiles/pmc_traffic_c3.json'))["kernel_sources_sha"], bench.kernel_sources_sha())
PY
