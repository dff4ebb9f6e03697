#!/bin/bash
# copy the round's evidence set from gpurun_out/r03 (scripts/gpu_round3.sh) into profiles/
set -e
R=gpurun_out/r03
cp $R/pmc_traffic_c3.json profiles/pmc_traffic_c3.json
cp $R/pmc_instr_c3.json profiles/pmc_instr_c3.json
cp $R/c3_kernel_stats.txt profiles/r03_c3_kernel_stats.txt
cp $R/c3_timeline.txt profiles/r03_c3_timeline.txt
cp $R/pmc_summary.txt profiles/r03_pmc_counters_c3.txt
cp $R/pmc_valu_summary.txt profiles/r03_pmc_instr_c3.txt
grep "^{" $R/bench_c3.json | tail -1 > profiles/r03_bench_c3.json
grep "^{" $R/bench_c5.json | tail -1 > profiles/r03_bench_c5.json
python scripts/show_bench.py profiles/r03_bench_c3.json
