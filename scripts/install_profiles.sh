#!/bin/bash
# copy the round's evidence set from gpurun_out/r05e (scripts/gpu_round5.sh) into profiles/
set -e
R=gpurun_out/r05e
cp $R/pmc_traffic_c3.json profiles/pmc_traffic_c3.json
cp $R/pmc_instr_c3.json profiles/pmc_instr_c3.json
cp $R/c3_kernel_stats.txt profiles/r05_c3_kernel_stats.txt
cp $R/c3_rocprofv3_kernel_stats.csv profiles/r05_c3_rocprofv3_kernel_stats.csv 2>/dev/null || true
cp $R/c3_timeline.txt profiles/r05_c3_timeline.txt
cp $R/pmc_summary.txt profiles/r05_pmc_counters_c3.txt
cp $R/pmc_valu_summary.txt profiles/r05_pmc_instr_c3.txt
cp $R/pit_methods.txt profiles/r05_pit_methods.txt
cp $R/in_flight.txt profiles/r05_in_flight.txt
cp $R/gpu_tests.txt profiles/r05_gpu_tests.txt
grep "^{" $R/bench_c3.json | tail -1 > profiles/r05_bench_c3.json
grep "^{" $R/bench_c5.json | tail -1 > profiles/r05_bench_c5.json
python scripts/show_bench.py profiles/r05_bench_c3.json
