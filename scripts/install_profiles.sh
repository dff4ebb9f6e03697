#!/bin/bash
# copy the round's evidence set from gpurun_out/r06e (scripts/gpu_round6.sh) into profiles/
set -e
R=gpurun_out/r06e
cp $R/pmc_traffic_c3.json profiles/pmc_traffic_c3.json
cp $R/pmc_instr_c3.json profiles/pmc_instr_c3.json
cp $R/c3_kernel_stats.txt profiles/r06_c3_kernel_stats.txt
cp $R/c3_rocprofv3_kernel_stats.csv profiles/r06_c3_rocprofv3_kernel_stats.csv 2>/dev/null || true
cp $R/c3_timeline.txt profiles/r06_c3_timeline.txt
cp $R/pmc_summary.txt profiles/r06_pmc_counters_c3.txt
cp $R/pmc_valu_summary.txt profiles/r06_pmc_instr_c3.txt
cp $R/pit_methods.txt profiles/r06_pit_methods.txt
cp $R/in_flight.txt profiles/r06_in_flight.txt
cp $R/gpu_tests.txt profiles/r06_gpu_tests.txt
cp $R/bench_c3_line.json profiles/r06_bench_c3_line.json
cp $R/bench_c3_detail.json profiles/r06_bench_c3_detail.json
cp $R/bench_c3.txt profiles/r06_bench_c3.txt
cp $R/bench_c5_line.json profiles/r06_bench_c5_line.json
cp $R/bench_c5_detail.json profiles/r06_bench_c5_detail.json
cat $R/bps_run.txt $R/bps_kernel_stats.txt $R/bps_pmc.txt > profiles/r06_bps_profile.txt
