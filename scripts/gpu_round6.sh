#!/bin/bash
# Round-6 evidence set (everything lands in gpurun_out/r06e/; scripts/install_profiles.sh copies what is to be judged into profiles/):
#   GPU test suite, PMC instruction and traffic passes, default bench line, kernel statistics of the same bench command + steady-state timeline,
#   c5 line, tolerance / recipe tables.
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06e; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > $R/gpu_tests.txt 2>&1; tail -3 $R/gpu_tests.txt
bash scripts/gpu_pmc_valu.sh > $R/pmc_valu.log 2>&1
bash scripts/gpu_pmc.sh > $R/pmc.log 2>&1
cp gpurun_out/pmc_traffic_c3.json gpurun_out/pmc_instr_c3.json gpurun_out/pmc_summary.txt gpurun_out/pmc_valu_summary.txt $R/ 2>/dev/null
cp gpurun_out/pmc_traffic_c3.json gpurun_out/pmc_instr_c3.json profiles/ 2>/dev/null      # the bench line below quotes them (same kernel sources)
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmcv_*
timeout 1500 python bench.py --detail-out $R/bench_c3_detail.json > $R/bench_c3_line.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err; wc -c $R/bench_c3_line.json
python scripts/show_bench.py $R/bench_c3_detail.json > $R/bench_c3.txt 2>&1
( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 0 --steps 5 > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt
find $R/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/c3_rocprofv3_kernel_stats.csv
rm -rf $R/kt
( cd /tmp; timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$R/kt2 -o c3 -- python $GRAFT_REPO_ROOT/scripts/prefetch_trace.py c3 6 > $GRAFT_REPO_ROOT/$R/trace_run.txt 2> $GRAFT_REPO_ROOT/$R/kt2.log )
DB=$(find $R/kt2 -name "*results.db" | head -1)
python scripts/rocpd_timeline.py $DB pit_setup_kernel 7 > $R/c3_timeline.txt
rm -rf $R/kt2
timeout 900 python bench.py --workload c5 --steps 3 --detail-out $R/bench_c5_detail.json > $R/bench_c5_line.json 2> $R/bench_c5.err
bash scripts/gpu_bps_prof.sh r06e_bps > /dev/null 2>&1; cp gpurun_out/r06e_bps/run.txt $R/bps_run.txt; cp gpurun_out/r06e_bps/kernel_stats.txt $R/bps_kernel_stats.txt; cp gpurun_out/r06e_bps/pmc.txt $R/bps_pmc.txt
timeout 600 python scripts/share_probe.py c3 40 1e-4 1 2 3 > $R/in_flight.txt 2>&1
FULL=1 timeout 900 python scripts/pit_methods.py > $R/pit_methods.txt 2>&1
ls -la $R
