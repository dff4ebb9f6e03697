#!/bin/bash
# Round profile set (everything lands in gpurun_out/prof_r/; copy what is to be judged into profiles/):
#   kernel statistics + timeline of the default bench command, PMC traffic passes, bench lines of every workload.
cd $GRAFT_REPO_ROOT
R=gpurun_out/prof_r; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/kt -o c3 -- python $GRAFT_REPO_ROOT/bench.py --bank 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$R/kt_bench.json 2> $GRAFT_REPO_ROOT/$R/kt.log )
DB=$(find $R/kt -name "*results.db" | head -1)
python scripts/rocpd_stats.py $DB > $R/c3_kernel_stats.txt
python scripts/rocpd_timeline.py $DB > $R/c3_timeline.txt
find $R/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/c3_rocprofv3_kernel_stats.csv
bash scripts/gpu_pmc.sh > $R/pmc.log 2>&1
cp gpurun_out/pmc_traffic_c3.json gpurun_out/pmc_summary.txt $R/ 2>/dev/null
for w in c3 ns c2 c5; do timeout 900 python bench.py --workload $w > $R/bench_$w.json 2> $R/bench_$w.err; done
rm -rf $R/kt
ls -la $R
