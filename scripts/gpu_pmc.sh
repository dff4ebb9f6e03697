#!/bin/bash
# HBM traffic counters of one C3 pass (separate --pmc passes, no tracing domains besides the kernel trace).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --bank 0 --exact-steps 1 --no-extra-shapes --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
  echo "$C rc=$?"
done
cd $GRAFT_REPO_ROOT
python scripts/rocpd_pmc.py --json gpurun_out/pmc_traffic_c3.json c3 $(find gpurun_out/pmc_FETCH_SIZE -name "*results.db" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*results.db" | head -1) | tee gpurun_out/pmc_summary.txt
