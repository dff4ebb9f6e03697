#!/bin/bash
# Instruction counters of one C3 pass (own --pmc passes, kernel trace only): what the VALU-issue rooflines of bench.py are checked against.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
DBS=""
for C in "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/gpurun_out/pmcv_$tag -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --bank 0 --exact-steps 0 --no-extra-shapes --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmcv_$tag.log 2>&1
  echo "$C rc=$?"
  DBS="$DBS $(find $GRAFT_REPO_ROOT/gpurun_out/pmcv_$tag -name '*results.db' | head -1)"
done
cd $GRAFT_REPO_ROOT
python scripts/rocpd_pmc.py --instr-json gpurun_out/pmc_instr_c3.json c3 $DBS > /dev/null
python scripts/rocpd_pmc.py $DBS | grep -i "kernel\|train_seg\|bps_stream\|apply_cplx\|train_la\|pit_jacobi\|unwrap_apply" | tee gpurun_out/pmc_valu_summary.txt
