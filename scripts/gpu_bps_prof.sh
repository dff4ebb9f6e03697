#!/bin/bash
# Phase search alone (scripts/bps_time.py): kernel statistics and issue / stall counters of bps_stream_kernel.  Output: gpurun_out/$1/
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/${1:-bpsprof}; mkdir -p $R
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/kt -o b -- python $GRAFT_REPO_ROOT/scripts/bps_time.py c3 20 > $R/run.txt 2> $R/kt.log
DB=$(find $R/kt -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $DB | grep -i "kernel\|bps_stream\|unwrap\|analyse\|linspace" > $R/kernel_stats.txt
rm -rf $R/kt
DBS=""
for C in "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/pmc_$tag -o b -- python $GRAFT_REPO_ROOT/scripts/bps_time.py c3 4 > $R/pmc_$tag.log 2>&1
  echo "$C rc=$?"
  DBS="$DBS $(find $R/pmc_$tag -name '*results.db' | head -1)"
done
python $GRAFT_REPO_ROOT/scripts/rocpd_pmc.py $DBS | grep -i "kernel\|bps_stream" > $R/pmc.txt
for C in $R/pmc_*; do [ -d $C ] && rm -rf $C; done
cat $R/run.txt $R/kernel_stats.txt $R/pmc.txt
