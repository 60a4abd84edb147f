#!/bin/bash
# Profile set of the runtime-shaped wide WaveNet kernel (run on the GPU box):  tools/profile_wide.sh r03_wide
#   gpurun_out/<tag>/timing.txt             tools/quick_time_wide.py (us per 128-sample step, parity against the oracle)
#   gpurun_out/<tag>/kernel_stats_<c>.csv   rocprofv3 --kernel-trace --stats of the same command, per channel count
#   gpurun_out/<tag>/pmc_summary_<c>.txt    PMC counters (separate --pmc passes, no trace domains mixed in)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $OUT/timing.txt
for cfg in "32 16" "64 32"; do
  set -- $cfg
  C=$1; H=$2
  python $R/tools/quick_time_wide.py $C $H 256 >> $OUT/timing.txt 2>&1
  rm -rf $OUT/stats
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o prof -- python $R/tools/quick_time_wide.py $C $H 256 > $OUT/stats_$C.log 2>&1
  DB=$(find $OUT/stats -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats_$C.csv "python tools/quick_time_wide.py $C $H 256 under rocprofv3 --kernel-trace --stats" 2>> $OUT/stats_$C.log
  i=0
  rm -rf $OUT/pmc$C; mkdir -p $OUT/pmc$C
  while read -r group; do
    [ -z "$group" ] && continue
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $group -f csv -d $OUT/pmc$C/pass$i -o pmc -- python $R/tools/quick_time_wide.py $C $H 256 > $OUT/pmc$C/pass$i.log 2>&1
  done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
FETCH_SIZE
WRITE_SIZE
GROUPS
  python $R/tools/pmc_summary.py $OUT/pmc$C WaveNetGenericKernel > $OUT/pmc_summary_$C.txt 2>&1
  rm -rf $OUT/stats $OUT/pmc$C/pass*/
done
cat $OUT/timing.txt; head -6 $OUT/kernel_stats_*.csv; cat $OUT/pmc_summary_*.txt
