#!/bin/bash
# The rotation regime (one batch of 8 x 1024 A1 Standard streams: 2 GB of state walked once per step, nothing of it survives in the 256 MB
# Infinity Cache) under rocprofv3: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in their own passes.  tools/profile_rotation.sh <tag>
#   gpurun_out/<tag>/time.txt  kernel_stats.csv  pmc_summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/quick_time_own.py BossWN-standard.nam 8192"
K=100 $CMD > $OUT/time.txt 2>/dev/null
rm -rf $OUT/stats
K=60 rocprofv3 --kernel-trace --stats -d $OUT/stats -o prof -- $CMD > $OUT/stats.log 2>&1
DB=$(find $OUT/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats.csv "K=60 tools/quick_time_own.py BossWN-standard.nam 8192 (240 steps of 8192 streams, two half-batch launches each) under rocprofv3 --kernel-trace --stats" 2>> $OUT/stats.log
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  K=10 rocprofv3 --kernel-trace --pmc $group -f csv -d $OUT/pmc/pass$i -o pmc -- $CMD > $OUT/pmc_pass$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT/pmc WaveNetSpecKernel > $OUT/pmc_summary.txt
rm -rf $OUT/stats $OUT/pmc/pass*/
cat $OUT/time.txt $OUT/kernel_stats.csv $OUT/pmc_summary.txt
