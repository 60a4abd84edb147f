#!/bin/bash
# One round's judged profile set for the headline bench (run on the GPU box):  tools/profile_round.sh r02_x
# Other workloads:  tools/profile_round.sh r02_cfg3 WaveNetFrameKernel --workload config3   (kernel-name pattern, then bench.py arguments)
#   gpurun_out/<tag>/bench.json            the bench line (default command)
#   gpurun_out/<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>/pmc_summary.txt       PMC counters, one rocprofv3 --pmc pass per counter group (no trace domains mixed in)
# Copy what should be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
KPAT=${2:-WaveNetSpecKernel}
shift; shift
BARGS="$@"
export NA_PMC_BENCH_ARGS="$BARGS"
python $R/bench.py $BARGS > $OUT/bench.json 2> $OUT/bench.err
rm -rf $OUT/stats
# (the traced / counted runs are the timed workload alone: no rotation regime, exact-f32 or resident-launch side runs, whose dispatches of the
# same kernel would be averaged in)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o prof -- python $R/bench.py --no-cpu-baseline --no-parity-check --no-host-path --rotate 0 --no-exact-f32 $BARGS > $OUT/stats.log 2>&1
DB=$(find $OUT/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats.csv "python bench.py --no-cpu-baseline --no-parity-check --no-host-path --rotate 0 --no-exact-f32 $BARGS under rocprofv3 --kernel-trace --stats" 2>> $OUT/stats.log || ls -R $OUT/stats >> $OUT/stats.log
bash $R/tools/pmc_passes.sh $TAG/pmc > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc $KPAT > $OUT/pmc_summary.txt
rm -rf $OUT/stats $OUT/pmc/pass*/  # keep the summaries, not the raw traces
head -c 600 $OUT/bench.json; echo; cat $OUT/kernel_stats.csv | head -8; cat $OUT/pmc_summary.txt
