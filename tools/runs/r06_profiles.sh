#!/bin/bash
# round 6: the judged profile set (bench line + rocprofv3 --kernel-trace --stats + PMC passes) of every BASELINE config and the narrow models
cd $GRAFT_REPO_ROOT
T=${1:-r06_p1}
bash tools/profile_round.sh $T WaveNetSpecKernel > gpurun_out/$T.log 2>&1
bash tools/profile_round.sh ${T}_cfg3 WaveNetSpecKernel --workload config3 --steps 500 > gpurun_out/${T}_cfg3.log 2>&1
bash tools/profile_round.sh ${T}_cfg4 Recurrent --workload config4 --steps 500 > gpurun_out/${T}_cfg4.log 2>&1
bash tools/profile_round.sh ${T}_cfg5 WaveNetSpecKernel --workload config5 --steps 500 > gpurun_out/${T}_cfg5.log 2>&1
bash tools/profile_round.sh ${T}_nano1024 WaveNetSpecKernel --workload nano --steps 500 > gpurun_out/${T}_nano1024.log 2>&1
bash tools/profile_round.sh ${T}_feather1024 WaveNetSpecKernel --workload feather --steps 500 > gpurun_out/${T}_feather1024.log 2>&1
for d in $T ${T}_cfg3 ${T}_cfg4 ${T}_cfg5 ${T}_nano1024 ${T}_feather1024; do echo "== $d"; head -c 400 gpurun_out/$d/bench.json; echo; head -4 gpurun_out/$d/kernel_stats.csv; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAVES |SQ_WAIT_ANY|SQ_WAVE_CYCLES" gpurun_out/$d/pmc_summary.txt; done
