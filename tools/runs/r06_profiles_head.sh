#!/bin/bash
# end of round 6: the headline set and config 4 (the one workload whose kernel changed after r06_p1) again, at the final tree
cd $GRAFT_REPO_ROOT
T=r06_p2
bash tools/profile_round.sh $T WaveNetSpecKernel > gpurun_out/$T.log 2>&1
bash tools/profile_round.sh ${T}_cfg4 Recurrent --workload config4 --steps 500 > gpurun_out/${T}_cfg4.log 2>&1
for d in $T ${T}_cfg4; do echo "== $d"; head -c 400 gpurun_out/$d/bench.json; echo; head -4 gpurun_out/$d/kernel_stats.csv; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_WAVES |SQ_WAIT_ANY|SQ_WAVE_CYCLES" gpurun_out/$d/pmc_summary.txt; done
