#!/bin/bash
# round 4, fifth GPU call: compact rings -- the suite, then old rings / new rings A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
NA_AB_ARGS="--no-parity-check" timeout 600 bash tools/ab_bench.sh "- _oldrings" 1000 2>&1 | tee $O/ab_standard.txt
NA_AB_ARGS="--workload config3 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _oldrings" 300 2>&1 | tee $O/ab_config3.txt
NA_AB_ARGS="--workload lite --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _oldrings" 500 2>&1 | tee $O/ab_lite.txt
NA_AB_ARGS="--workload feather --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _oldrings" 500 2>&1 | tee $O/ab_feather.txt
NA_AB_ARGS="--workload standard --streams 1152 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _oldrings" 500 2>&1 | tee $O/ab_std1152.txt
python bench.py --steps 500 | head -c 3000
