#!/bin/bash
# round 4, sixth GPU call: old / exact / exact + compact rings on one box; the reworked A2 saturation test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spec.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
NA_AB_ARGS="--no-parity-check" timeout 900 bash tools/ab_bench.sh "- _exact _oldrings" 1000 2>&1 | tee $O/ab_standard.txt
NA_AB_ARGS="--workload standard --streams 896 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _exact _oldrings" 500 2>&1 | tee $O/ab_std896.txt
NA_AB_ARGS="--workload standard --streams 1152 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _exact _oldrings" 500 2>&1 | tee $O/ab_std1152.txt
NA_AB_ARGS="--workload lite --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _exact _oldrings" 500 2>&1 | tee $O/ab_lite.txt
NA_AB_ARGS="--workload config3 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _exact _oldrings" 300 2>&1 | tee $O/ab_config3.txt
