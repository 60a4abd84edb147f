#!/bin/bash
# round 4: short histories in LDS -- spec / parity suites, then A/B against the register path (NA_SPK_LDSH=0 build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_spec.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fixtures.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
NA_AB_ARGS="--no-parity-check" timeout 900 bash tools/ab_bench.sh "- _nolh" 1000 2>&1 | tee $O/ab_standard.txt
NA_AB_ARGS="--workload lite --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nolh" 500 2>&1 | tee $O/ab_lite.txt
NA_AB_ARGS="--workload standard --streams 512 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nolh" 500 2>&1 | tee $O/ab_std512.txt
NA_AB_ARGS="--workload standard --streams 64 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nolh" 500 2>&1 | tee $O/ab_std64.txt
