#!/bin/bash
# round 4, second GPU call: range-contract tests, saturation cost A/B, joined half-launches, Infinity Cache probe (4 loads in flight per lane)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spec.py tests/test_gpu_families.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
timeout 300 tools/microbench/bin/mall_working_set > $O/mall.txt 2>&1; cat $O/mall.txt
NA_AB_ARGS="--workload a2full --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nosat" 500 2>&1 | tee $O/ab_a2full.txt
NA_AB_ARGS="--workload config5 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nosat" 300 2>&1 | tee $O/ab_config5.txt
timeout 600 python tools/split_launch_probe.py 2>&1 | tee $O/split_launch.txt
