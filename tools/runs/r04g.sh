#!/bin/bash
# round 4, seventh GPU call: range-event flag in LDS (tests), TRAPSTS probe, stream skew on the compact layout, no-traffic ablation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spec.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
tools/microbench/bin/trapsts_probe | tee $O/trapsts.txt
NA_AB_ARGS="--no-parity-check" timeout 900 bash tools/ab_bench.sh "_quick _skew3 _skew5 _skew7 _abl4" 1000 2>&1 | tee $O/ab_skew.txt
NA_AB_ARGS="--workload a2full --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nosat" 500 2>&1 | tee $O/ab_a2full.txt
NA_AB_ARGS="--workload config5 --no-parity-check" timeout 600 bash tools/ab_bench.sh "- _nosat" 300 2>&1 | tee $O/ab_config5.txt
