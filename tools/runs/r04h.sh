#!/bin/bash
# round 4, eighth GPU call: one-tile-per-wave chain for packed Nano (tests + A/B against NA_SP_NO_T1), full suite, library vs Standard-only library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
for w in nano; do for s in 1024 512 256 64; do
NA_AB_ARGS="--workload $w --streams $s --no-parity-check" timeout 600 bash tools/ab_bench.sh "- -@NA_SP_NO_T1=1" 500 2>&1 | tee -a $O/ab_nano.txt
done; done
NA_AB_ARGS="--no-parity-check" timeout 600 bash tools/ab_bench.sh "- _quick" 1000 2>&1 | tee $O/ab_quick.txt
