#!/bin/bash
# round 4: free-running half chains in the pipelined host interface -- tests, HostPipeBench with and without (NA_HOST_HALVES=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for rep in 1 2 3; do
  for h in 1 0; do
    echo "halves=$h"; NA_HOST_HALVES=$h neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000 | tee -a $O/hostpipe_halves$h.txt
  done
done
