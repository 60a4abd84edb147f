#!/bin/bash
# round 4: pipelined host buffers after moving the chain streams' creation to the set-up side; batch + multi tests
cd /root/repo; O=gpurun_out/r04x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for rep in 1 2 3; do
  for v in "default" "NA_HOST_HALVES=0"; do
    echo "== $v"; if [ "$v" = default ]; then neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; else env $v neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; fi
  done
done | tee $O/hostpipe.txt
