#!/bin/bash
cd /root/repo; O=gpurun_out/r04v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
NA_HOST_DIRECT=0 timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "half or halves" > $O/pytest2.log 2>&1; echo "pytest (copy engines) rc $?"; tail -3 $O/pytest2.log
for rep in 1 2 3; do
  for v in "default" "NA_HOST_HALVES=0"; do
    echo "== $v"; if [ "$v" = default ]; then neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; else env $v neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; fi
  done
done | tee $O/hostpipe.txt
