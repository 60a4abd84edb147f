#!/bin/bash
# round 4: polled waits in the host-buffer entry points (NA_SPIN_WAIT=0: the blocking ones)
cd /root/repo; O=/root/repo/gpurun_out/r04al; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1
for rep in 1 2 3; do for v in 1 0; do echo "== spin $v"; NA_SPIN_WAIT=$v timeout 120 neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000 < /dev/null; done; done | tee $O/hostpipe.txt
echo "== one stream"; for v in 1 0; do NA_SPIN_WAIT=$v timeout 120 neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1 128 3000 < /dev/null; done | tee -a $O/hostpipe.txt
