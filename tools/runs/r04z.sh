#!/bin/bash
# round 4: recurrent layers of up to 1024 units (a workgroup of up to 16 waves per stream) -- parity, then ms per 128-sample block at 64 / 256 streams
cd /root/repo; O=gpurun_out/r04z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recurrent or lstm or gru or keras" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
for m in lstm:1:128 lstm:2:64 lstm:1:256 lstm:2:256 lstm:1:512 lstm:1:1024 gru:1:128 gru:1:256 gru:2:200; do python tools/quick_time_recurrent.py $m 64 256 2>&1 | grep " x " | sed 's/ | four.*//'; done | tee $O/times.txt
