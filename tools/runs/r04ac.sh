#!/bin/bash
# round 4: WaveNet layer arrays of 65 .. 128 channels (WaveNetWideKernel) -- parity, real-time check, timings
cd /root/repo; O=gpurun_out/r04ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavenet or wide or channels" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
for c in "64 32" "96 48" "128 64"; do for S in 64 256 512; do python tools/quick_time.py $c $S 2>&1 | tail -1; done; done | tee $O/times.txt
