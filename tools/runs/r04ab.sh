#!/bin/bash
# round 4: one gate row per lane by default -- parity (all recurrent / keras tests), then timings of the runtime-shaped shapes DESIGN.md quotes, both ways
cd /root/repo; O=gpurun_out/r04ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for m in "lstm:1:18 1024" "lstm:3:16 1024" "lstm:2:40 1024" "lstm:4:32 256" "lstm:3:24 1024" "lstm:2:64 64" "lstm:2:64 1024" "lstm:1:128 64" "gru:2:64 64" "lstm:1:256 64" "lstm:2:256 64"; do
  for rpl in 1 4 8; do echo -n "$m rpl $rpl: "; NA_REC_RPL=$rpl python tools/quick_time_recurrent.py $m 2>&1 | grep " x " | sed 's/ | four.*//; s/.*per wave //'; done
done | tee $O/times.txt
