#!/bin/bash
# round 4: pipelined host buffers, half-batch launches with full-size (default now) vs half-size workgroups vs no halves, one box
cd /root/repo; O=gpurun_out/r04u; mkdir -p $O
for rep in 1 2 3; do
  for v in "default" "NA_SP_SPB=1" "NA_HOST_HALVES=0"; do
    echo "== $v"; if [ "$v" = default ]; then neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; else env $v neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000; fi
  done
done | tee $O/hostpipe.txt
