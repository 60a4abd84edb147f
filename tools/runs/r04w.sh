#!/bin/bash
# round 4: chain start stagger (NA_CHAIN_STAGGER_US) -- pipelined host buffers (2 / 3 in flight) and the device-pointer headline
cd /root/repo; O=gpurun_out/r04w; mkdir -p $O
for rep in 1 2; do
  for st in 16 0 8 24; do
    echo "== stagger $st"; NA_CHAIN_STAGGER_US=$st neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000
  done
done | tee $O/hostpipe.txt | grep "==\|zero_copy" | cut -c1-200
for st in 16 0; do for i in 1 2; do NA_CHAIN_STAGGER_US=$st timeout 300 python bench.py --no-cpu-baseline > $O/b.json 2>/dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('standard stagger $st',round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4))"; done; done
