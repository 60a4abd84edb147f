#!/bin/bash
# round 4: compile-time knobs of the chains re-measured under the free-running half-batch launches (v1: no GCN pressure trackers, v2: unpacked tanh everywhere, v3: aux re-read)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ak; mkdir -p $O
for rep in 1 2 3; do for suf in "" _v1 _v2 _v3; do
  NA_LIB_SUFFIX=$suf timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-parity-check < /dev/null > $O/b.json 2>/dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('standard', '${suf:-default}', round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4))"
done; done 2>&1 | tee $O/ab.txt
