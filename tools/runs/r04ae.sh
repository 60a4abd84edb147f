#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ae; mkdir -p $O
timeout 60 tools/microbench/bin/anyorder_probe < /dev/null 2>&1 | tee $O/anyorder.txt
for q in 4 8; do for c in 2 3 4; do for i in 1 2; do
  GPU_MAX_HW_QUEUES=$q NA_HOST_CHAINS=$c timeout 300 python bench.py --no-cpu-baseline --no-host-path > $O/b.json 2>/dev/null < /dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('standard hw queues $q chains $c',round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4))"
done; done; done 2>&1 | tee $O/queues.txt
