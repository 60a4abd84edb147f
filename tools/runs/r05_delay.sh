#!/bin/bash
# round 5: start offset of the second workgroup of every CU inside the resident launch (NA_RESIDENT_DELAY_US), one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05a}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-path --no-exact-f32 --no-parity-check --rotate 0"
for rep in 1 2; do
for d in 0 6 10 13 16 19 24; do
  NA_RESIDENT=1 NA_RESIDENT_DELAY_US=$d timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('delay $d', d['launch_mode'], round(d['ms_per_step']*1e3,2), round(d['latency_per_buffer_ms']*1e3,2))" >> $O/delay.txt
done
NA_RESIDENT=0 timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains', d['launch_mode'], round(d['ms_per_step']*1e3,2), round(d['latency_per_buffer_ms']*1e3,2))" >> $O/delay.txt
done
cat $O/delay.txt
