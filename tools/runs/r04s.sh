#!/bin/bash
# round 4: streams per workgroup of the half-batch launches (the heuristic picks SPB = 1, 256-VGPR workgroups, for launches of <= 512 workgroups)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04s; mkdir -p $O
for w in standard config3 config5 feather; do for spb in 0 1 2; do for i in 1 2; do
  f=$O/bench_${w}_spb${spb}_$i.json
  if [ $spb = 0 ]; then timeout 300 python bench.py --no-cpu-baseline --workload $w > $f 2> ${f%.json}.err; else NA_SP_SPB=$spb timeout 300 python bench.py --no-cpu-baseline --workload $w > $f 2> ${f%.json}.err; fi
  python -c "
import json;j=json.loads(open('$f').read().strip().splitlines()[-1]);print('$w spb $spb',round(j['ms_per_step']*1e3,2),'launches',j['launches_per_step'],'frac',round(j['roofline']['frac'],4),'parity',j['parity_rms'])"
done; done; done
