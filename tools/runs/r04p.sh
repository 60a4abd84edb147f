#!/bin/bash
# round 4: half-batch launches for fused / packed batches -- tests, then every workload both ways
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for w in standard config3 config5 nano feather; do for mode in "" "--caller-stream"; do for i in 1 2; do
  f=$O/bench_${w}_${mode#--}_$i.json
  timeout 300 python bench.py --no-cpu-baseline --workload $w $mode > $f 2> ${f%.json}.err
  python -c "
import json;j=json.loads(open('$f').read().strip().splitlines()[-1]);print('$w','${mode#--}' or 'own',round(j['ms_per_step']*1e3,2),'launches',j['launches_per_step'],'frac',round(j['roofline']['frac'],4),'parity',j['parity_rms'])"
done; done; done
