#!/bin/bash
# round 4: 2 / 3 / 4 free-running chains, per workload
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "halves or half" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
NA_HOST_CHAINS=3 timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "halves or half" > $O/pytest3.log 2>&1; echo "pytest(3 chains) rc $?"; tail -3 $O/pytest3.log
for w in standard config3 config5 nano feather; do for c in 2 3 4; do for i in 1 2; do
  f=$O/bench_${w}_c${c}_$i.json
  NA_HOST_CHAINS=$c timeout 300 python bench.py --no-cpu-baseline --workload $w > $f 2> ${f%.json}.err
  python -c "
import json;j=json.loads(open('$f').read().strip().splitlines()[-1]);print('$w chains $c',round(j['ms_per_step']*1e3,2),'launches',j['launches_per_step'],'frac',round(j['roofline']['frac'],4),'parity',j['parity_rms'])"
done; done; done
