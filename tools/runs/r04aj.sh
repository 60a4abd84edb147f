#!/bin/bash
# round 4: array 0's rechannel in front of the prologue barrier (one barrier stage less) -- bit identity / parity, then A/B per workload
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04aj; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_spec.py tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2
for w in standard config3 config5 nano feather lite; do for suf in "" _nofuse; do for i in 1 2; do
  NA_LIB_SUFFIX=$suf timeout 300 python bench.py --no-cpu-baseline --no-host-path --workload $w < /dev/null > $O/b.json 2>/dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$w', '${suf:-fused}', round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4),'parity',j['parity_rms'])"
done; done; done 2>&1 | tee $O/ab.txt
