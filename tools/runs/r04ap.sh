#!/bin/bash
# round 4: stage 1's weight DMA issued at kernel start instead of at the start of stage 0 (_late = before)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ap; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1
for w in standard nano config5 lite; do for rep in 1 2 3; do for suf in "" _late; do
  NA_LIB_SUFFIX=$suf timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-parity-check --workload $w < /dev/null > $O/b.json 2>/dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$w', '${suf:-early}', round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4))"
done; done; done 2>&1 | tee $O/ab.txt
