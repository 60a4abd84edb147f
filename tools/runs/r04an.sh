#!/bin/bash
# round 4: packed prologue with its input loads issued together -- timeline of Nano x 1024, parity, then the narrow workloads
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04an; mkdir -p $O
NA_LIB_SUFFIX=_trace NA_TRACE_MODEL=BossWN-nano.nam NA_TRACE_STREAMS=1024 NA_TRACE_WAVES=8 NA_TRACE_STAGES=23 timeout 200 python tools/trace_split_timeline.py < /dev/null 2>&1 | grep -E "kernel entry|^ 0 |^ 1 " | tee $O/timeline_nano.txt
timeout 1200 python -m pytest tests/test_gpu_spec.py tests/test_gpu_batch.py -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1
for w in nano feather config3; do for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-host-path --workload $w < /dev/null > $O/b.json 2>/dev/null; python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$w', round(j['ms_per_step']*1e3,2),'frac',round(j['roofline']['frac'],4),'parity',j['parity_rms'])"
done; done 2>&1 | tee $O/ab.txt
