#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04at; mkdir -p $O
for i in 1 2 3 4; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps: wall', round(j['ms_per_step']*1e3,2), 'events', round(j['kernel_ms_avg']*1e3,2), 'value', round(j['value'],1), 'frac', round(j['roofline']['frac'],4), 'wall frac', round(j['roofline']['frac_wall_clock'],4), j['parity_rms'])"; done
timeout 3000 python -m pytest tests -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1
