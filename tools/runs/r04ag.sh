#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ag; mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench20_$i.json 2> $O/bench20_$i.err; done
timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r04ag/bench*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j['roofline']
    print(f.split('/')[-1], 'steps', j['steps'], 'value', round(j['value'],1), 'ms_per_step', round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'frac', round(r['frac'],4), 'wall', round(r['frac_wall_clock'],4), 'parity', j['parity_rms'])
PY
NA_LSTM_NO_WAVE_RT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "keras or stack" < /dev/null 2>&1 | tail -2
