#!/bin/bash
# round 4: why was the one-launch step 46 us in r04n?  control (quick_time, torch events only) vs bench both ways on one box
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04o; mkdir -p $O
for i in 1 2; do
  python tools/quick_time.py 16 8 1024 | tee -a $O/quick.txt
  timeout 300 python bench.py --no-cpu-baseline --caller-stream > $O/bench_caller_$i.json 2> $O/bench_caller_$i.err
  timeout 300 python bench.py --no-cpu-baseline > $O/bench_own_$i.json 2> $O/bench_own_$i.err
  NA_HOST_HALVES=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_ownsingle_$i.json 2> $O/bench_ownsingle_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04o/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'ms_per_step', round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'launches', j['launches_per_step'], 'frac', round(j['roofline']['frac'],4), 'iso', round(j['kernel_ms_median_isolated']*1e3,2), 'ramp', j['clock_ramp']['untimed_steps'])
    except Exception as e: print(f, 'ERR', e)
PY
