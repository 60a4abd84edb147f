#!/bin/bash
# round 4: free-running half-batch launches behind NA_BatchProcessDevice (own streams) -- tests + headline both ways + host pipeline
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline > $O/bench_own_$i.json 2> $O/bench_own_$i.err
  timeout 300 python bench.py --no-cpu-baseline --caller-stream > $O/bench_caller_$i.json 2> $O/bench_caller_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04n/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'ms_per_step', round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'launches', j['launches_per_step'], 'frac', round(j['roofline']['frac'],4), 'iso', round(j['kernel_ms_median_isolated']*1e3,2), 'parity', j['parity_rms'])
    except Exception as e: print(f, 'ERR', e)
PY
for w in config3 config5; do timeout 300 python bench.py --no-cpu-baseline --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json;j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]);print('$w',round(j['ms_per_step']*1e3,2),j['launches_per_step'],round(j['roofline']['frac'],4),j['parity_rms'])"; done
