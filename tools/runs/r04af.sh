#!/bin/bash
# round 4: full GPU suite + smoke + default bench at HEAD
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04af; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 600 python bench.py < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench20.json 2> $O/bench20.err
python - <<'PY'
import json
for f in ("bench","bench20"):
    j=json.loads(open('/root/repo/gpurun_out/r04af/%s.json'%f).read().strip().splitlines()[-1]); r=j['roofline']
    print(f, 'value', round(j['value'],1), 'ms_per_step', round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'launches', j['launches_per_step'], 'frac', round(r['frac'],4), 'wall', round(r['frac_wall_clock'],4), 'meas', r['frac_measured_bytes'], 'cpu', (j.get('cpu_baseline') or {}).get('value'), 'parity', j['parity_rms'], 'host', (j.get('pcie_inclusive') or {}).get('ms_per_buffer'))
PY
