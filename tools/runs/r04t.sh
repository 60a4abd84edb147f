#!/bin/bash
# round 4: full GPU suite, smoke, the judged profile set r04_p2 (final build of the half-batch chains)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04t
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r04t/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04t/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04t/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r04t/smoke.log
bash tools/runs/r04_profiles.sh r04_p2 > gpurun_out/r04t/profiles.log 2>&1
python - <<'PY'
import json
for d in ("r04_p2","r04_p2_cfg3","r04_p2_cfg5","r04_p2_cfg4","r04_p2_nano1024","r04_p2_feather1024"):
    j=json.loads(open('gpurun_out/%s/bench.json'%d).read().strip().splitlines()[-1]); r=j['roofline']
    print(d, round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'launches', j['launches_per_step'], 'frac', round(r['frac'],4), 'cpu', (j.get('cpu_baseline') or {}).get('value'), 'parity', j['parity_rms'])
    print(open('gpurun_out/%s/kernel_stats.csv'%d).read().splitlines()[3:6])
PY
for i in 1 2 3; do neuralaudio_amd/HostPipeBench tests/golden/models/BossWN-standard.nam 1024 128 3000 | tee -a gpurun_out/r04t/hostpipe.txt; done
