#!/bin/bash
# round 4: the judged profile set r04_p2 on the final build (profiling runs without the host-path section of the bench)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04y
bash tools/runs/r04_profiles.sh r04_p2 > gpurun_out/r04y/profiles.log 2>&1
python - <<'PY'
import json
for d in ("r04_p2","r04_p2_cfg3","r04_p2_cfg5","r04_p2_cfg4","r04_p2_nano1024","r04_p2_feather1024"):
    j=json.loads(open('gpurun_out/%s/bench.json'%d).read().strip().splitlines()[-1]); r=j['roofline']
    print(d, round(j['ms_per_step']*1e3,2), 'kernel', round(j['kernel_ms_avg']*1e3,2), 'launches', j['launches_per_step'], 'frac', round(r['frac'],4), 'cpu', (j.get('cpu_baseline') or {}).get('value'), 'parity', j['parity_rms'], 'host', (j.get('pcie_inclusive') or {}).get('ms_per_buffer'), (j.get('host_buffer_latency_ms') or {}).get('p50'))
    print(open('gpurun_out/%s/kernel_stats.csv'%d).read().splitlines()[3:6])
    print([l for l in open('gpurun_out/%s/pmc_summary.txt'%d).read().splitlines() if l.split()[0] in ('FETCH_SIZE','WRITE_SIZE','SQ_WAVES')])
PY
