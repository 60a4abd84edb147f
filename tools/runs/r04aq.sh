#!/bin/bash
# round 4: every bench workload at HEAD, on the batch's own streams and on a caller's stream (one box, 500 steps each)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04aq; mkdir -p $O
printf "%-10s %-14s %10s %9s %7s %s\n" workload streams us/step launches frac kernel | tee $O/table.txt
for w in standard lite feather nano a2full config3 config4 config5 lstm1x16 lstm2x8; do for mode in "" "--caller-stream"; do
  timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-parity-check --steps 500 --workload $w $mode < /dev/null > $O/b.json 2>/dev/null
  python -c "
import json;j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);r=j['roofline']
print('%-10s %-14s %10.2f %9d %7.4f %s' % ('$w', '${mode:-own-streams}', j['ms_per_step']*1e3, j['launches_per_step'], r['frac'], r['kernel']))" | tee -a $O/table.txt
done; done
