#!/bin/bash
# round 4: where the two chains sit in time (rocprofv3 kernel trace, csv) -- Standard, config 3, config 5
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ad; mkdir -p $O
for w in standard config3 config5; do
  rm -rf /tmp/ct
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/ct -o ct -- python /root/repo/bench.py --steps 400 --no-cpu-baseline --no-parity-check --no-host-path --workload $w > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null )
  echo "== $w"; timeout 120 python tools/chain_phase.py /tmp/ct 200 < /dev/null | sed "s/^/  /"
  F=$(find /tmp/ct -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$F" ] && head -3 "$F" | cut -c1-400 > $O/csv_head_$w.txt
done 2>&1 | tee $O/chain_phase.txt
