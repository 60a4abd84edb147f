#!/bin/bash
# round 5: the non-temporal variant for the lite and A2 families (config 3: 453 MB of state, config 5: 943 MB) against NA_WN_NT=0, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05p}; mkdir -p $O
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["launch_mode"], round(d["ms_per_step"]*1e3,2), round(d["roofline"]["frac"],3))'
for rep in 1 2 3; do
for w in config3 config5; do
  timeout 300 python bench.py --workload $w --steps 500 --no-cpu-baseline --no-parity-check --no-host-path 2>/dev/null | python -c "$P" "$w nt" >> $O/nt.txt
  NA_WN_NT=0 timeout 300 python bench.py --workload $w --steps 500 --no-cpu-baseline --no-parity-check --no-host-path 2>/dev/null | python -c "$P" "$w off" >> $O/nt.txt
done; done
cat $O/nt.txt
