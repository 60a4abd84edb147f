#!/bin/bash
# round 5: the rotation regime (8 x 1024 Standard streams, 2 GB of state walked once per step) -- resident launch vs free-running chains vs ordered
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05c}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-path --no-exact-f32 --no-parity-check --steps 500"
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["rotation"]; print(sys.argv[1], d["launch_mode"], round(d["ms_per_step"]*1e3,2), "| rotation", r.get("launch_mode"), round(r["us_per_1024_step"],2), round(r["frac"],3))'
for rep in 1 2; do
NA_RESIDENT=1 timeout 300 $B 2>/dev/null | python -c "$P" resident >> $O/rot.txt
timeout 300 $B 2>/dev/null | python -c "$P" chains >> $O/rot.txt
NA_HOST_HALVES=0 timeout 300 $B 2>/dev/null | python -c "$P" ordered >> $O/rot.txt
done
cat $O/rot.txt
