#!/bin/bash
# round 5: shader clock and power while the headline step runs back to back
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05f}; mkdir -p $O
rocm-smi --showclocks --showpower > $O/idle.txt 2>&1
(NA_RESIDENT=0 python bench.py --no-cpu-baseline --no-host-path --no-exact-f32 --no-parity-check --rotate 0 --steps 400000 > $O/bench_long.json 2>/dev/null) &
BP=$!
sleep 9
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temp" >> $O/busy.txt; echo -- >> $O/busy.txt; sleep 1; done
wait $BP
cat $O/busy.txt | head -40; python -c "import json; d=json.load(open('$O/bench_long.json')); print(d['ms_per_step'])"
