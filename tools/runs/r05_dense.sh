#!/bin/bash
# round 5: dense packs (four Nano streams at 16 / 8 virtual channels): parity, then timing against the 16 / 16 layout
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05i}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spec.py -m gpu -x -q -k "packed or config3 or pack_factor or mixed" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2; do
for d in 0 1; do
  for S in 1024 4096 8192; do
    echo -n "dense=$d " >> $O/time.txt; K=200 timeout 200 python tools/quick_time_own.py BossWN-nano.nam $S NA_WN_DENSE=$d 2>/dev/null >> $O/time.txt
  done
  NA_WN_DENSE=$d timeout 300 python bench.py --workload config3 --steps 500 --no-cpu-baseline --no-parity-check --no-host-path 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense=$d config3', d['launch_mode'], round(d['ms_per_step']*1e3,2), round(d['roofline']['frac'],3))" >> $O/time.txt
done; done
cat $O/time.txt
