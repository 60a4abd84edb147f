#!/bin/bash
# round 5: cache-policy bits on the ring traffic OUTSIDE the Infinity Cache (8192 streams = 2 GB of state per step) and inside it (1024)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05h}; mkdir -p $O
for rep in 1 2; do
for v in _q _qnts _qntl _qntb _qsc1s _qnts64; do
  echo -n "$v  " >> $O/nt.txt; NA_LIB_SUFFIX=$v K=100 timeout 200 python tools/quick_time_own.py BossWN-standard.nam 8192 2>/dev/null >> $O/nt.txt
  echo -n "$v  " >> $O/nt.txt; NA_LIB_SUFFIX=$v K=400 timeout 200 python tools/quick_time_own.py BossWN-standard.nam 1024 2>/dev/null >> $O/nt.txt
done; done
cat $O/nt.txt
bash tools/runs/r05_power.sh $1
