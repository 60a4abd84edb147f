#!/bin/bash
# round 5: the product's non-temporal variant for batches beyond the Infinity Cache (Cfg::NT) against NA_WN_NT=0, one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05o}; mkdir -p $O
for rep in 1 2 3; do
for S in 8192 2048 1280; do
  K=100 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S 2>/dev/null >> $O/nt.txt
  K=100 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_WN_NT=0 2>/dev/null >> $O/nt.txt
done; done
cat $O/nt.txt
