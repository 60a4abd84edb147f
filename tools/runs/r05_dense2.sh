#!/bin/bash
# round 5: dense packs at small stream counts (the 16 / 16 layout has the one-tile-per-wave chain there)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05j}; mkdir -p $O
for rep in 1 2; do
for S in 16 64 256 512 768 1024 1366 2048; do
for d in 0 1; do
    echo -n "dense=$d " >> $O/time.txt; K=300 timeout 200 python tools/quick_time_own.py BossWN-nano.nam $S NA_WN_DENSE=$d 2>/dev/null >> $O/time.txt
done; done; done
cat $O/time.txt
