#!/bin/bash
# round 5: does the resident launch do better with fewer workgroups than slots?  streams = 2 x grid x 8 so that every workgroup runs 8 blocks per command
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05d}; mkdir -p $O
for g in 512 480 448 416 384 320 256; do
  S=$((g*16))
  K=100 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_RESIDENT=1 NA_RESIDENT_GRID=$g >> $O/grid.txt 2>/dev/null
  K=100 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_RESIDENT=0 >> $O/grid.txt 2>/dev/null
done
cat $O/grid.txt
