#!/bin/bash
# round 5: batches just beyond the Infinity Cache (the north star's 1250 streams per GPU = 304 MB): which layers' ring traffic to mark
# non-temporal so that the rest fits the cache?  QUICK builds with NT from d >= 256 / 512, the product (d >= 128), and no NT
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05q}; mkdir -p $O
for rep in 1 2 3; do
for S in 1250 1536; do
  echo -n "off     " >> $O/band.txt; K=200 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_WN_NT=0 2>/dev/null >> $O/band.txt
  echo -n "d>=128  " >> $O/band.txt; K=200 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_WN_NT_MB=200 2>/dev/null >> $O/band.txt
  echo -n "d>=256  " >> $O/band.txt; K=200 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_WN_NT_MB=200 NA_LIB_SUFFIX=_ntd256 2>/dev/null >> $O/band.txt
  echo -n "d>=512  " >> $O/band.txt; K=200 timeout 200 python tools/quick_time_own.py BossWN-standard.nam $S NA_WN_NT_MB=200 NA_LIB_SUFFIX=_ntd512 2>/dev/null >> $O/band.txt
done; done
cat $O/band.txt
