#!/bin/bash
# round 5: what each part of the headline chain costs at the socket power cap -- step time, shader clock and power of ablation builds
# (QUICK builds: the Standard chain only; NA_ABL masks: wavenet_split_dev.h).  At the cap the step time is the energy per step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05k}; mkdir -p $O
for rep in 1 2; do
for m in 0 1 2 4 16 32 64 256; do
  echo -n "abl=$m  " >> $O/energy.txt
  NA_LIB_SUFFIX=_abl$m python tools/power_sample.py 3 >> $O/energy.txt 2>&1
done; done
cat $O/energy.txt
