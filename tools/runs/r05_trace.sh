#!/bin/bash
# round 5: per-stage timeline of one workgroup inside the resident launch vs inside the free-running chains (trace build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05e}; mkdir -p $O
export NA_LIB_SUFFIX=_trace NA_TRACE_OWN=1 NA_TRACE_STEPS=40
for wg in 0 200; do
  NA_TRACE_BLOCK=$wg NA_RESIDENT=1 timeout 200 python tools/trace_split_timeline.py > $O/timeline_resident_wg$wg.txt 2>&1
  NA_TRACE_BLOCK=$wg NA_RESIDENT=0 timeout 200 python tools/trace_split_timeline.py > $O/timeline_chains_wg$wg.txt 2>&1
done
NA_TRACE_BLOCK=100 NA_RESIDENT=1 NA_RESIDENT_GRID=256 NA_TRACE_STREAMS=512 timeout 200 python tools/trace_split_timeline.py > $O/timeline_resident_1percu_wg100.txt 2>&1
head -40 $O/timeline_resident_wg200.txt; head -40 $O/timeline_chains_wg200.txt;  head -5 $O/timeline_resident_1percu_wg100.txt
