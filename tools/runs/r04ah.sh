#!/bin/bash
# round 4: per-stage timeline of one workgroup, one ordered launch vs two free-running chains (trace build)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ah; mkdir -p $O
for blk in 0 200; do
  NA_LIB_SUFFIX=_trace NA_TRACE_BLOCK=$blk NA_TRACE_STEPS=40 timeout 200 python tools/trace_split_timeline.py < /dev/null > $O/timeline_ordered_wg$blk.txt 2>&1
  NA_LIB_SUFFIX=_trace NA_TRACE_BLOCK=$blk NA_TRACE_STEPS=40 NA_TRACE_OWN=1 timeout 200 python tools/trace_split_timeline.py < /dev/null > $O/timeline_chains_wg$blk.txt 2>&1
done
paste <(grep -E "^ ?[0-9]+ " $O/timeline_ordered_wg0.txt | awk '{print $1, $2, $NF}') <(grep -E "^ ?[0-9]+ " $O/timeline_chains_wg0.txt | awk '{print $2, $NF}') | head -30
grep "kernel entry" $O/*.txt
