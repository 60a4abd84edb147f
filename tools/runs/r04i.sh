#!/bin/bash
# round 4: cache-policy bits on the ring traffic of the long dilations (nt on loads / stores / both, from d = 256 or 128; sc1); per-stage timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
NA_AB_ARGS="--no-parity-check" timeout 1500 bash tools/ab_bench.sh "_quick _ntl _nts _ntb _ntb128 _sc1b" 1000 2>&1 | tee $O/ab_nt.txt
NA_LIB_SUFFIX=_trace NA_TRACE_BLOCK=0 python tools/trace_split_timeline.py 2>&1 | tee $O/timeline_wg0.txt
NA_LIB_SUFFIX=_trace NA_TRACE_BLOCK=300 python tools/trace_split_timeline.py 2>&1 | tee $O/timeline_wg300.txt
