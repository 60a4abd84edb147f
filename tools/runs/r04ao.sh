#!/bin/bash
# round 4: instruction cache under the chains (two workgroups of a CU in different phases of a straight-line 50 KB kernel) vs ordered launches
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ao; mkdir -p $O
for mode in "" "--caller-stream"; do
  tag=${mode:+ordered}; tag=${tag:-chains}
  rm -rf /tmp/ic_$tag
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d /tmp/ic_$tag -o pmc -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-host-path $mode > $O/$tag.log 2>&1 < /dev/null )
  mkdir -p $O/$tag; cp -r /tmp/ic_$tag/* $O/$tag/ 2>/dev/null
  echo "== $tag"; python tools/pmc_summary.py /tmp/ic_$tag WaveNetSpecKernel 2>&1 | tail -8
done
