#!/bin/bash
# round 4: rocprofv3 kernel statistics of the kernels behind the lifted size limits (WaveNetWideKernel, RecurrentWaveRtKernel with a workgroup per stream)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04am; mkdir -p $O
run() { # tag, command...
  tag=$1; shift
  rm -rf /tmp/st_$tag
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$tag -o prof -- "$@" > $O/$tag.log 2>&1 < /dev/null )
  DB=$(find /tmp/st_$tag -name "*.db" 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" $O/${tag}_kernel_stats.csv "$* under rocprofv3 --kernel-trace --stats" 2>> $O/$tag.log
  grep " x \|us/step" $O/$tag.log | tail -2; head -5 $O/${tag}_kernel_stats.csv
}
run wide128 python /root/repo/tools/quick_time.py 128 64 256
run wide96 python /root/repo/tools/quick_time.py 96 48 256
run lstm1x256 python /root/repo/tools/quick_time_recurrent.py lstm:1:256 64
run lstm1x128 python /root/repo/tools/quick_time_recurrent.py lstm:1:128 64
run gru1x256 python /root/repo/tools/quick_time_recurrent.py gru:1:256 64
