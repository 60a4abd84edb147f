#!/bin/bash
# round 4: wide recurrent layers -- gate rows per lane (NA_REC_RPL) x unroll of the L2 weight stream (library variants)
cd /root/repo; O=gpurun_out/r04aa; mkdir -p $O
for m in lstm:1:256 lstm:2:256 lstm:1:128 gru:1:256 lstm:1:512; do
  for suf in "" _u4 _u8; do for rpl in 4 2 1; do
    echo -n "$m unroll${suf:-_u2} rpl $rpl: "; NA_LIB_SUFFIX=$suf NA_REC_RPL=$rpl python tools/quick_time_recurrent.py $m 64 2>&1 | grep " x " | sed 's/ | four.*//; s/.*per wave //'
  done; done
done | tee $O/times.txt
