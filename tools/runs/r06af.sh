# rocprofv3 kernel stats of the recurrent kernels at scale, final tree: the four-streams-per-wave kernel (LSTM / GRU 1x16 x 3072 .. 8192, tools/runs/r06q_quadperf.py)
# and the two-layer pipeline (LSTM 2x16 x 256 .. 2560, tools/runs/r06d_lstm.py)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06af; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for s in r06q_quadperf r06d_lstm; do
  rm -rf $O/stats
  rocprofv3 --kernel-trace --stats -d $O/stats -o prof -- python $R/tools/runs/$s.py > $O/$s.log 2>&1
  DB=$(find $O/stats -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $O/${s}_kernel_stats.csv "python tools/runs/$s.py under rocprofv3 --kernel-trace --stats (all batch sizes of the script in one average)" 2>> $O/$s.log
  grep streams $O/$s.log; head -6 $O/${s}_kernel_stats.csv
done
rm -rf $O/stats
