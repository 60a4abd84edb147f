# recurrent files without the SLP vectoriser (this tree) against with it (rebuilt on the box): step times; then the runtime-kernel victims
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
echo "== REC_SLP=-fno-slp-vectorize (this tree) ==" | tee $O/rtperf.txt
python tools/runs/r06y_rtperf.py 2>&1 | grep "us per step" | tee -a $O/rtperf.txt
NA_LSTM_LANE_KERNEL=1 python tools/runs/r06y_rtperf.py 2>&1 | grep "us per step" | sed 's/^/lane kernels: /' | tee -a $O/rtperf.txt
timeout 600 python tools/runs/r06u_victims.py "(rt" 2>&1 | grep differing | tee $O/victims_rt.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
rm -f neuralaudio_amd/csrc/build/lstm_kernels.o neuralaudio_amd/csrc/build/gru_kernels.o
( cd neuralaudio_amd/csrc && make -j8 REC_SLP= 2>&1 | grep -E "error" )
echo "== REC_SLP= (compiler pairs scalar FMAs: the build up to r06w) ==" | tee -a $O/rtperf.txt
python tools/runs/r06y_rtperf.py 2>&1 | grep "us per step" | tee -a $O/rtperf.txt
NA_LSTM_LANE_KERNEL=1 python tools/runs/r06y_rtperf.py 2>&1 | grep "us per step" | sed 's/^/lane kernels: /' | tee -a $O/rtperf.txt
