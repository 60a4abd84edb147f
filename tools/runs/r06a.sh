mkdir -p gpurun_out/r06a
( time timeout 1000 python -m pytest tests -x -q -m gpu --durations=40 ) > gpurun_out/r06a/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06a/pytest.log
for i in 1 2 3; do ( time NA_LSTM_LANE_KERNEL=1 timeout 500 python -m pytest -q -x -m gpu --durations=15 tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py ) > gpurun_out/r06a/lane_$i.log 2>&1; echo "rc=$?" >> gpurun_out/r06a/lane_$i.log; done
tail -5 gpurun_out/r06a/pytest.log; tail -4 gpurun_out/r06a/lane_*.log
