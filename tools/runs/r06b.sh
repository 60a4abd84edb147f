mkdir -p gpurun_out/r06b
( time timeout 300 python -m pytest tests/test_gpu_stall.py -x -q -m gpu --durations=10 ) > gpurun_out/r06b/stall.log 2>&1; echo "rc=$?" >> gpurun_out/r06b/stall.log
( time timeout 1000 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r06b/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06b/pytest.log
( time python bench.py ) > gpurun_out/r06b/bench.log 2>&1
tail -n 25 gpurun_out/r06b/stall.log; tail -n 6 gpurun_out/r06b/pytest.log; tail -n 3 gpurun_out/r06b/bench.log
