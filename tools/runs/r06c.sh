mkdir -p gpurun_out/r06c
( time timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_latency.py -x -q -m gpu --durations=10 -s ) > gpurun_out/r06c/scale.log 2>&1; echo "rc=$?" >> gpurun_out/r06c/scale.log
( time timeout 1000 python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r06c/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06c/pytest.log
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06c/bench.log 2>&1
tail -n 30 gpurun_out/r06c/scale.log; tail -n 6 gpurun_out/r06c/pytest.log; tail -n 3 gpurun_out/r06c/bench.log | cut -c1-300
