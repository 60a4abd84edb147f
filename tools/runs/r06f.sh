mkdir -p gpurun_out/r06f
( time timeout 1500 python -m pytest tests/test_gpu_families.py -q -m gpu_soak --durations=25 ) > gpurun_out/r06f/soak.log 2>&1; echo "rc=$?" >> gpurun_out/r06f/soak.log
tail -n 40 gpurun_out/r06f/soak.log
