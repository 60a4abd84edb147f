# the whole soak matrix (21 forced runs x 250 tests) behind a warm start, then the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
( time timeout 1700 python -m pytest tests/test_gpu_families.py -q -m gpu_soak --durations=25 ) > gpurun_out/r06k/soak.log 2>&1; echo "rc=$?" >> gpurun_out/r06k/soak.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06k/bench20.log 2>&1
tail -n 32 gpurun_out/r06k/soak.log; tail -n 4 gpurun_out/r06k/bench20.log | cut -c1-400
