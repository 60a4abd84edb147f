# after the allowQuad fix: the new recurrent tests, the quad-forced soak case, the race hunt script (mixed batch now on the one-stream kernel), the full suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06r
( timeout 300 python -m pytest tests/test_gpu_recurrent_quad.py -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r06r/quad.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_families.py -q -m gpu_soak -k "NA_REC_QUAD_MIN or NA_REC_NOPIPE or NA_BATCH_NO_GRAPH" 2>&1 | tail -6 ) > gpurun_out/r06r/soak3.log 2>&1
( for i in 1 2; do python tools/runs/r06n_quadrace.py std 64 400 128 2>&1 | grep "^mix" | cut -c1-200; done ) > gpurun_out/r06r/race.log 2>&1
( time timeout 900 python -m pytest tests -x -q -m gpu --durations=8 ) > gpurun_out/r06r/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06r/pytest.log
cat gpurun_out/r06r/quad.log gpurun_out/r06r/soak3.log gpurun_out/r06r/race.log; tail -n 18 gpurun_out/r06r/pytest.log
