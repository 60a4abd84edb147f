# full -m gpu suite on a fresh lease (cold or warm), then the three soak cases that failed in r06f, then the round's profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r06g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06g/pytest.log
( time timeout 900 python -m pytest tests/test_gpu_families.py -q -m gpu_soak -k "NA_WN_SPEC or NA_REC_QUAD_MIN or (NA_WN_KERNEL and split)" --durations=8 ) > gpurun_out/r06g/soak3.log 2>&1; echo "rc=$?" >> gpurun_out/r06g/soak3.log
tail -n 8 gpurun_out/r06g/pytest.log; tail -n 12 gpurun_out/r06g/soak3.log
bash tools/runs/r06_profiles.sh r06_p1 2>&1 | tail -60
