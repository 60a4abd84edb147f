# the whole soak matrix at the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ae; mkdir -p $O
( time timeout 2700 python -m pytest tests -q -m gpu_soak --durations=25 ) > $O/pytest_soak.log 2>&1; echo "rc=$?" >> $O/pytest_soak.log; grep -E "passed|failed|rc=|real" $O/pytest_soak.log
