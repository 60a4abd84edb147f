cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06s
for i in 1 2; do ( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=6 ) > gpurun_out/r06s/pytest_$i.log 2>&1; echo "rc=$?" >> gpurun_out/r06s/pytest_$i.log; grep -E "passed|failed|rc=|real" gpurun_out/r06s/pytest_$i.log; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
