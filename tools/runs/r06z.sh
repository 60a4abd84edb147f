# the full GPU suite and the recurrent soak cases at the tree with lstm_kernels.hip / gru_kernels.hip built without the SLP vectoriser
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=|real" $O/pytest_gpu.log
( time timeout 1500 python -m pytest tests -q -m gpu_soak -k "LSTM or REC or GRU" --durations=10 ) > $O/pytest_soak_rec.log 2>&1; echo "rc=$?" >> $O/pytest_soak_rec.log; grep -E "passed|failed|rc=|real" $O/pytest_soak_rec.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 500 --warmup 100 2>/dev/null | cut -c1-400
