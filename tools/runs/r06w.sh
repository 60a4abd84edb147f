# the full GPU suite at the tree with the pair-layout four-streams kernel (r06v ran the soak matrix, the probes and the bench; its -m gpu
# run stopped at a wrong assertion of the new test)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
for i in 1 2; do ( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/pytest_gpu_$i.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_$i.log; grep -E "passed|failed|rc=|real" $O/pytest_gpu_$i.log; done
NA_REC_QUAD_MIN=1 timeout 600 python -m pytest tests/test_gpu_recurrent_quad.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
