cd $GRAFT_REPO_ROOT
echo "== NA_REC_QUAD_MIN=1"; NA_REC_QUAD_MIN=1 python tools/runs/r06l_quad.py 2>&1 | grep -v amdgpu.ids
echo "== default"; python tools/runs/r06l_quad.py 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_recurrent_quad.py -x -q -m gpu 2>&1 | tail -3
