cd $GRAFT_REPO_ROOT
python tools/runs/r06u_victims.py quad 2>&1 | grep -v amdgpu.ids
python tools/runs/r06p_quadrace2.py 64 own 2 2>&1 | grep "^separate" | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_recurrent_quad.py -x -q -m gpu 2>&1 | tail -3
python tools/runs/r06q_quadperf.py 2>&1 | grep streams
