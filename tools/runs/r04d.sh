#!/bin/bash
# round 4, fourth GPU call: the whole suite on exact rings (first layers of an array excluded) + RCCL one-rank test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log
