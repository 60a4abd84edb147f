#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
timeout 600 python tools/split_launch_probe2.py 2>&1 | tee $O/phase.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "keras" 2>&1 | tail -5
