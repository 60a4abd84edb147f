#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04ai; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -A14 "slowest" $O/pytest.log | head -16
