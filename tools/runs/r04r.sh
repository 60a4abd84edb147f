#!/bin/bash
# round 4: full GPU suite on the build with free-running half-batch chains, then the judged profile set r04_p2
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04r
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04r/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04r/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04r/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r04r/smoke.log
bash tools/runs/r04_profiles.sh r04_p2 2>&1 | tail -60
