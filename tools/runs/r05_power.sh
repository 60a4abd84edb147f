#!/bin/bash
# round 5: which workloads hold the 1400 W socket power cap?
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05g}; mkdir -p $O
{
python tools/power_sample.py 4
python tools/power_sample.py 4 -- --resident
python tools/power_sample.py 4 -- --caller-stream
python tools/power_sample.py 4 -- --streams 512
python tools/power_sample.py 4 -- --streams 256
NA_WN_KERNEL=frame NA_PS_STEPS_PER_S=15000 python tools/power_sample.py 4
NA_PS_STEPS_PER_S=11000 python tools/power_sample.py 4 -- --workload config3
NA_PS_STEPS_PER_S=25000 python tools/power_sample.py 4 -- --workload config4
NA_PS_STEPS_PER_S=12000 python tools/power_sample.py 4 -- --workload config5
NA_PS_STEPS_PER_S=45000 python tools/power_sample.py 4 -- --workload nano
NA_PS_STEPS_PER_S=45000 python tools/power_sample.py 4 -- --workload feather
} > $O/power.txt 2>&1
cat $O/power.txt
