# after the pair-layout fix of the four-streams-per-wave kernel: the hardware probe's numbers on file, the victims beside the aggressor,
# the full GPU suite, the soak matrix, the kernel's step times, smoke and the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/pk_lds_opsel tools/microbench/pk_lds_opsel.hip 2>&1 | tail -3
timeout 300 /tmp/pk_lds_opsel 6 burners > $O/pk_lds_opsel.txt 2>&1; tail -14 $O/pk_lds_opsel.txt
timeout 600 python tools/runs/r06u_victims.py 2>&1 | grep -v amdgpu.ids > $O/victims.txt; tail -16 $O/victims.txt
( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=|real" $O/pytest_gpu.log
( time timeout 2400 python -m pytest tests -q -m gpu_soak --durations=25 ) > $O/pytest_soak.log 2>&1; echo "rc=$?" >> $O/pytest_soak.log; grep -E "passed|failed|rc=|real" $O/pytest_soak.log
python tools/runs/r06q_quadperf.py 2>&1 | grep streams | tee $O/quadperf.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
