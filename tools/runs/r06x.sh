# the op_sel hazard: other delivery paths (ds_read_b32 / b64, global_load) in the probe, and the runtime-shaped recurrent kernel -- whose
# compiler-made packed FMAs take ds_read_b32 / global_load results with op_sel (tools/analysis/pk_opsel_sources.py) -- beside the aggressor
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/pk_lds_opsel tools/microbench/pk_lds_opsel.hip 2>&1 | grep -v "occupancy\|warnings gen" | tail -3
timeout 400 /tmp/pk_lds_opsel 6 burners > $O/pk_lds_opsel.txt 2>&1; grep "victim" $O/pk_lds_opsel.txt | cut -c1-170
timeout 900 python tools/runs/r06u_victims.py "(rt" 2>&1 | grep -v amdgpu.ids | tee $O/victims_rt.txt
