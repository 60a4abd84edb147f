# config 4: the GRU streams of the pipelined launch four per workgroup (default) or two per workgroup on waves {0,3} / {1,2} (NA_REC_MIX=1|2, experiment)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ak; mkdir -p $O
for m in 0; do for i in 1 2; do NA_REC_MIX=$m python bench.py --workload config4 --no-cpu-baseline --no-host-path --steps 1500 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NA_REC_MIX=$m config4 us per step %.2f parity %s' % (d['ms_per_step']*1000, d.get('parity_rms')))"; done; done | tee $O/mix.txt
