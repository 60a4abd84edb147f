# A/B on one box: the layer-0 wave publishes its progress after the first step of a group (this tree) or at its end (KEXTRA=-DNA_PIPE_PUBLISH_LATE=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
run() {
  python tools/runs/r06d_lstm.py 2>&1 | grep -E "^(256|512|1024) streams" | sed "s/^/$1: lstm2x16 /" | tee -a $O/ab.txt
  for i in 1 2 3; do python bench.py --workload config4 --no-cpu-baseline --no-host-path --steps 1000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: config4 us per step %.2f frac %.4f' % (d['ms_per_step']*1000, d['roofline']['frac']))"; done | tee -a $O/ab.txt
}
run early
rm -f neuralaudio_amd/csrc/build/recurrent_dpp_kernels.o; ( cd neuralaudio_amd/csrc && make -j8 KEXTRA=-DNA_PIPE_PUBLISH_LATE=1 2>&1 | grep -E "error" )
run late
rm -f neuralaudio_amd/csrc/build/recurrent_dpp_kernels.o; ( cd neuralaudio_amd/csrc && make -j8 2>&1 | grep -E "error" )
run early-again
