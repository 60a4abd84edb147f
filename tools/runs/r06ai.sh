# Nano / Feather x 1024: the existing layout knobs once more at the final tree (bench.py --workload nano|feather, us per step from the event marks)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ai; mkdir -p $O
one() { # label, env..., workload
  local label="$1"; shift; local wl="$1"; shift
  env "$@" python bench.py --workload $wl --no-cpu-baseline --no-host-path --no-parity-check --rotate 0 --no-exact-f32 --steps 1500 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s %-34s %.2f us per step  launches %s  pack %s  kernel %s' % ('$wl', '$label', d['ms_per_step']*1000, r.get('launches_per_step'), r.get('stream_pack_factor'), r.get('kernel')))" | tee -a $O/sweep.txt
}
for wl in nano feather; do
  one default $wl X=1
  one default-again $wl X=1
  one NA_HOST_HALVES=0 $wl NA_HOST_HALVES=0
  one NA_WN_DENSE=0 $wl NA_WN_DENSE=0
  one NA_WN_PACK=2 $wl NA_WN_PACK=2
  one NA_WN_PACK=2,NA_HOST_HALVES=0 $wl NA_WN_PACK=2 NA_HOST_HALVES=0
  one NA_WN_PACK=1 $wl NA_WN_PACK=1
  one NA_SP_SPB=2 $wl NA_SP_SPB=2
  one NA_SP_SPB=1 $wl NA_SP_SPB=1
  one NA_WN_KERNEL=frame $wl NA_WN_KERNEL=frame
  one NA_HOST_CHAINS=3 $wl NA_HOST_CHAINS=3
done
