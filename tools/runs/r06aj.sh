# config 3 / config 5 / headline under the batch-level knobs at the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aj; mkdir -p $O
one() { local label="$1"; shift; local wl="$1"; shift
  env "$@" python bench.py $wl --no-cpu-baseline --no-host-path --no-parity-check --rotate 0 --no-exact-f32 --steps 800 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-20s %-26s %.2f us per step  launches %s' % ('$wl', '$label', d['ms_per_step']*1000, r.get('launches_per_step')))" | tee -a $O/sweep.txt
}
for wl in "--workload config3" "--workload config5" "--steps 800"; do
  one default "$wl" X=1
  one default-again "$wl" X=1
  one NA_HOST_HALVES=0 "$wl" NA_HOST_HALVES=0
  one NA_HOST_CHAINS=3 "$wl" NA_HOST_CHAINS=3
  one NA_WN_NT=0 "$wl" NA_WN_NT=0
  one NA_WN_NT_MB=100 "$wl" NA_WN_NT_MB=100
  one NA_SP_SPB=1 "$wl" NA_SP_SPB=1
done
