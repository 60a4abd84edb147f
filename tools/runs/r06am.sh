# A1 Standard over the batch size under the launch-shape knobs (bench.py --streams N, event marks): do the size thresholds still sit where they should?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06am; mkdir -p $O
for S in 128 256 384 512 640 768 1024 1536 2048 3072 4096; do
  line="streams $S:"
  for k in "X=1" "NA_HOST_HALVES=0" "NA_SP_SPB=1" "NA_SP_SPB=2" "NA_SP_SPB=1 NA_HOST_HALVES=0"; do
    v=$(env $k python bench.py --streams $S --no-cpu-baseline --no-host-path --no-parity-check --rotate 0 --no-exact-f32 --steps 600 --warmup 150 --ramp-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f/%s' % (d['ms_per_step']*1000, d['roofline'].get('launches_per_step')))")
    line="$line  [$k] $v"
  done
  echo "$line" | tee -a $O/sizes.txt
done
