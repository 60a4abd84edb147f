# usage: tools/wl_bench.sh "<workload>[:streams][@VAR=value...] ..." [steps] -- bench.py --workload variants with extra environment, one line each
cd $GRAFT_REPO_ROOT
steps=${2:-500}
for v in $1; do
  head=${v%%@*}
  wl=${head%%:*}; st=""; [ "$wl" != "$head" ] && st="--streams ${head#*:}"
  envs=""; rest=${v#*@}; [ "$rest" != "$v" ] && envs=$(echo "$rest" | tr '@' ' ')
  env $envs python bench.py --workload $wl $st --steps $steps --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1000,2), 'us', round(d['value'],1), d['unit'], 'frac', round(d['roofline']['frac'],4), d['roofline'].get('kernel'))"
done
