# usage: tools/ab_bench.sh "<variant list>" [steps] -- A/B bench of library variants on one box, interleaved, 3 rounds.
# NA_AB_ARGS: extra bench.py arguments (e.g. "--workload nano").
# variant = <suffix>[@VAR=value[@VAR=value...]]: libNeuralAudioCAPI<suffix>.so ("-" = the default library) with extra environment
cd $GRAFT_REPO_ROOT
steps=${2:-500}
for rep in 1 2 3; do
for v in $1; do
  sfx=${v%%@*}; [ "$sfx" = "-" ] && sfx=""
  envs=""; rest=${v#*@}; [ "$rest" != "$v" ] && envs=$(echo "$rest" | tr '@' ' ')
  env NA_LIB_SUFFIX=$sfx $envs python bench.py --steps $steps --warmup 50 --no-cpu-baseline ${NA_AB_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1000,2), round(d['roofline']['frac'],4))"
done; done
