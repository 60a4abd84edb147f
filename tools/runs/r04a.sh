#!/bin/bash
# round 4, first GPU call: the suite on the new build, the Infinity Cache working-set probe, bench lines of every BASELINE config
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -5 $O/pytest.log
timeout 300 tools/microbench/bin/mall_working_set > $O/mall.txt 2>&1; cat $O/mall.txt
timeout 300 python bench.py > $O/bench_standard.json 2> $O/bench_standard.err; head -c 1500 $O/bench_standard.json; echo
for w in config3 config4 config5 nano feather lite a2full; do
  timeout 300 python bench.py --workload $w --steps 500 --warmup 100 > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$w", "us/step %.2f" % (d["ms_per_step"] * 1e3), "kernel %.2f" % (d["kernel_ms_avg"] * 1e3), r["bound"], "frac %.3f" % r["frac"], "wall %.3f" % r["frac_wall_clock"],
          "meas", r["frac_measured_bytes"], "parity", d["parity_rms"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$w failed", e, open("$O/bench_$w.err").read()[-800:])
PY
done
