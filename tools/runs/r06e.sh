mkdir -p gpurun_out/r06e
for pm in 0 1 2 3 4 5; do echo "== NA_PM=$pm" >> gpurun_out/r06e/pm.log; NA_PM=$pm python tools/runs/r06d_lstm.py 2>&1 | grep streams >> gpurun_out/r06e/pm.log; 
 NA_PM=$pm python bench.py --workload config4 --no-cpu-baseline --no-host-path --steps 2000 --warmup 200 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('config4', d['ms_per_step'], d['kernel_ms_avg'])" >> gpurun_out/r06e/pm.log 2>&1
done
cat gpurun_out/r06e/pm.log
