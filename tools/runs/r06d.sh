mkdir -p gpurun_out/r06d
( time timeout 600 python -m pytest tests/test_gpu_recurrent_quad.py tests/test_gpu_batch.py -x -q -m gpu -k "lstm or config4 or recurrent or two_layer or quad" --durations=5 ) > gpurun_out/r06d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06d/tests.log
for v in "" "NA_REC_NOPIPE=1"; do
  echo "== $v" >> gpurun_out/r06d/cfg4.log
  env $v python bench.py --workload config4 --no-cpu-baseline --no-host-path --steps 2000 --warmup 200 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_avg'], d['roofline']['frac'], d['parity_rms'], d['latency_per_buffer_ms'])" >> gpurun_out/r06d/cfg4.log 2>&1
done
for v in "" "NA_REC_NOPIPE=1"; do
  echo "== lstm 2x16 only, 1024 / 512 / 2048 streams $v" >> gpurun_out/r06d/cfg4.log
  env $v python tools/runs/r06d_lstm.py >> gpurun_out/r06d/cfg4.log 2>&1
done
tail -n 8 gpurun_out/r06d/tests.log; cat gpurun_out/r06d/cfg4.log
