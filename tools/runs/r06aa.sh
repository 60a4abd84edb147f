# layer pipeline, layer-1 wave: next group's h0 values and the progress word requested in front of the current group's steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_recurrent_quad.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "config4 or lstm or LSTM or recurrent" 2>&1 | tail -2
python tools/runs/r06d_lstm.py 2>&1 | grep streams | tee $O/lstm2x16.txt
for i in 1 2 3; do python bench.py --workload config4 --no-cpu-baseline --no-host-path --steps 1000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4 us per step', d['ms_per_step']*1000, 'frac', d['roofline']['frac'])"; done | tee $O/cfg4.txt
