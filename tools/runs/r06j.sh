cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
for v in "" "NA_TEST_TORCH_FIRST=1"; do
  echo "== full suite $v" >> gpurun_out/r06j/log.txt
  ( time env $v timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) >> gpurun_out/r06j/log.txt 2>&1
done
echo "== NA_BATCH_NO_GRAPH=1 batch + multi + fuzz" >> gpurun_out/r06j/log.txt
( NA_BATCH_NO_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 ) >> gpurun_out/r06j/log.txt 2>&1
cat gpurun_out/r06j/log.txt | cut -c1-200
