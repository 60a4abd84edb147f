cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
which gdb rocgdb > gpurun_out/r06h/gdb.txt 2>&1
for v in "" "NA_TEST_NO_WARM=1"; do
  echo "== multi alone $v" >> gpurun_out/r06h/log.txt
  ( env $v timeout 300 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -15 ) >> gpurun_out/r06h/log.txt
done
echo "== full suite NO_WARM" >> gpurun_out/r06h/log.txt
( NA_TEST_NO_WARM=1 timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) >> gpurun_out/r06h/log.txt
echo "== full suite warm, again" >> gpurun_out/r06h/log.txt
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 ) >> gpurun_out/r06h/log.txt
if which rocgdb > /dev/null 2>&1; then
  echo "== rocgdb" >> gpurun_out/r06h/log.txt
  ( timeout 600 rocgdb -batch -ex run -ex bt -ex "info sharedlibrary" --args python -m pytest tests -x -q -m gpu 2>&1 | tail -80 ) >> gpurun_out/r06h/log.txt
fi
cat gpurun_out/r06h/gdb.txt; cat gpurun_out/r06h/log.txt | cut -c1-220
