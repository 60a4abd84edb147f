cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
( timeout 900 rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "thread apply all bt 12" --args python -m pytest tests -x -q -m gpu -p no:faulthandler > gpurun_out/r06i/gdb_full.txt 2>&1 )
grep -n "SIGSEGV\|received signal" gpurun_out/r06i/gdb_full.txt | head
awk '/received signal/{p=1} p{print} /info sharedlibrary/{exit}' gpurun_out/r06i/gdb_full.txt | grep -v "^0x0000\|Yes (\*)\|Yes   " | head -150 | cut -c1-260
echo "== stall + multi only (warm)"
( timeout 300 python -m pytest tests/test_gpu_stall.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -5 )
echo "== batch + multi only (warm)"
( timeout 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -5 )
