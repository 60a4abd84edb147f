#!/bin/bash
# round 5: the "lo" product of the f16 split on the K = 16 MFMA (half the passes): parity, then time / clock / power against the K = 32 build
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05l}; mkdir -p $O
NA_LIB_SUFFIX=_lo16 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_spec.py -m gpu -x -q -k "standard or Standard or spec or chain" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
for v in _lo0 _lo16; do
  echo -n "$v  " >> $O/lo16.txt
  NA_LIB_SUFFIX=$v python tools/power_sample.py 3 >> $O/lo16.txt 2>&1
done; done
cat $O/lo16.txt
