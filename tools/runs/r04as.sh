#!/bin/bash
# round 4: ring history requested TWO layers ahead in the variants compiled for two waves per SIMD (_pfd1 = one layer ahead, as before)
cd /root/repo; export TMPDIR=/tmp; O=/root/repo/gpurun_out/r04as; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_spec.py tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "not recurrent and not lstm and not gru and not keras" < /dev/null > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E  " $O/pytest.log | head -5
NA_SP_SPB=1 timeout 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu < /dev/null > $O/pytest_spb1.log 2>&1; echo "pytest (SPB=1) rc $?"; grep -E "passed|failed" $O/pytest_spb1.log | tail -1
for rep in 1 2 3; do
  for v in "spb2" "spb1_pfd2" "spb1_pfd1"; do case $v in spb2) E="NA_X=0";; spb1_pfd2) E="NA_SP_SPB=1";; spb1_pfd1) E="NA_SP_SPB=1 NA_LIB_SUFFIX=_pfd1";; esac
    echo -n "standard $v: "; env $E timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-parity-check < /dev/null 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,2), round(j['roofline']['frac'],4))"
  done
done
for w in feather "standard --streams 512" "standard --streams 256" "lite --streams 512"; do for suf in "" _pfd1; do echo -n "$w ${suf:-pfd2}: "; NA_LIB_SUFFIX=$suf timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-parity-check --workload $w < /dev/null 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step']*1e3,2), round(j['roofline']['frac'],4))"; done; done
