#!/bin/bash
# one PMC pass (instruction mix) for the bench kernel: tools/pmc_quick.sh <outdir> [ENV=VAL...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $OUT/pass1 -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pass1.log 2>&1
python $R/tools/pmc_summary.py $OUT
