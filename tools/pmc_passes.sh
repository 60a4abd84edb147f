#!/bin/bash
# Collect PMC counters for the bench kernel in separate passes (one rocprofv3 run per counter group).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [extra env assignments...]   (NA_PMC_BENCH_ARGS: extra bench.py arguments, e.g. "--workload config3")
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $group -f csv -d $OUT/pass$i -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-host-path --rotate 0 --no-exact-f32 ${NA_PMC_BENCH_ARGS:-} > $OUT/pass$i.log 2>&1
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
ls $OUT
