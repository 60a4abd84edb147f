#!/bin/bash
# round 4: the two workgroups of a CU out of phase -- start delay of the second half of the grid (100 .. 600 x 64 cycles = 3 .. 20 us)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
NA_AB_ARGS="--no-parity-check" timeout 1500 bash tools/ab_bench.sh "_quick _dl100 _dl200 _dl300 _dl400 _dl600" 1000 2>&1 | tee $O/ab_delay.txt
