cd $GRAFT_REPO_ROOT
T=tests/test_gpu_batch.py::test_table_launches_inside_a_captured_multi_unit_batch_replay_their_own_tables
for suf in "" _qp1 _qp2 _qp4; do
  f=0
  for i in $(seq 1 14); do
    NA_LIB_SUFFIX=$suf NA_REC_QUAD_MIN=1 NA_T_DIAG=1 timeout 200 python -m pytest $T -x -q -m gpu -s > /tmp/o.txt 2>&1 || { f=$((f+1)); grep -E "DIAG row .* one" /tmp/o.txt | head -2; }
  done
  echo "== variant '$suf': $f failures of 14"
done
