cd $GRAFT_REPO_ROOT
for suf in "" _nopk "" _nopk; do
  echo "== variant '$suf'"
  for a in "std+nano 24 300 192" "std 64 300 128"; do NA_LIB_SUFFIX=$suf python tools/runs/r06n_quadrace.py $a 2>&1 | grep "^mix"; done
done
