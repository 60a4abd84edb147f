cd $GRAFT_REPO_ROOT
for suf in _ldsfix "" _ldsfix _ldsfix; do echo "== variant '$suf'"; for i in 1 2; do NA_LIB_SUFFIX=$suf python tools/runs/r06n_quadrace.py std 64 400 128 2>&1 | grep "^mix" | cut -c1-200; done; done
