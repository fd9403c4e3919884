#!/bin/bash
# round 4, verdict item 7: profiles/policy_cost.py with each variant library (build/libnavsim_<name>.so)
#   profiles/_diag/r04_policy_ab.sh "<lib names>"
R="${GRAFT_REPO_ROOT:-/root/repo}"
for L in $1; do
  echo "== $L"
  NAVSIM_LIB="$R/build/$L" python3 "$R/profiles/policy_cost.py" 2>/dev/null | grep -v "^{\|^}" | tr -d '\n'; echo
done
