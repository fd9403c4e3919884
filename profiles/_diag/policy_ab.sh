#!/bin/bash
# A/B of the reference-faithful pedestrian pipeline (profiles/policy_cost.py) between library builds in build/
R="${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do
  echo "== $L"
  NAVSIM_LIB="$R/build/$L" python3 $R/profiles/policy_cost.py 2>/dev/null | tail -3
  cat $R/gpurun_out/policy_cost.json 2>/dev/null | tr -d '\n'; echo
done
