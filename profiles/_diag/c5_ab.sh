#!/bin/bash
# A/B of navsim_regen kernels in the c5 bench: one kernel trace per library in build/
R="${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do
  echo "== $L"
  NAVSIM_LIB="$R/build/$L" bash $R/profiles/_diag/c5_prof.sh ab_$L 2>&1 | grep -v "at::native\|math_kernel\|rocprofv3\|dt_rows"
done
