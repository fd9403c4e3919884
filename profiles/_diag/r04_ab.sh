#!/bin/bash
# same-box A/B over several workloads: profiles/_diag/r04_ab.sh "libA.so libB.so" "c2|--workload c2" "c2_512|--workload c2 --envs 512" ...
R="${GRAFT_REPO_ROOT:-/root/repo}"
LIBS="$1"; shift
for spec in "$@"; do
  name="${spec%%|*}"; args="${spec#*|}"
  echo "== $name"
  bash $R/profiles/_diag/lib_ab.sh "$args" $LIBS
done
