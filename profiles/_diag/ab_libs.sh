#!/bin/bash
# same-box A/B of library builds (build/libnavsim_<name>.so, profiles/_diag/build_variant.sh):
#   profiles/_diag/ab_libs.sh <workload> <name> [<name> ...]      ("tree" = the in-tree library)
R="${GRAFT_REPO_ROOT:-/root/repo}"; WL="$1"; shift
for i in 1 2; do
  for n in "$@"; do
    if [ "$n" = "tree" ]; then unset NAVSIM_LIB; else export NAVSIM_LIB="$R/build/libnavsim_$n.so"; fi
    echo -n "$n: "
    python3 "$R/bench.py" --workload "$WL" --no-cpu-baseline --no-cold-pass --no-noise-off-pass 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M  kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us', [round(v/1e6,2) for v in d['repeats']['values']])"
  done
done
