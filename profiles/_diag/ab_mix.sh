#!/bin/bash
# same-box A/B: the round-5 tree (build/r05) and this tree with each of the given libraries of build/ (each twice, alternated)
#   profiles/_diag/ab_mix.sh "<bench args>" lib1.so lib2.so ...
R="${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="$1"; shift
one() {  # tree, label, lib
  NAVSIM_LIB="$3" python3 $1/bench.py $ARGS --no-cpu-baseline --no-extras --repeats 3 --no-noise-off-pass --no-cold-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms/step %.4f' % d['ms_per_step'])"
}
for rep in 1 2; do
  (unset NAVSIM_LIB; python3 $R/build/r05/bench.py $ARGS --no-cpu-baseline --no-extras --repeats 3 --no-noise-off-pass --no-cold-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r05', [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms/step %.4f' % d['ms_per_step'])")
  for L in "$@"; do one "$R" "$L" "$R/build/$L"; done
done
