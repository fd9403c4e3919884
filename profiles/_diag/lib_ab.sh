#!/bin/bash
# same-box A/B of library builds in build/:  profiles/_diag/lib_ab.sh "<bench args>" lib1.so lib2.so ...   (each twice, alternated)
R="${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="$1"; shift
for rep in 1 2; do
for L in "$@"; do
  NAVSIM_LIB="$R/build/$L" python3 $R/bench.py $ARGS --no-cpu-baseline --repeats 3 --no-noise-off-pass --no-cold-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L', [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms/step %.4f' % d['ms_per_step'])"
done; done
