#!/bin/bash
# same-box A/B of libraries of build/ over several workloads:  profiles/_diag/ab_libs.sh "c2 c3 c4" lib1.so lib2.so ...   (each twice, alternated)
R="${GRAFT_REPO_ROOT:-/root/repo}"
WL="$1"; shift
for w in $WL; do
for rep in 1 2; do
for L in "$@"; do
  NAVSIM_LIB="$R/build/$L" python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras --repeats 3 --no-noise-off-pass --no-cold-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', '$L', [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms/step %.4f' % d['ms_per_step'])"
done; done; done
