#!/bin/bash
# same-box A/B of this tree against the round-5 tree kept under build/r05 (sources of 5d439c4 + its library):
#   profiles/_diag/ab_r05.sh "<bench args>"        (each twice, alternated)
R="${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="$1"
for rep in 1 2; do
for T in "$R/build/r05" "$R"; do
  python3 $T/bench.py $ARGS --no-cpu-baseline --no-extras --repeats 3 --no-noise-off-pass --no-cold-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$T'.replace('$R','.'), [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms/step %.4f' % d['ms_per_step'])"
done; done
