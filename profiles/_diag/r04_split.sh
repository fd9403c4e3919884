#!/bin/bash
# split step vs fused kernel, same process settings:  profiles/_diag/r04_split.sh <lib.so> "<bench args>" <split values...>
R="${GRAFT_REPO_ROOT:-/root/repo}"
LIB="$1"; ARGS="$2"; shift; shift
for rep in 1 2; do
for G in "$@"; do
  NAVSIM_LIB="$R/build/$LIB" python3 $R/bench.py $ARGS --step-split $G --no-cpu-baseline --repeats 3 --no-noise-off-pass --no-cold-pass 2>&1 | python3 -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('split $G FAILED'); sys.exit()
d=json.loads(ls[-1]); print('split $G', [round(v/1e6,2) for v in d['repeats']['values']], 'ms/step %.4f' % d['ms_per_step'])"
done; done
