#!/bin/bash
# round 4: threads per arena re-swept with the index rows in LDS (pick_step_block's thresholds date from round 2)
#   profiles/_diag/r04_block_sweep.sh  ->  stdout: arenas, block, value (M env-steps/s), kernel_ms
R="${GRAFT_REPO_ROOT:-/root/repo}"
for E in 256 512 768 1024 1536 2048 3072 4096; do
  for B in 0 256 512 1024; do
    python3 "$R/bench.py" --envs $E --step-block $B --no-extras --no-cpu-baseline --no-noise-off-pass --no-cold-pass --repeats 3 2>/dev/null \
      | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print($E, $B, [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])
"
  done
done
