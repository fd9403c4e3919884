#!/bin/bash
# the rect records in global memory (--rect-lds 1) against staged in LDS (--rect-lds 2) over the batch size, c2 world
R="${GRAFT_REPO_ROOT:-/root/repo}"
for E in 128 256 512 1024 1536 2048 2560 3072 4096; do
  for M in 1 2; do
    python3 $R/bench.py --envs $E --rect-lds $M --no-cpu-baseline --repeats 3 --no-noise-off-pass --no-cold-pass ${NAVSIM_SWEEP_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('envs $E rect_lds $M', [round(v/1e6,2) for v in d['repeats']['values']], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
  done
done
