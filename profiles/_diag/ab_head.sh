#!/bin/bash
# same-box A/B of the working tree against the committed HEAD built in build/wt_head (git worktree add build/wt_head HEAD;
# bash build/wt_head/nav-gym_amd/csrc/build.sh):  profiles/_diag/ab_head.sh <workload> [bench args...]
R="${GRAFT_REPO_ROOT:-/root/repo}"; WL="$1"; shift
one() { python3 "$1/bench.py" --workload "$WL" --no-cpu-baseline --no-cold-pass --no-noise-off-pass "${@:2}" 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2' if False else '', round(d['value']/1e6,3), 'M  kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us', [round(v/1e6,2) for v in d['repeats']['values']])"; }
for i in 1 2; do
  echo -n "head: "; one "$R/build/wt_head" "$@"
  echo -n "tree: "; one "$R" "$@"
done
