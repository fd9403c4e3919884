#!/bin/bash
# one PMC pass of the bench: profiles/pmc_pass.sh <tag> "<counter list>" [bench args]
set -u
TAG="$1"; CNT="$2"; shift 2
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$R/bench.py" "$@" --repeats 1 --no-noise-off-pass --no-cold-pass --no-cpu-baseline > "$OUT/run.log" 2>&1
python3 "$R/profiles/summarize.py" "$OUT" 2>&1 | grep -v "^==" | grep -v "^{"
