#!/bin/bash
# gpurun wrapper used for every profile / bench run of a round: records the commit the snapshot is taken from
# (profiles/.head, read by profiles/summarize.py -- the GPU box has no .git) and forwards to gpurun.
#   profiles/gpu.sh --timeout 900 -- '<command>'
cd "$(dirname "${BASH_SOURCE[0]}")/.."
H="$(git rev-parse --short HEAD)"
git diff --quiet HEAD -- . ':!gpurun_out' 2>/dev/null || H="$H-dirty"
echo "$H" > profiles/.head
exec /usr/local/graft/bin/gpurun "$@"
