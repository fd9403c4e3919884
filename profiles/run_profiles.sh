#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats and separate PMC passes of the bench command.
# usage: profiles/run_profiles.sh <tag> [bench args...]
set -u
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
python3 "$R/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
