#!/bin/bash
# per-kernel times of the split step:  profiles/_diag/r04_split_trace.sh <lib.so> <tag> "<bench args>"
R="${GRAFT_REPO_ROOT:-/root/repo}"
LIB="$1"; TAG="$2"; ARGS="$3"
OUT="$R/gpurun_out/r04_split/trace_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NAVSIM_LIB="$R/build/$LIB"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o trace -- python3 "$R/bench.py" $ARGS --no-cpu-baseline --repeats 1 --no-noise-off-pass --no-cold-pass --steps 100 > "$OUT/bench.log" 2>&1
cd "$R"
f=$(ls "$OUT"/t/*/*kernel_stats.csv "$OUT"/t/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("split_","navsim_step","launch_order")):
        short = n.split("(")[0].replace("(anonymous namespace)::","")[:70]
        print("%-72s calls %6s avg %9.1f ns  total %.3f ms" % (short, r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"])/1e6))
PY
rm -rf "$OUT/t"
