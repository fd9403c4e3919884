#!/bin/bash
# kernel trace of profiles/_diag/gym_c3_steps.py:  profiles/_diag/r04_replan_trace.sh <out tag>   (NAVSIM_LIB optional)
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r05_replan/trace_$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o trace -- python3 "$R/profiles/_diag/gym_c3_steps.py" > "$OUT/log.txt" 2>&1
cd "$R"
tail -n 1 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_stats.csv"), recursive=True):
    for row in list(csv.DictReader(open(f)))[:14]:
        print(row["Name"][:90].replace("(anonymous namespace)::", ""), row["Calls"], round(float(row["AverageNs"]) / 1e3, 2), row["Percentage"])
PY
rm -rf "$OUT/t"
