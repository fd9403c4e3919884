#!/bin/bash
# kernel trace + stats of a diag script:  profiles/_diag/trace_stats.sh <out dir under gpurun_out> <tag> <script.py>   (env passes through)
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o trace -- python3 "$R/$3" > "$OUT/log_$2.txt" 2>&1
cd "$R"
tail -n 1 "$OUT/log_$2.txt"
python3 - "$OUT/stats_$2.txt" <<'PY'
import csv, glob, sys
with open(sys.argv[1], "w") as out:
    for f in glob.glob("/tmp/ts/**/*kernel_stats.csv", recursive=True):
        for row in list(csv.DictReader(open(f)))[:28]:
            line = "%-100s %7s %10.2f %7s" % (row["Name"][:100].replace("(anonymous namespace)::", ""), row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"])
            print(line); out.write(line + "\n")
PY
rm -rf /tmp/ts
