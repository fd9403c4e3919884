#!/bin/bash
# kernel timeline (start / end per launch, by queue) of any python command, around the middle of its run:
#   profiles/_diag/timeline.sh <out file under gpurun_out> <rows> <script.py> [args...]
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$1"; mkdir -p "$(dirname "$OUT")"; ROWS="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o trace -- python3 "$R/$1" "${@:2}" > "$OUT.log" 2>&1
cd "$R"
tail -n 1 "$OUT.log" | cut -c1-300
python3 - "$OUT" "$ROWS" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
steps = [i for i, r in enumerate(rows) if "navsim_step" in r["Kernel_Name"]]
mid = steps[len(steps) // 2] if steps else len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
with open(sys.argv[1], "w") as out:
    for r in rows[mid:mid + n]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]
        line = "%9.1f %9.1f %7.1f q%-3s grid %-8s wg %-5s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), name)
        print(line); out.write(line + "\n")
PY
rm -rf /tmp/tl
