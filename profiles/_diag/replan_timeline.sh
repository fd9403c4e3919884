#!/bin/bash
# kernel timeline (start / end per launch) of profiles/_diag/gym_c3_steps.py around the middle of the timed window:
#   profiles/_diag/replan_timeline.sh <out tag>      -> gpurun_out/r05_replan/timeline_<tag>.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r05_replan"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o trace -- python3 "$R/profiles/_diag/gym_c3_steps.py" > "$OUT/timeline_$1.log" 2>&1
cd "$R"
python3 - "$OUT/timeline_$1.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
with open(sys.argv[1], "w") as out:
    for r in rows[mid:mid + 40]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]
        line = "%9.1f %9.1f  q%-3s grid %-8s wg %-5s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                          r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), name)
        print(line); out.write(line + "\n")
PY
rm -rf /tmp/tl
