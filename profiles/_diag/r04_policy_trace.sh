#!/bin/bash
# kernel trace of profiles/policy_cost.py with a variant library:  profiles/_diag/r04_policy_trace.sh <lib name> <out tag>
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r04_fuse/trace_$2"; mkdir -p "$OUT"
export NAVSIM_LIB="$R/build/$1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o trace -- python3 "$R/profiles/policy_cost.py" > "$OUT/log.txt" 2>&1
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("policy_", "ped_scan", "navsim_step")):
            print(row["Name"][:100], row["Calls"], row["AverageNs"])
seen = set()
for f in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"][:60]
        if n in seen or not any(k in n for k in ("policy_", "ped_scan")): continue
        seen.add(n)
        print(n, {k: row[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X") if k in row})
PY
rm -rf "$OUT/t"
