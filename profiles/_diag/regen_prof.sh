#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/regenprof -o rp -- python3 $R/profiles/regen_cost.py ${MODE:-regen+plan} > $R/gpurun_out/regenprof.log 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/regenprof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
