#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5prof -o c5 -- python3 $R/bench.py --workload c5 --no-cpu-baseline > $R/gpurun_out/c5prof.log 2>&1
cd $R
tail -1 gpurun_out/c5prof.log | cut -c1-200
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/c5prof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
