#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/policyprof -o pp -- python3 $R/profiles/policy_cost.py > $R/gpurun_out/policyprof.log 2>&1
cd $R
tail -12 gpurun_out/policyprof.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/policyprof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
