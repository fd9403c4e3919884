#!/bin/bash
# registers / scratch / LDS / duration of every kernel of a diag script:  profiles/_diag/kernel_regs.sh <script.py>
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kr && rocprofv3 --kernel-trace --output-format csv -d /tmp/kr -o trace -- python3 "$R/$1" > /tmp/kr.log 2>&1
cd "$R"; tail -n 1 /tmp/kr.log
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/kr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
        acc[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print("%-70s vgpr %s agpr %s sgpr %s scratch %s lds %s wg %s grid %s  n %d avg %.1f us" % (k + (len(v), sum(v) / len(v) / 1e3)))
PY
rm -rf /tmp/kr
