#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rt -o rt -- python3 $R/profiles/_diag/regen_twice.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/rt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
pos = collections.defaultdict(lambda: [[], []])
seen = collections.Counter()
for i, (n, d) in enumerate(zip(names, dur)):
    key = None
    for k in ("regen_indoor", "regen_maps", "regen_commit", "navsim_step_kernel"):
        if k in n: key = k
    if key is None: continue
    if key == "navsim_step_kernel":
        if "regen_commit" not in names[i - 1]: seen.clear(); continue     # the step itself
        key = "first observations"
    which = seen[key]; seen[key] += 1
    if which < 2: pos[key][which].append(d)
for k, (a, b) in pos.items():
    a, b = a[20:], b[20:]
    print("%-20s after the step kernel %.1f us   repeated right after itself %.1f us   (%d launches)" % (k, sum(a) / len(a) / 1e3, sum(b) / len(b) / 1e3, len(a)))
PY
rm -rf gpurun_out/rt
