#!/bin/bash
# Round profile set, on the GPU box:  profiles/run_round.sh <round tag, e.g. r02>
# kernel trace + separate PMC passes of `bench.py --workload cN` for every BASELINE workload, and the kernel
# trace of the reference-faithful pedestrian pipeline (profiles/policy_cost.py).  Summaries land in
# gpurun_out/prof_<tag>_<workload>/ ; copy what is to be judged into profiles/<tag>_<workload>/.
set -u
TAG="$1"
R="${GRAFT_REPO_ROOT:-/root/repo}"
for W in c1 c2 c3 c4; do
  bash "$R/profiles/run_profiles.sh" "${TAG}_$W" --workload $W --steps 100 > /dev/null 2>&1
done
bash "$R/profiles/run_profiles.sh" "${TAG}_c5" --workload c5 --steps 100 > /dev/null 2>&1
OUT="$R/gpurun_out/prof_${TAG}_policy"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/profiles/policy_cost.py" > "$OUT/trace.log" 2>&1
cd "$R"
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
cp gpurun_out/policy_cost.json "$OUT/policy_cost.json" 2>/dev/null
for W in c1 c2 c3 c4 c5; do echo "== $W"; grep -E "navsim_step_kernel launches|HBM read bytes|L2 hit" "gpurun_out/prof_${TAG}_$W/summary.txt"; done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("policy_", "ped_scan", "navsim_step", "ped_update")):
            print(row["Name"][:90], row["Calls"], row["AverageNs"])
PY
rm -rf "$OUT/trace"
