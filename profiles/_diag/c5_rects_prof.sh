#!/bin/bash
# kernel averages of c5 with and without the records written by navsim_regen:  profiles/gpu.sh --timeout 900 -- 'bash profiles/_diag/c5_rects_prof.sh'
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_c5_rects"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in "" "--rects"; do
  T="plain"; [ -n "$V" ] && T="rects"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$T" -o t -- python3 "$R/bench.py" --workload c5 $V --steps 100 --repeats 1 --no-noise-off-pass --no-cold-pass --no-cpu-baseline > "$OUT/$T.log" 2>&1
  echo "== $T"; python3 - "$OUT/$T" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("regen_", "navsim_step", "launch_order")):
            print("%-110s %6s %10.1f" % (row["Name"][:110], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
rm -rf "$OUT/plain" "$OUT/rects"
