#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
for L in p6ps0 p6ps512; do
  export NAVSIM_LIB=$R/build/libnavsim_$L.so
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ts
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o trace -- python3 $R/bench.py --workload c5 --steps 100 --repeats 1 --no-noise-off-pass --no-cold-pass --no-extras --no-cpu-baseline --graph off > /tmp/ts.log 2>&1
  echo "== $L"
  python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/ts/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:8]:
        n=row["Name"].replace("(anonymous namespace)::","")[:70]
        if "math_kernel" in n or "at::" in n: continue
        print("%-72s %6s %9.2f" % (n,row["Calls"],float(row["AverageNs"])/1e3))
PY
done
