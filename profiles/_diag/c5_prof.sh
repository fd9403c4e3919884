#!/bin/bash
# kernel trace of the c5 bench: per-kernel average durations of navsim_regen's pipeline
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-c5prof}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o c5 -- python3 $R/bench.py --workload c5 ${NAVSIM_BENCH_ARGS:-} --no-cpu-baseline --repeats 1 --no-noise-off-pass --no-cold-pass > $R/gpurun_out/$TAG.log 2>&1
cd $R
tail -1 gpurun_out/$TAG.log | cut -c1-200
python3 - "$TAG" <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1],recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:90].replace("(anonymous namespace)::",""), r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
rm -rf gpurun_out/$TAG
