#!/bin/bash
# c5 with pipelined pre-generation + in-step install: rate, counters, step-kernel duration (rocprofv3 stats)
#   profiles/_diag/c5_pipe.sh <tag> [bench args...]      (NAVSIM_LIB passes through)
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r05_pipe"; mkdir -p "$OUT"; TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp -o trace -- python3 "$R/bench.py" --workload c5 --no-cpu-baseline --steps 400 --warmup 50 "$@" > "$OUT/log_$TAG.txt" 2>&1
cd "$R"
grep "^{" "$OUT/log_$TAG.txt" | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']/1e6,3), 'M', round(d['ms_per_step']*1e3,1), 'us/step', d['config']['regen_counters_rank0'])"
python3 - "$OUT/stats_$TAG.txt" <<'PY'
import csv, glob, sys
with open(sys.argv[1], "w") as out:
    for f in glob.glob("/tmp/cp/**/*kernel_stats.csv", recursive=True):
        for row in list(csv.DictReader(open(f)))[:8]:
            line = "%-60s %7s %10.2f %7s" % (row["Name"][:60].replace("(anonymous namespace)::", ""), row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"])
            print(line); out.write(line + "\n")
PY
rm -rf /tmp/cp
