#!/bin/bash
# The bench lines of a round WITHOUT a profiler (the numbers quoted in DESIGN.md):  profiles/run_round_clean.sh <tag>
# -> gpurun_out/prof_<tag>_<workload>/bench_clean.json (+ bench_driver_shape.json for c2, bench_clean_indoor*.json)
TAG="$1"
R="${GRAFT_REPO_ROOT:-/root/repo}"
for W in c1 c2 c3 c4 c5; do
  O="$R/gpurun_out/prof_${TAG}_$W"; mkdir -p "$O"
  python3 "$R/bench.py" --workload $W 2>/dev/null | grep '^{"metric"' | tail -1 > "$O/bench_clean.json"
done
O="$R/gpurun_out/prof_${TAG}_c2"
python3 "$R/bench.py" --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > "$O/bench_driver_shape.json"
for I in 0.5 1.0; do
  python3 "$R/bench.py" --indoor-ratio $I --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$O/bench_clean_indoor$I.json"
done
python3 "$R/bench.py" --workload c4 --indoor-ratio 1.0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$R/gpurun_out/prof_${TAG}_c4/bench_clean_indoor1.0.json"
python3 "$R/bench.py" --workload c5 --pregen --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$R/gpurun_out/prof_${TAG}_c5/bench_clean_pregen.json"
python3 "$R/bench.py" --no-rects --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$O/bench_clean_norects.json"
for E in 512 1024; do
  python3 "$R/bench.py" --envs $E --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$O/bench_clean_envs$E.json"
done
python3 "$R/bench.py" --workload c5 --graph on --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > "$R/gpurun_out/prof_${TAG}_c5/bench_clean_graph.json"
