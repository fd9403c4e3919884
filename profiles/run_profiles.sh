#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats and separate PMC passes of the bench command.
# usage: profiles/run_profiles.sh <tag> [bench args...]
set -u
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$* --repeats 1 --no-noise-off-pass --no-cold-pass --no-extras --graph off"      # one kind of launch in the profile: K noise-on steps, plain launches
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
# read-request size split: calibrates FETCH_SIZE for THIS access pattern (guide: FETCH_SIZE tallies 128-B requests at 64 B)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_ea" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_ea.log" 2>&1
# vector-issue fraction (traffic.json valu_issue_frac -> bench.py roofline.issue_frac_profiled)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_issue" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_issue.log" 2>&1
# lanes on per vector instruction (bench.py roofline.valu.exec_lane_frac)
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_lanes" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_lanes.log" 2>&1
if [ "${NAVSIM_PROFILE_DEEP:-0}" = "1" ]; then
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum --output-format csv -d "$OUT/pmc_tcc" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_tcc.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d "$OUT/pmc_tcp" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_tcp.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d "$OUT/pmc_lat" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_lat.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq2" -o pmc -- python3 "$R/bench.py" $ARGS --no-cpu-baseline > "$OUT/pmc_sq2.log" 2>&1
fi
python3 "$R/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
# keep what is judged (stats csv, summary, traffic / pmc json, the bench line), drop the per-dispatch raw csvs:
# gpurun only copies back 64 MiB
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
grep -h '^{"metric"' "$OUT/trace.log" | tail -1 > "$OUT/bench.json"
rm -rf "$OUT"/trace "$OUT"/pmc_*/
cat "$OUT/summary.txt"
