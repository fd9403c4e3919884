#!/bin/bash
# counters of ped_update_kernel on c3 (the step kernel's are in profiles/r03_c3): where do its 34 us go?
#   profiles/gpu.sh --timeout 900 -- 'bash profiles/_diag/ped_update_pmc.sh'
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_pedupd"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload c3 --repeats 1 --no-noise-off-pass --no-cold-pass --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_issue" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_issue.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/pmc_f64" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_f64.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq2" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq3" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_sq3.log" 2>&1
NAVSIM_PROFILE_KERNEL=ped_update_kernel python3 "$R/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -3 "$OUT"/pmc_f64.log "$OUT"/pmc_sq2.log "$OUT"/pmc_sq3.log > "$OUT/pmc_logs_tail.txt" 2>&1
rm -rf "$OUT"/trace "$OUT"/pmc_*/
cat "$OUT/summary.txt"
