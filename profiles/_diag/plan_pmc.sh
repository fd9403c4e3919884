#!/bin/bash
# PMC counters of the planner's launches (regen_plan_kernel, replan_kernel) in the reference's configuration through the gym API,
# navsim_regen after every step (NAVSIM_PIPELINE=0: every planner launch has searches to run).  Runs on the GPU box.
# usage: profiles/_diag/plan_pmc.sh <out name under gpurun_out/>
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NAVSIM_PIPELINE=0 NAVSIM_ENVS=1024 NAVSIM_STEPS=40
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/a" -o pmc -- python3 "$R/profiles/_diag/gym_refdef_steps.py" > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o pmc -- python3 "$R/profiles/_diag/gym_refdef_steps.py" > "$OUT/b.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d "$OUT/c" -o pmc -- python3 "$R/profiles/_diag/gym_refdef_steps.py" > "$OUT/c.log" 2>&1
python3 "$R/profiles/_diag/pmc_by_kernel.py" "$OUT" regen_plan_kernel replan_kernel > "$OUT/plan_pmc.txt" 2>&1
rm -rf "$OUT/a" "$OUT/b" "$OUT/c"
cat "$OUT/plan_pmc.txt"
