#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel trace/stats + PMC passes) into a small text summary + traffic.json.

usage: summarize.py <dir with trace/ and pmc_*/ sub-directories>  (profiles/run_profiles.sh calls it on the GPU box)

Which launches of navsim_step_kernel are STEPS: the kernel also runs in its reset-only form -- once for the first
observation (reset_obs) and, in worlds that regenerate maps (c5), once inside every navsim_regen.  Those launches are
told apart by dispatch order, not by duration: a launch of the step kernel whose predecessor on the
device is one of navsim_regen's kernels (regen_* / dt_* / rect_* / costmap / plan) is the reset-only launch that
navsim_regen ends with; the very first launch of the process is reset_obs.  Everything else is a step.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("NAVSIM_PROFILE_KERNEL", "navsim_step_kernel")
REGEN_MARKS = ("regen_", "dt_columns", "dt_rows", "rect_", "costmap_kernel", "plan_kernel", "first_obs")


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


def step_flags(rows, name_key, order_key):
    """rows of one csv (kernel trace or counter collection) -> {dispatch id: True for a step launch of KERNEL}."""
    seen = {}
    for r in rows:
        seen[int(r[order_key])] = r[name_key]
    flags, prev, first = {}, "", True
    for d in sorted(seen):
        nm = seen[d]
        if KERNEL in nm:
            flags[d] = not first and not any(m in prev for m in REGEN_MARKS)
            first = False
        prev = nm
    return flags


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
step_ns, reset_ns, last = [], [], None
for f in find("*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    okey = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else "Start_Timestamp"
    fl = step_flags(rows, "Kernel_Name", okey)
    for row in rows:
        if KERNEL in row.get("Kernel_Name", ""):
            ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if fl[int(row[okey])]:
                step_ns.append(ns)
                last = row
            else:
                reset_ns.append(ns)
s = step_ns
if step_ns:
    print("%s step launches=%d avg_us=%.2f min_us=%.2f max_us=%.2f   (reset-only launches set aside: %d, avg_us=%.2f)"
          % (KERNEL, len(s), sum(s) / len(s) / 1e3, min(s) / 1e3, max(s) / 1e3, len(reset_ns),
             (sum(reset_ns) / len(reset_ns) / 1e3) if reset_ns else 0.0))
    print("last step dispatch:", {k: last[k] for k in last if k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")})
print("== PMC (per %s STEP dispatch, mean over the step dispatches of each pass) ==" % KERNEL)
res = {}
for f in find("*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    fl = step_flags(rows, "Kernel_Name", "Dispatch_Id")
    acc = defaultdict(list)
    for row in rows:
        if KERNEL in row.get("Kernel_Name", "") and fl[int(row["Dispatch_Id"])]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
for k in sorted(res):
    print("%-22s %.6g" % (k, res[k]))
if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
    f = res.get("FETCH_SIZE", 0.0) * 1024.0
    w = res.get("WRITE_SIZE", 0.0) * 1024.0
    print("FETCH bytes/launch as reported %.4g  (x2 if wide streaming: %.4g)   WRITE bytes/launch %.4g" % (f, 2 * f, w))
    json.dump({"fetch_bytes_reported": f, "write_bytes": w, "counters": res}, open(os.path.join(out, "pmc.json"), "w"), indent=1)

# what the profiled command was (its own JSON line) and which sources it ran
bench = {}
for f in find("trace.log") + find("bench.json"):
    for ln in open(f, errors="replace"):
        if ln.startswith('{"metric"'):
            try:
                bench = json.loads(ln)
            except Exception:
                pass
head = "?"
hp = os.path.join(ROOT, "profiles", ".head")
if os.path.exists(hp):
    head = open(hp).read().strip()
head = os.environ.get("NAVSIM_COMMIT", head)
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
try:
    from nav_gym_amd import lib
    src_sha = lib.source_hash()
except Exception:
    src_sha = None

valu_issue = None
if "SQ_ACTIVE_INST_VALU" in res and "GRBM_GUI_ACTIVE" in res:
    # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; a SIMD issues one vector instruction per 4 cycles (wave64 on 16 lanes)
    n_simd = 1024.0
    cycles = res["GRBM_GUI_ACTIVE"] / 8.0
    valu_issue = res["SQ_ACTIVE_INST_VALU"] * 4.0 / (n_simd * cycles)
    print("vector issue: SQ_ACTIVE_INST_VALU %.4g x 4 / (%d SIMDs x %.4g cycles) = %.3f of the issue cycles"
          % (res["SQ_ACTIVE_INST_VALU"], n_simd, cycles, valu_issue))
if "TCC_EA0_RDREQ_128B_sum" in res:
    rd = res.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + res.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + res["TCC_EA0_RDREQ_128B_sum"] * 128
    w = res.get("WRITE_SIZE", 0.0) * 1024.0
    print("HBM read bytes/launch from the request-size split %.4g (FETCH_SIZE x2 = %.4g) ; + writes = %.4g"
          % (rd, 2 * res.get("FETCH_SIZE", 0) * 1024.0, rd + w))
    cfgb = bench.get("config", {})
    json.dump({"hbm_bytes_per_launch": rd + w, "read_bytes": rd, "write_bytes": w,
               "commit": head, "kernel_src_sha": src_sha,
               "envs_per_gpu": cfgb.get("envs_per_gpu"), "rect_table": cfgb.get("rect_table"),
               "indoor_ratio": cfgb.get("indoor_ratio", 0.0),
               "workload": cfgb.get("workload"),
               "kernel_avg_us": (sum(s) / len(s) / 1e3) if step_ns else None,
               "step_launches": len(step_ns), "reset_only_launches_set_aside": len(reset_ns),
               "valu_issue_frac": valu_issue,
               "method": "TCC_EA0_RDREQ_{32,64,128}B x size + WRITE_SIZE x 1024 (separate --pmc passes); "
                         "FETCH_SIZE x 2 agrees (gfx950 tallies 128-B reads at 64 B); step launches told from "
                         "reset-only launches by dispatch order",
               "counters": res}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
if "TCC_HIT_sum" in res:
    print("L2 hit rate %.3f" % (res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])))
