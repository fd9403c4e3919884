#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused MI355X NavGym step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1: BASELINE.json configs[1] -- 4096 arenas, 1081-beam lidar, 500x500 static occupancy map,
diff-drive (KetiRobot kinematics), no pedestrians, auto-respawn of finished arenas in place.
N > 1: one rank per GPU -- either launched by torch.distributed.run (WORLD_SIZE set), or, when called plainly
as `python bench.py --gpus N`, this script spawns the N ranks itself before touching the GPU.
`--scaling strong`: the workload's TOTAL (c2: 4096) is split over the ranks by sharding.shard_range (ragged totals
allowed).  `--scaling weak`: every rank owns the workload's per-GPU arena count (c2: 4096 per GPU; c4 / c5: 1/8 of their
8-GPU totals, so `--workload c4 --gpus 8` is the configured 16384 and `--workload c5 --gpus 8` the configured 4096).
Default (`--scaling auto`): with N > 1 ONE invocation measures both -- `value` is the STRONG number (BASELINE.json's metric
reads "4096 envs ..., 1/2/4/8 MI355X": a fixed total), `value_weak` the weak one, each with its own `envs_total`; with
N = 1 the two coincide.  Arenas are keyed by their global index either way.  The step has no exchange, so `value` involves
no data-path collective; with N > 1 the line also carries `value_with_obs_gather`: the same K steps with the optional RCCL
all-gather of the observation rows after every step (`--gather none` skips that pass).
With N = 1 the default run appends short windows of the other BASELINE workloads (`other_workloads`: c3, c4, c5) and the
same steps through the gym API (`value_gym_api`: NavGymEnv.step with torch actions); `--no-extras` skips them.

One "step" = one launch of navsim_step over all local arenas, inputs resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))

# vector instructions of one probe round of the march (position 4, tile address 4, the two indices' rectangle addresses 2, packed
# cell 1, "no record" test 1, two-rectangle distance 9, exact sqrt of the integer d2 6, hit test 1, march step 3, next-t test and
# select 2), counted in the ISA of the c2 kernel
C1_CPU_KIND = "port (oracle/navsim_ref.c: a restatement, not the reference binary), one thread"
PROBE_ROUND_VALU = 33
PROBE_ROUND_FROM = ("navsim_step_kernel<256, false, FieldU16TT<false>, NAVSIM_MARCH_F32, 2, false, false>, loop .LBB13_100 of "
                    "hipcc -S --cuda-device-only nav-gym_amd/csrc/navsim_step_inst.hip (profiles/r06_c2/probe_round_isa.txt)")

WORKLOADS = {
    # envs = arenas per GPU (weak scaling), total = arenas of the whole job (strong scaling)
    "c1": dict(envs=1, total=1, beams=64, size=100, peds=0),
    "c2": dict(envs=4096, total=4096, beams=1081, size=500, peds=0),
    # diagnostic shape (not a BASELINE config): c2 on 200 x 200 maps, whose record table (10 KB) fits LDS eight times per CU
    "c2s": dict(envs=4096, total=4096, beams=1081, size=200, peds=0),
    "c3": dict(envs=4096, total=4096, beams=1081, size=500, peds=20),
    "c4": dict(envs=2048, total=16384, beams=1081, size=1000, peds=0),      # 16384 arenas over 8 GPUs
    # 4096 arenas over 8 GPUs, Husky, 20 pedestrians, a NEW random map at every episode end (navsim_regen)
    "c5": dict(envs=512, total=4096, beams=1081, size=500, peds=20, robot="husky", regen=True),
}


def shard_of(wl, scaling, rank, world_size):
    """(first global arena, arenas) of this rank.  weak: `envs` arenas per rank; strong: `total` split by
    sharding.shard_range (the first total % world ranks take one more)."""
    if scaling == "strong":
        from nav_gym_amd.sharding import shard_range
        return shard_range(wl["total"], rank, world_size)
    return rank * wl["envs"], wl["envs"]


def algorithmic_bytes_per_env_step(H, W, B, S, n_peds, s_map):
    """map streamed once + ranges written + packed obs written + scalar state."""
    return H * W * s_map + 4 * B + 4 * (S * B + 7 + 2 + 2) + 96 + 96 * n_peds


def build_sim(wl, base, E, seed=1234, device="cuda:0"):
    """The arenas [base, base + E) of the workload on `device`."""
    import numpy as np
    import torch
    from nav_gym_amd import abi, lib, robots, sim, world
    cfg = lib.default_config(n_envs=E, map_h=wl["size"], map_w=wl["size"], max_peds=max(wl["peds"], 1),
                             ped_model=abi.PED_SFM if wl["peds"] else abi.PED_NONE,
                             n_spawn=16, auto_reset=(abi.AUTORESET_NEXT_STEP if wl.get("next_step") else abi.AUTORESET_SAME_STEP),
                             seed=seed, env_index_base=base,
                             field_format={"f32": abi.FIELD_F32}.get(wl.get("field"), abi.FIELD_U16T))
    if wl["beams"] == 1081:
        world.lidar_1081(cfg)
    else:
        world.lidar_full_circle(cfg, wl["beams"])
    if wl.get("regen"):
        cfg.regen_cap = max(16, E // 16)          # arenas regenerated per step at most (c5: ~5 finish per step)
    cfg.regen_indoor_ratio = float(wl.get("indoor_ratio", 0.0))
    if wl.get("pipeline", 0):                     # --pregen-pipeline P: episodes shorter than 4 P steps restart in place ...
        cfg.regen_min_steps = 0 if wl.get("no_rule") else 4 * int(wl["pipeline"])      # ... or (--pregen-no-rule) nobody does
        if wl.get("install", True):               # ... and the finished arenas install their staged worlds inside the step
            cfg.regen_cap = E
    occ = world.make_maps(E, wl["size"], seed, env_index_base=base, indoor_ratio=wl.get("indoor_ratio", 0.0))
    goal = (10.0, 20.0) if wl["size"] >= 400 else (2.0, 4.0)
    arrays = world.make_world(cfg, occ, n_peds=wl["peds"], device=device, min_goal_dist=goal[0], max_goal_dist=goal[1],
                              robot_clearance=1.2 if wl["size"] >= 400 else 0.9,
                              # rect records: the march's shortcut around most field reads.  With navsim_regen every step
                              # they stay on for worlds of OUTDOOR maps (the records of a new map come out of the pass that
                              # writes its field, kernels_reset.hpp regen_maps_item: c5 4.30 -> 4.55 M env-steps/s); a world
                              # that also draws corridor maps would run the verified builder for a handful of new maps
                              # per step, which is latency-bound (~0.1 ms) and costs more than it saves
                              rect_table=wl.get("rects", (float(wl.get("indoor_ratio", 0.0)) == 0.0) if wl.get("regen") else None))
    dev = torch.device(device)
    robot = wl.get("robot", "keti")
    cfg.axle_offset = robots.ROBOTS[robot]["axle_offset"]
    for i, v in enumerate(np.asarray(robots.ROBOTS[robot]["threshold_footprint"], dtype=np.float64).reshape(-1)):
        cfg.robot_seen_footprint[i] = float(v)
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = sim.scan_threshold(cfg, torch.from_numpy(robots.footprint_array(robot, name)).to(dev))
    # a world that regenerates after every step: the restarted arenas' first observations come from navsim_regen's masked
    # launch instead of a second scan inside the step (include/navsim.h defer_reset_scan) -- where a launch is one
    # generation of workgroups (c5: 512 arenas per GPU).  --defer-reset-scan 0/1 overrides.
    if wl.get("regen") and not wl.get("pregen", False):
        want = wl.get("defer_reset_scan", -1)
        cfg.defer_reset_scan = int(want) if want in (0, 1) else int(E <= 1024)
    s = sim.NavSim(cfg, arrays, device=device)
    s.reset_obs()
    if wl.get("regen") and wl.get("pregen", False):
        s.enable_pregen(pipeline=int(wl.get("pipeline", 0)), install=bool(wl.get("pipeline", 0)) and wl.get("install", True))    # next worlds staged ahead of time on a side stream (navsim_regen_swap)
    return cfg, s, arrays, occ


def cpu_baseline(wl, seconds=15.0):
    """The CPU oracle (oracle/, kind "port": a restatement, not the reference binary -- the
    reference's own step() cannot execute) on a bounded sample of the same
    workload, all host cores, arenas split statically over threads."""
    import numpy as np
    import torch
    from nav_gym_amd import abi, lib, robots, world
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota()
    # threads = the CPU time the process may actually use: the cgroup's quota where there is one (the GPU boxes of this
    # pool show 256 CPUs and grant 16: more threads than that only take turns -- profiles/r05_cpu/scaling.txt)
    cores = max(1, min(visible, int(quota + 0.999))) if quota else visible
    E = min(wl["envs"], 8 * cores)
    cfg = lib.default_config(n_envs=E, map_h=wl["size"], map_w=wl["size"], max_peds=max(wl["peds"], 1),
                             ped_model=abi.PED_SFM if wl["peds"] else abi.PED_NONE, n_spawn=16, auto_reset=1, seed=1234)
    if wl["beams"] == 1081:
        world.lidar_1081(cfg)
    else:
        world.lidar_full_circle(cfg, wl["beams"])
    occ = world.make_maps(E, wl["size"], 1234)
    field = torch.from_numpy(ref.build_dt(occ))
    goal = (10.0, 20.0) if wl["size"] >= 400 else (2.0, 4.0)
    arrays = world.make_world(cfg, occ, n_peds=wl["peds"], device="cpu", field=field, min_goal_dist=goal[0],
                              max_goal_dist=goal[1], robot_clearance=1.2 if wl["size"] >= 400 else 0.9)
    host = {k: v.numpy() for k, v in arrays.items()}
    host["scan_threshold"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    host["scan_discomfort"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    nthr = min(cores, E)
    # the fields -- what the march reads at random -- in pages first touched by the thread that marches them (pinned, static
    # split: navsim_field_local_copy_cpu): on the GPU box's two sockets every thread then reads its own node's memory
    host["field"] = ref.field_local_copy(host["field"], nthr)
    r = ref.RefSim(cfg, host, keep=("field",))
    r.reset_obs()
    rng = np.random.default_rng(0)

    def actions(n):
        return np.stack([rng.uniform(0, 0.5, (n, E)), rng.uniform(-0.64, 0.64, (n, E))], axis=2)
    # The loop is the oracle's own (navsim_step_threads_cpu, round 5): POSIX threads inside the library, every thread owns
    # E / nthr arenas for the whole call and never waits for another (arenas are independent), no Python between steps.
    # (Round 4 dispatched every step through a Python thread pool: 256 futures per ~3 ms of native work -- 13x on 256 threads.)
    r.step_native_threads(actions(2), nthr)                      # warm-up: page the fields in, start the threads once
    t0 = time.perf_counter()
    per_call = 4
    r.step_native_threads(actions(per_call), nthr)
    one_call = max(time.perf_counter() - t0, 1e-4)
    per_call = int(max(2, min(2000, 2 * round(0.5 * per_call * 1.0 / one_call))))      # ~1 s of work per call, even
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        r.step_native_threads(actions(per_call), nthr)
        n += per_call
    dt = time.perf_counter() - t0
    # SURVEY.md 8d extras, bounded to a few seconds each: the same loop on ONE thread, and a
    # "reference-shaped" step (1 arena, 512 beams over 2*pi, 1000x1000 map, 10 pedestrians, every
    # pedestrian's own 512-beam scan computed each step as env.py:685-693 does)
    # ONE thread on one thread's share of the arenas (the same working set per core as in the all-core run)
    E1 = max(1, E // nthr)
    c1t = cfg.copy(); c1t.n_envs = E1
    h1t = {k: (v if k in ("scan_threshold", "scan_discomfort", "beam_table") else v[:E1]) for k, v in host.items()}
    r1t = ref.RefSim(c1t, h1t)
    r1t.reset_obs()
    acts1 = lambda n: np.stack([rng.uniform(0, 0.5, (n, E1)), rng.uniform(-0.64, 0.64, (n, E1))], axis=2)
    r1t.step_native_threads(acts1(2), 1)
    t1 = time.perf_counter()
    m = 0
    while time.perf_counter() - t1 < 3.0:
        r1t.step_native_threads(acts1(per_call), 1)
        m += per_call
    one_thread = E1 * m / (time.perf_counter() - t1)
    # SURVEY.md 8d work counters: distance-field probes per ray of the oracle's march (calc_range, env.py:425) on
    # this workload's arenas -- a few steps on the calling thread (the histogram is per thread)
    ref.probe_hist(reset=True)
    n_probe_steps = 0
    t3 = time.perf_counter()
    while n_probe_steps < 3 or (time.perf_counter() - t3 < 1.5 and n_probe_steps < 50):
        r.step(np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1))
        n_probe_steps += 1
    hist = ref.probe_hist(reset=True)
    rays = int(hist.sum())
    cum = np.cumsum(hist)
    probes = dict(mean=float((hist * np.arange(256)).sum() / max(rays, 1)),
                  p50=int(np.searchsorted(cum, 0.50 * rays)), p99=int(np.searchsorted(cum, 0.99 * rays)),
                  max_bin=int(np.nonzero(hist)[0].max()) if rays else 0, rays=rays,
                  sample="%d arenas x %d steps of the oracle (robot scans incl. crash re-scans)" % (E, n_probe_steps))
    ref_shaped_us = None
    try:
        c1 = lib.default_config(n_envs=1, map_h=1000, map_w=1000, max_peds=10, ped_model=abi.PED_SFM, n_spawn=4,
                                auto_reset=1, seed=77)
        world.lidar_full_circle(c1, 512)
        occ1 = world.make_maps(1, 1000, 77)
        f1 = torch.from_numpy(ref.build_dt(occ1))
        a1 = world.make_world(c1, occ1, n_peds=10, device="cpu", field=f1)
        h1 = {k: v.numpy() for k, v in a1.items()}
        h1["scan_threshold"] = ref.scan_threshold(c1, robots.footprint_array("keti", "threshold_footprint"))
        h1["scan_discomfort"] = ref.scan_threshold(c1, robots.footprint_array("keti", "discomfort_threshold_footprint"))
        r1 = ref.RefSim(c1, h1)
        r1.reset_obs()
        t2 = time.perf_counter()
        k = 0
        while time.perf_counter() - t2 < 2.0 and k < 5000:
            r1.step(np.array([[rng.uniform(0, 0.5), rng.uniform(-0.64, 0.64)]]))
            r1.ped_scans()
            k += 1
        ref_shaped_us = (time.perf_counter() - t2) / k * 1e6
    except Exception:
        pass
    c1_us = cpu_c1_us_per_step()
    return dict(value=E * n / dt, unit="env-steps/s", cores=nthr, kind="port", c1_us_per_step_1_thread=c1_us,
                sample="%d arenas x %d steps of the same workload (oracle/navsim_ref.c navsim_step_threads_cpu: %d pinned POSIX "
                       "threads, arenas split statically, fields node-local, %.1f s); value_1_thread: one thread on %d arenas"
                       % (E, n, nthr, dt, E1),
                value_1_thread=one_thread, parallel_efficiency=(E * n / dt) / max(one_thread * nthr, 1e-9),
                cpus_visible=visible, cpu_quota=quota,
                probes_per_ray=probes,
                reference_shaped_us_per_step=ref_shaped_us,
                reference_shaped="1 arena, 512 beams, 1000x1000 map, 10 pedestrians + their 512-beam scans, 1 thread "
                                 "(restatement, not the reference binary)")


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cgroup v2 cpu.max, v1 cfs quota / period), None = unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher around it: start N fresh worker processes, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set like torch.distributed.run does), BEFORE this process has
    touched torch or the GPU; relay rank 0's JSON line; fail if any rank fails."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves its peers waiting in a collective: once one has failed, the others get 30 s
    # to finish on their own and are then terminated (exact PIDs, never by pattern)
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 30.0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(30.0)                       # rank 0 has exited: its pipe is at EOF, the reader ends at once
    text = b"".join(out0).decode()
    sys.stdout.write(text)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %s" % ", ".join("rank %d exit %d" % rc for rc in bad))
    if not any(ln.startswith("{") for ln in text.splitlines()):
        raise SystemExit("bench.py: every rank exited 0 but rank 0 printed no JSON line")


def cpu_c1_us_per_step():
    """BASELINE.json configs[0] -- "1 env NavGym-v0, 64-beam lidar, 100 x 100 static map, no pedestrians, CPU reference step()":
    the oracle on ONE thread on that world, microseconds per step (the GPU side is other_workloads.c1).  Part of the CPU baseline:
    the one place besides tests/ and smoke() where bench.py runs the oracle."""
    import numpy as np
    import torch
    from nav_gym_amd import abi, lib, robots, world
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref
    rng = np.random.default_rng(7)
    try:
        w1 = WORKLOADS["c1"]
        cc = lib.default_config(n_envs=1, map_h=w1["size"], map_w=w1["size"], max_peds=1, ped_model=abi.PED_NONE, n_spawn=16,
                                auto_reset=1, seed=1234)
        world.lidar_full_circle(cc, w1["beams"])
        occ_c1 = world.make_maps(1, w1["size"], 1234)
        a_c1 = world.make_world(cc, occ_c1, n_peds=0, device="cpu", field=torch.from_numpy(ref.build_dt(occ_c1)), min_goal_dist=2.0,
                                max_goal_dist=4.0, robot_clearance=0.9)
        h_c1 = {k: v.numpy() for k, v in a_c1.items()}
        h_c1["scan_threshold"] = ref.scan_threshold(cc, robots.footprint_array("keti", "threshold_footprint"))
        h_c1["scan_discomfort"] = ref.scan_threshold(cc, robots.footprint_array("keti", "discomfort_threshold_footprint"))
        r_c1 = ref.RefSim(cc, h_c1)
        r_c1.reset_obs()
        a1 = lambda n: np.stack([rng.uniform(0, 0.5, (n, 1)), rng.uniform(-0.64, 0.64, (n, 1))], axis=2)
        r_c1.step_native_threads(a1(200), 1)
        t4 = time.perf_counter()
        k1 = 0
        while time.perf_counter() - t4 < 1.0:
            r_c1.step_native_threads(a1(2000), 1)
            k1 += 2000
        return (time.perf_counter() - t4) / k1 * 1e6
    except Exception:
        return None


def profiled_counters(workload, E, field, rects, indoor_ratio=0.0):
    """Counter figures of the step kernel from the committed PMC profile of this workload (rocprofv3 --pmc passes,
    profiles/run_profiles.sh -> traffic.json): HBM bytes per launch and the vector-issue fraction.  Counters cannot be
    collected inside the timed process, so they are quoted -- and ONLY when the profile was taken from the same
    library sources (lib.source_hash) and launch shape; otherwise (None, reason)."""
    from nav_gym_amd import lib
    reason = None
    # profiles/<round>_<workload>/ and its variants of other launch shapes (…_indoor: corridor maps only)
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        for variant in ("", "_indoor"):
            tp = os.path.join(ROOT, "profiles", "%s_%s%s" % (rnd, workload, variant), "traffic.json")
            if not os.path.exists(tp):
                continue
            try:
                t = json.load(open(tp))
            except Exception:
                continue
            src = "profiles/%s_%s%s/traffic.json@%s" % (rnd, workload, variant, t.get("commit", "?"))
            if t.get("kernel_src_sha") != lib.source_hash():
                reason = reason or "%s is of other kernel sources (%s, this build %s)" % (src, t.get("kernel_src_sha"), lib.source_hash())
                continue
            if (t.get("envs_per_gpu") != E or field != "u16t" or t.get("rect_table", True) != rects or
                    abs(float(t.get("indoor_ratio") or 0.0) - float(indoor_ratio)) > 1e-9):
                reason = reason or "%s is of another launch shape" % src
                continue
            t["source"] = src
            return t, None
        if reason:
            return None, reason
    return None, "no committed PMC profile of this workload"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs = ranks of ONE node (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override arenas per GPU (weak scaling)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="weak: the workload's per-GPU arena count on every rank; strong: its TOTAL split over the ranks; auto: "
                         "with N > 1 both from one invocation (value = strong, value_weak = weak), with N = 1 they coincide")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the short windows of the other workloads (other_workloads) and the gym-API pass (value_gym_api)")
    ap.add_argument("--total-envs", type=int, default=0, help="override the total arena count (strong scaling; may be ragged)")
    ap.add_argument("--gather", default="auto", choices=["auto", "none", "all"],
                    help="auto: with N > 1 time a second pass with the RCCL all-gather of the observation rows after every "
                         "step and report it as value_with_obs_gather; all: gather inside `value` itself; none: never")
    ap.add_argument("--indoor-ratio", type=float, default=0.0,
                    help="fraction of corridor maps (create_indoor_map) among the arenas; the rest are outdoor maps")
    ap.add_argument("--field", default="u16t", choices=["u16t", "f32"], help="distance-field storage")
    ap.add_argument("--noise-std", type=float, default=0.02,
                    help="scan_noise_std of every arena in the timed steps (SURVEY.md 8d: 0.02 for throughput runs)")
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of K steps (the first one is `value`)")
    ap.add_argument("--no-noise-off-pass", action="store_true",
                    help="skip the extra K steps timed with scan noise off (profiling runs: one kind of launch only)")
    ap.add_argument("--step-block", type=int, default=0, help="threads per arena (0 = library default)")
    ap.add_argument("--march-rule", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="navsim_config.march_rule (include/navsim.h NAVSIM_MARCH_*): -1 the library default (1 = F32)")
    ap.add_argument("--spinup-ms", type=float, default=500.0,
                    help="untimed GPU work before the warm-up steps (leaves the idle power state); 0 = none")
    ap.add_argument("--no-cold-pass", action="store_true", help="skip the extra K steps timed without spin-up (value_no_spinup)")
    ap.add_argument("--cold-idle-s", type=float, default=3.0, help="idle seconds in front of the value_no_spinup pass")
    ap.add_argument("--lpt-period", type=int, default=0, help="steps between launch-order sorts (0 = NavSim's default)")
    ap.add_argument("--ped-split", type=int, default=0, choices=[0, 1, 2],
                    help="navsim_config.ped_split: 0 library default, 1 pedestrians inside the step, 2 ped_update_kernel first")
    ap.add_argument("--rect-lds", type=int, default=0, choices=[0, 1, 2],
                    help="navsim_config.rect_lds: 0 library default (record table staged in LDS for small launches), 1 never, 2 always")
    ap.add_argument("--no-rects", action="store_true",
                    help="march through the packed field only, without the two-rectangle tile records (A/B)")
    ap.add_argument("--rects", action="store_true",
                    help="keep the tile records also in a world that regenerates its maps (c5; A/B)")
    ap.add_argument("--defer-reset-scan", type=int, default=-1, choices=[-1, 0, 1],
                    help="regenerating workloads: 1 = navsim_regen scans the restarted arenas, 0 = the step does (-1: by batch size)")
    ap.add_argument("--pregen", action="store_true",
                    help="c5: worlds staged ahead on a side stream + navsim_regen_swap instead of navsim_regen after every step "
                         "(measured: +11-14 %% at 128-256 arenas per GPU, +-0 at the 512 of c5 where the step kernel fills the chip)")
    ap.add_argument("--pregen-pipeline", type=int, default=0, metavar="P",
                    help="with --pregen: a staging pass every P steps, waited for two periods later (NavSim.enable_pregen(pipeline=P)); "
                         "sets cfg.regen_min_steps = 4 P -- episodes shorter than that restart on their old map")
    ap.add_argument("--pregen-no-rule", action="store_true",
                    help="with --pregen-pipeline: cfg.regen_min_steps = 0 -- every finished arena gets a new map, as without the pipeline; "
                         "one that finishes before its world is staged is regenerated on the spot (navsim_regen after every step)")
    ap.add_argument("--pregen-swap-kernel", action="store_true",
                    help="with --pregen-pipeline: install the staged worlds with navsim_regen_swap after the step instead of inside it (A/B)")
    ap.add_argument("--autoreset", default="same_step", choices=["same_step", "next_step"],
                    help="what ends an episode (include/navsim.h NAVSIM_AUTORESET_*): same_step = the arena restarts inside the step "
                         "that ends its episode (the bench's form since round 1); next_step = that step returns the terminal "
                         "observation and the NEXT one resets the arena (gymnasium's next-step mode)")
    ap.add_argument("--only-windows", default="", help="diagnostic: print only these windows of other_workloads (comma-separated) and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the K timed steps as one captured hipGraph: auto = where a step is several launches (c5's "
                         "navsim_regen: +6 %); a one-kernel step gains nothing from it (c2 +-0)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction control flow only, no GPU work (tests/test_host_logic.py)")
    args = ap.parse_args()

    if args.gpus is None:                   # `torchrun --nproc-per-node N bench.py`: the launcher's world is the answer
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:             # an EXPLICIT --gpus that contradicts the launcher is an error
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    # NAVSIM_BENCH_BACKEND=gloo + NAVSIM_BENCH_ONE_GPU=1: control-flow test of the N > 1 path with
    # several ranks sharing one GPU (the driver's real runs use nccl = RCCL, one rank per GPU)
    backend = os.environ.get("NAVSIM_BENCH_BACKEND", "nccl")
    if args.dry_run:
        return dry_run(args, rank, world_size, backend)

    import torch
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product path has no CPU fallback")
    if args.only_windows:
        res = extras(args, "cuda:0")
        for k, v in res["other_workloads"].items():
            keep = ("value", "ms_per_step", "kernel_ms", "regen_counters", "error", "launch", "us_per_step", "us_per_step_device", "kernel_us",
                    "cpu_us_per_step_1_thread", "cpu_kind", "workload")
            print(k, json.dumps({kk: vv for kk, vv in v.items() if kk in keep}))
        return
    # Libraries write to the process's stdout behind Python's back -- RCCL prints a version banner through C stdio, which a
    # redirected stdout holds until the process EXITS, i.e. after the result line (seen with NAVSIM_BENCH_FORCE_DIST=1 on the
    # one-GPU box).  The contract is ONE json line on stdout: file descriptor 1 is stderr for the rest of the run, and the line
    # goes to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("NAVSIM_BENCH_ONE_GPU"):
        local_rank = 0
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    # NAVSIM_BENCH_FORCE_DIST=1: a process group also at N = 1 -- the nccl (RCCL) branches below (device-side barrier and
    # MAX reduction, the obs gather) then run on a one-GPU box, where gloo dry runs never reach them
    if world_size > 1 or os.environ.get("NAVSIM_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))     # (only the forced one-rank group gets here without one)
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the communicator itself, before any kernel of ours has run: a failure up to here is the box's (no usable IPC,
            # no device for RCCL), one after the marker below is ours -- a faulting kernel of this library surfaces as an
            # ncclUnhandledCudaError at the next collective (tests/test_gpu_parity.py tells the two apart by the marker)
            probe = torch.zeros(1, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
        print("navsim-bench: process group ready (%s, %d ranks)" % (backend, dist.get_world_size()), file=sys.stderr, flush=True)
    coll_dev = device if backend == "nccl" else "cpu"

    ctx = dict(rank=rank, world_size=world_size, dist=dist, backend=backend, device=device, coll_dev=coll_dev)
    if args.scaling == "auto" and world_size > 1:
        # BOTH curves from one invocation: `value` is the strong number (the metric's fixed total), value_weak the weak one
        out = measure(args, "strong", ctx)
        weak = measure(args, "weak", ctx, light=True)
        if rank == 0:
            out["value_weak"] = weak["value"]
            out["weak"] = {k: weak[k] for k in ("value", "ms_per_step", "scaling")}
            out["weak"].update(envs_total=weak["config"]["envs_total"], envs_per_gpu=weak["config"]["envs_per_gpu"],
                               kernel_ms=weak["roofline"]["kernel_ms"], value_with_obs_gather=weak.get("value_with_obs_gather"))
            out["value_strong"] = out["value"]
    else:
        out = measure(args, "weak" if args.scaling == "auto" else args.scaling, ctx)
        if rank == 0:
            out["value_weak" if out["scaling"] == "weak" else "value_strong"] = out["value"]
    if rank == 0 and world_size == 1 and not args.no_extras and args.workload == "c2" and not args.envs:
        out.update(extras(args, device))
        c1 = out.get("other_workloads", {}).get("c1")
        if isinstance(c1, dict) and "error" not in c1:      # BASELINE.json configs[0]: both sides of "CPU reference step()"
            c1["cpu_us_per_step_1_thread"] = (out.get("cpu_baseline") or {}).get("c1_us_per_step_1_thread")
            c1["cpu_kind"] = C1_CPU_KIND
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def extras(args, device):
    """N = 1, default run only: what the driver's line would otherwise not show (round-3 verdict).  Short windows (200 steps
    after 30) of the other BASELINE workloads, each on a freshly built world, and the c2 steps through the gym API --
    NavGymEnv.step(torch actions): the action copy, the step, .bool() of the done flags, the observation dict."""
    import torch
    res = {"other_workloads": {}}
    # "c5_pipelined": c5 with the round-5 reset path -- worlds staged ahead by passes on a side stream (one every 4 steps),
    # installed inside the step's own launch (navsim_step_install), cfg.regen_min_steps = 16 (an episode shorter than that
    # restarts on its old map: the one rule the reference does not have, counted in regen_short)
    # (measured LAST, behind the gym-API windows: a process that has used a HIGH-PRIORITY HIP stream replays hipGraphs ~10 us per
    #  kernel slower from then on -- the reference-default window fell from 0.87 M to 0.59 M env-steps/s when a world with such a
    #  stream had been built and dropped before it, profiles/_diag/after_pregen.py; the pipelined passes now run on a stream
    #  of ordinary priority, the order stays)
    def workload_window(name):
        wl = dict(WORKLOADS[name.split("_")[0]]); wl["field"] = "u16t"; wl["indoor_ratio"] = 0.0
        if name.startswith("c5_pipelined"):
            wl.update(pregen=True, pipeline=4, install=True, no_rule=name.endswith("no_rule"))
        if name.startswith("c5_next_step"):
            # gymnasium's next-step auto-reset (NAVSIM_AUTORESET_NEXT_STEP): a finished arena returns its terminal observation (after
            # a crash the reference's re-scan at the reverted pose: a second scan same-step restarts without final_obs never pay)
            # and is reset by the NEXT call -- its workgroup installs the staged world at the front of the launch instead of
            # stepping; an arena that found nothing staged when its episode ended is regenerated BESIDE that launch on a third
            # stream (navsim_step_install_next); no rule (cfg.regen_min_steps = 0: a new map at every reset, like the reference)
            wl.update(pregen=True, pipeline=4, install=True, no_rule=True, next_step=True)
        try:
            cfg, sim, arrays, _ = build_sim(wl, 0, wl["envs"], device=device)
            E = cfg.n_envs
            sim.t["scan_noise_std"].fill_(args.noise_std); sim.cfg.add_scan_noise = int(args.noise_std > 0)
            g = torch.Generator(device=device); g.manual_seed(77)
            K, Wm = 200, 30
            acts = torch.rand((K + Wm, E, 2), generator=g, device=device, dtype=torch.float64)
            lin_hi, rot_hi = (1.0, 2.0) if wl.get("robot") == "husky" else (0.5, 0.64)
            acts[..., 0] *= lin_hi; acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * rot_hi
            regen = bool(wl.get("regen"))
            # one HIP event pair per step launch only where other kernels run between the steps (c5's navsim_regen): a pair per
            # step costs a few per cent of a 60 us step; elsewhere ONE pair brackets the K launches
            # (the pipelined reset path: its step launch is the whole step -- one pair around the K steps, like the one-kernel workloads)
            per_step = regen and not wl.get("pipeline")
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K if per_step else 1)]

            def one(t, e=None):
                sim.io.action = acts[t].data_ptr()
                sim._reorder()
                if e is not None:
                    e[0].record()
                sim.launch_step(reorder=False)
                if e is not None:
                    e[1].record()
                if regen:
                    sim.regen()
            for t in range(Wm):
                one(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if not per_step:
                ev[0][0].record()
            for t in range(K):
                one(Wm + t, ev[t] if per_step else None)
            if not per_step:
                ev[0][1].record()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / K
            graphed = None
            if regen and not wl.get("pipeline"):   # the regenerating loop is launch-bound (step + navsim_regen's kernels per step): the
                try:                    # same K steps again as ONE hipGraph replay -- that is the workload's `value`
                    cur0 = sim.cur
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for t in range(K):
                            one(Wm + t)
                    torch.cuda.synchronize()
                    graph.replay(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    graph.replay()
                    torch.cuda.synchronize()
                    graphed = time.perf_counter() - t0
                    del graph
                except Exception as exc:
                    sim.cur = cur0
                    torch.cuda.synchronize()
                    graphed = None
            plain = el
            if graphed is not None:
                el = graphed
            rects = "rect_table" in sim.t
            lds = rects and "rect_index" in sim.t and bool(cfg.closed_maps)
            s_map = (sim.t["rect_index"].shape[1] / float(cfg.map_h * cfg.map_w)) if lds else (0.25 if rects else 2)
            A = algorithmic_bytes_per_env_step(cfg.map_h, cfg.map_w, cfg.n_beams, cfg.n_scan_stack, wl["peds"], s_map)
            res["other_workloads"][name] = {
                "value": E * K / el, "ms_per_step": el / K * 1e3, "kernel_ms": kernel_ms, "envs_per_gpu": E, "steps": K,
                "frac": A * E / (kernel_ms * 1e-3) / 8.0e12, "s_map": s_map,
                "launch": "hipGraph replay of the K steps" if graphed is not None else "one launch per step",
                "value_plain_launches": E * K / plain,
                "workload": "%d arenas x %d beams, %dx%d maps, %d pedestrians, %s%s" % (
                    E, cfg.n_beams, cfg.map_h, cfg.map_w, wl["peds"], wl.get("robot", "keti"), ", new map per episode" if regen else ""),
                "kernel_ms_from": ("one HIP event pair per step launch (the step kernel alone; ms_per_step also holds navsim_regen)" if per_step else
                                   "one HIP event pair around the %d launches / %d (includes the gaps between launches)" % (K, K))}
            if regen:
                res["other_workloads"][name]["regen_counters"] = sim.counters()
                res["other_workloads"][name]["regen_min_steps"] = int(cfg.regen_min_steps)
            if wl.get("pipeline"):
                res["other_workloads"][name]["reset_path"] = (
                    "staging passes every %d steps on a side stream, installed inside the step (navsim_step_install, slot tables); plain launches; %s"
                    % (wl["pipeline"], "NO rule: whoever finishes before its world is staged is regenerated on the spot (the rollout of "
                       "navsim_regen after every step, bit for bit)" if wl.get("no_rule") else
                       "cfg.regen_min_steps = %d: shorter episodes restart on their old map" % int(cfg.regen_min_steps)))
                if wl.get("next_step"):
                    res["other_workloads"][name]["autoreset"] = ("next step (NAVSIM_AUTORESET_NEXT_STEP): the call that ends an episode returns the "
                                                                 "terminal observation, the next call resets the arena (counted as a step of it)")
            del sim, arrays
            torch.cuda.empty_cache()
        except Exception as exc:                              # an extra must not cost the run its line
            res["other_workloads"][name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    only = [n for n in (getattr(args, "only_windows", None) or "").split(",") if n]   # diagnostic: just these windows of other_workloads
    for name in (only or ("c3", "c4", "c5")):
        if name != "c1":
            workload_window(name)
    if only and "c1" not in only:
        return res
    # BASELINE.json configs[0]: ONE arena, 64 beams, 100 x 100 map, no pedestrians -- a launch of one 64-thread workgroup;
    # us per step as the caller sees it (plain launches back to back) and the kernel alone (one event pair around the launches)
    try:
        wl1 = dict(WORKLOADS["c1"]); wl1["field"] = "u16t"; wl1["indoor_ratio"] = 0.0
        cfg1, sim1, arrays1, _ = build_sim(wl1, 0, 1, device=device)
        sim1.t["scan_noise_std"].fill_(args.noise_std); sim1.cfg.add_scan_noise = int(args.noise_std > 0)
        g1 = torch.Generator(device=device); g1.manual_seed(79)
        K1, W1 = 2000, 200
        acts1 = torch.rand((K1 + W1, 1, 2), generator=g1, device=device, dtype=torch.float64)
        acts1[..., 0] *= 0.5; acts1[..., 1] = (acts1[..., 1] * 2.0 - 1.0) * 0.64
        def one1(t):
            sim1.io.action = acts1[t].data_ptr()
            sim1.launch_step(reorder=False)
        for t in range(W1):
            one1(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for t in range(K1):
            one1(W1 + t)
        e1.record()
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t0
        graph1 = torch.cuda.CUDAGraph()                     # the kernel alone: the same launches replayed without host gaps
        cur0 = sim1.cur
        with torch.cuda.graph(graph1):
            for t in range(200):
                one1(W1 + t)
        torch.cuda.synchronize()
        graph1.replay(); torch.cuda.synchronize()
        g0, g1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record(); graph1.replay(); g1e.record()
        torch.cuda.synchronize()
        res["other_workloads"]["c1"] = {
            "us_per_step": el1 / K1 * 1e6, "us_per_step_device": e0.elapsed_time(e1) / K1 * 1e3,
            "kernel_us": g0.elapsed_time(g1e) / 200 * 1e3, "value": K1 / el1, "steps": K1,
            "workload": "1 arena x 64 beams, 100x100 map, no pedestrians (BASELINE.json configs[0])",
            "from": "us_per_step: wall clock of %d plain launches; us_per_step_device: one HIP event pair around them; kernel_us: 200 "
                    "of the launches replayed as one hipGraph (no host gaps) / 200" % K1}
        del graph1, sim1, arrays1
        torch.cuda.empty_cache()
    except Exception as exc:
        res["other_workloads"]["c1"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if only:
        c1 = res["other_workloads"]["c1"]
        if "error" not in c1:
            c1["cpu_us_per_step_1_thread"] = cpu_c1_us_per_step()
            c1["cpu_kind"] = C1_CPU_KIND
        return res
    def gym_window(E, K=200, Wm=30, **kw):
        """K calls of NavGymEnv.step(torch float64 actions [E,2]) after reset() on the device -> dict(value, ms_per_step, envs,
        steps, reset_first_ms, reset_steady_ms).  reset_first_ms: the first reset() of a new environment (allocations, the
        library's first launches, graph capture where the env uses graphs); reset_steady_ms: a second reset() of the same
        environment -- what an RL loop pays per reset of the whole batch (round-4 verdict: 689 ms vs 19 ms were these two)."""
        import nav_gym_env
        env = nav_gym_env.make("NavGym-v0", num_envs=E, device=device, seed=1234, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.reset()                                # maps, fields, records, costmaps, planned starts / goals / routes, first observations
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        env.reset()
        torch.cuda.synchronize()
        steady = time.perf_counter() - t0
        g = torch.Generator(device=device); g.manual_seed(78)
        acts = torch.rand((K + Wm, E, 2), generator=g, device=device, dtype=torch.float64)
        acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
        for t in range(Wm):
            env.step(acts[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(K):
            obs, rew, done, info = env.step(acts[Wm + t])
        host = time.perf_counter() - t0            # the loop returns before the GPU has finished: what the host needs per step
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        counters = env.counters()
        episodes = int(env.sim.t["episode"].sum().item())
        env.close()
        del env
        torch.cuda.empty_cache()
        return {"value": E * K / el, "ms_per_step": el / K * 1e3, "host_ms_per_step": host / K * 1e3, "envs": E, "steps": K, "reset_first_ms": first * 1e3,
                "reset_steady_ms": steady * 1e3, "episodes_started": episodes, "counters": counters}
    try:
        sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
        wl = WORKLOADS["c2"]
        c2_kw = dict(n_beams=wl["beams"], map_size=wl["size"], indoor_ratio=0.0)
        w = gym_window(wl["envs"], pedestrian_model="none", num_humans=0, **c2_kw)
        res["value_gym_api"] = w["value"]
        res["gym_api"] = {"ms_per_step": w["ms_per_step"], "envs": w["envs"], "steps": w["steps"],
                          "reset_first_ms": w["reset_first_ms"], "reset_steady_ms": w["reset_steady_ms"],
                          "what": "K calls of NavGymEnv.step(torch float64 actions [E,2]) on a c2-shaped world made by gym.make('NavGym-v0', "
                                  "num_envs=4096, n_beams=1081, map_size=500, pedestrian_model='none', indoor_ratio=0) + reset() on the device; "
                                  "returns the obs dict, reward, done.bool(), info (env.py:591-728's signature).  reset_first_ms: the first reset() "
                                  "of a new environment (allocation, first launches); reset_steady_ms: a second reset() of all arenas"}
        # the same API on the c3-shaped world: 20 social-force pedestrians per arena, whose routes are planned on the costmap and
        # re-planned at their goals by navsim_replan (plan_paths=True, the env's default: since round 5 beside the step,
        # navsim_step_part), and without
        for key, plan in (("plan_paths", True), ("no_plan_paths", False)):
            w = gym_window(wl["envs"], pedestrian_model="sfm", num_humans=20, plan_paths=plan, **c2_kw)
            res["gym_api"]["c3_world_" + key] = {k: w[k] for k in ("value", "ms_per_step", "reset_first_ms", "reset_steady_ms", "counters")}
    except Exception as exc:
        res.setdefault("gym_api", {})["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
    # The reference's OWN configuration (round-4 verdict: "the only configuration an hrl-nav user runs unmodified"): every
    # registered default of NavGym-v0 (__init__.py:4-40) -- 512 beams over 2 pi (keti_robot.py:44-48), 1000 x 1000 corridor maps
    # and 400 x 400 outdoor maps at indoor_ratio 0.5 (map_generator.py:97-143), 5-15 pedestrians on planned routes, scan
    # noise 0-0.05, a NEW map at every episode end -- batched: E arenas per GPU.
    try:
        ref_def = {}
        for model, E in (("sfm", 1024), ("sfm", 4096)):
            # (pregen_pipeline=0: navsim_regen after every step -- the form of rounds 3-5; the env's default for this world is
            #  the pipelined reset path, same rollout, measured LAST below as sfm_*_pregen_pipeline_4)
            w = gym_window(E, K=100, Wm=20, map_size="reference", randomize_maps=True, pedestrian_model=model, pregen_pipeline=0)
            ref_def["%s_%d" % (model, E)] = w
        res.setdefault("gym_api", {})["reference_defaults"] = dict(
            ref_def, what="gym.make('NavGym-v0', num_envs=E, map_size='reference', randomize_maps=True) and nothing else changed: the "
                          "registered kwargs of __init__.py:4-40 (indoor_ratio 0.5, 5-15 pedestrians, planned routes, scan noise), "
                          "KetiRobot's 512-beam lidar, 1000 x 1000 arenas (corridor maps fill them, outdoor maps use 400 x 400); "
                          "pedestrians: build-defined social force ('policy' needs human_policy.pth, missing upstream).  "
                          "sfm_E: with pregen_pipeline=0 (navsim_regen after every step); sfm_1024_env_default: NO kwarg beyond map_size and "
                          "randomize_maps -- the env's own default for this world is pregen_pipeline=8: worlds staged ahead on a side stream, "
                          "installed inside the step, no rule: the same rollout bit for bit; sfm_E_pregen_pipeline_4: a pass every 4 steps; "
                          "..._min_steps_16: with the rule that drops the fallback (shorter episodes keep their map)")
    except Exception as exc:
        res.setdefault("gym_api", {})["reference_defaults"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    # the round-5 reset path, last (see above): the reference-default world with the ONE kwarg that switches it on -- the worlds
    # staged ahead, installed inside the step, whoever finishes before its world is staged regenerated on the spot: the same
    # rollout as sfm_1024, bit for bit -- and with the rule that drops that fallback (an episode shorter than 16 steps restarts on
    # its old map: counters regen_short)
    try:
        for key, kw in (("sfm_1024_env_default", dict()),          # nothing but map_size and randomize_maps: the env picks pregen_pipeline = 8
                        ("sfm_1024_pregen_pipeline_4", dict(pregen_pipeline=4)),
                        ("sfm_4096_pregen_pipeline_4", dict(pregen_pipeline=4)),
                        ("sfm_1024_pregen_pipeline_4_min_steps_16", dict(pregen_pipeline=4, regen_min_steps=16))):
            w = gym_window(4096 if "4096" in key else 1024, K=100, Wm=20, map_size="reference", randomize_maps=True, pedestrian_model="sfm", **kw)
            if isinstance(res.get("gym_api", {}).get("reference_defaults"), dict):
                res["gym_api"]["reference_defaults"][key] = w
    except Exception as exc:
        res.setdefault("gym_api", {})["pregen_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
    workload_window("c5_pipelined")
    workload_window("c5_pipelined_no_rule")
    workload_window("c5_next_step_reset")
    return res


def measure(args, scaling, ctx, light=False):
    """One measurement of args.workload under `scaling`: W warm-up steps, EXACTLY K timed steps between barrier +
    synchronize, max over ranks -> the line (rank 0) / None.  light: the timed steps only (no repeats, no extra passes)."""
    import torch
    rank, world_size, dist, backend, device, coll_dev = (ctx[k] for k in ("rank", "world_size", "dist", "backend", "device", "coll_dev"))
    wl = dict(WORKLOADS[args.workload])
    if args.envs:
        wl["envs"] = args.envs
    if args.total_envs:
        wl["total"] = args.total_envs
    wl["field"] = args.field
    wl["indoor_ratio"] = args.indoor_ratio
    if args.no_rects:
        wl["rects"] = False
    if args.rects:
        wl["rects"] = True
    wl["pregen"] = bool(args.pregen or args.pregen_pipeline)
    wl["pipeline"] = int(args.pregen_pipeline)
    wl["install"] = not args.pregen_swap_kernel
    wl["no_rule"] = bool(args.pregen_no_rule)
    wl["next_step"] = args.autoreset == "next_step"
    wl["defer_reset_scan"] = args.defer_reset_scan
    base, E_local = shard_of(wl, scaling, rank, world_size)
    E_total = wl["total"] if scaling == "strong" else world_size * wl["envs"]
    if E_local < 1:
        raise SystemExit("bench.py: rank %d owns no arena (%d arenas over %d ranks)" % (rank, E_total, world_size))
    cfg, sim, arrays, _ = build_sim(wl, base, E_local, device=device)
    if args.step_block:
        sim.cfg.step_block = args.step_block
    if args.ped_split:
        sim.cfg.ped_split = args.ped_split
    sim.cfg.rect_lds = args.rect_lds
    if args.march_rule >= 0:
        sim.cfg.march_rule = args.march_rule
    if args.lpt_period:
        sim.lpt_period = args.lpt_period
    E, K, Wm = cfg.n_envs, args.steps, args.warmup
    # scan noise of the timed steps (env.py:437-440): every arena at --noise-std, counter-based Gaussian per beam
    sim.t["scan_noise_std"].fill_(args.noise_std)
    sim.cfg.add_scan_noise = int(args.noise_std > 0)

    # pre-generated in-range actions, resident in HBM; step t reads slice t (no copies in the loop)
    g = torch.Generator(device=device)
    g.manual_seed(1000 + rank)
    acts = torch.rand((K + Wm, E, 2), generator=g, device=device, dtype=torch.float64)
    acts[..., 0] *= 0.5
    acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    # the optional obs gather: rows of all ranks in global arena order, one RCCL call per step
    gatherer = None
    if args.gather != "none" and dist is not None and backend == "nccl":
        from nav_gym_amd.sharding import RowGather
        if scaling == "strong":
            gatherer = RowGather(E_total, sim.obs.shape[1:], torch.float32, device, rank, world_size)
        else:                               # weak: equal shards by construction
            gatherer = RowGather(world_size * E, sim.obs.shape[1:], torch.float32, device, rank, world_size)
    gather_on = [args.gather == "all" and gatherer is not None]

    regen = bool(wl.get("regen"))
    lin_hi, rot_hi = (1.0, 2.0) if wl.get("robot") == "husky" else (0.5, 0.64)
    acts[..., 0] *= lin_hi / 0.5
    acts[..., 1] *= rot_hi / 0.64

    def run(t, ev=None):
        sim.io.action = acts[t].data_ptr()
        sim._reorder()                  # longest-first launch order, re-sorted every few steps (inside the timed region)
        if ev is not None:
            ev[0].record()              # HIP events bracket the step launch itself
        sim.launch_step(reorder=False)
        if ev is not None:
            ev[1].record()
        if regen:                       # finished arenas restart on a freshly generated map, on the device
            sim.regen()
        if gather_on[0]:
            gatherer.run(sim.obs)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    # HIP events on the launch stream.  Where the step launch is the only kernel of a step (c1-c4) ONE pair brackets
    # the K launches and kernel_ms = that span / K: an upper bound of the kernel's duration (it includes the 1-2 us
    # between consecutive launches), and nothing is inserted between the launches that are being timed -- an event
    # pair per step costs 4 % of the throughput at 120 us per step.  Where other kernels run between the steps
    # (navsim_regen, the obs gather) every step launch gets its own pair.
    def timed():
        """EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        events_per_step = (regen or gather_on[0]) and graph is None
        fence()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(K if events_per_step else 1)]
        t0 = time.perf_counter()
        if not events_per_step:
            ev[0][0].record()
        if graph is not None:
            graph.replay()
        else:
            for t in range(K):
                run(Wm + t, ev[t] if events_per_step else None)
        if not events_per_step:
            ev[0][1].record()
        fence()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, sum(a.elapsed_time(b) for a, b in ev) / K

    # 25 steps of 0.12 ms are 3 ms of GPU work: after the host-side set-up the GPU is still in its idle power state and
    # the kernel runs 5 % slower than in a long run (measured: kernel 121 vs 116 us).  Half a second of untimed work
    # that touches no simulator state (the library's device sincos on a scratch tensor) precedes the warm-up steps.
    if args.spinup_ms > 0:
        from nav_gym_amd import sim as _simmod
        scratch = torch.rand(1 << 22, device=device, dtype=torch.float64)
        t_end = time.perf_counter() + args.spinup_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                _simmod.debug_math(0, scratch)
            torch.cuda.synchronize()
        del scratch
    for t in range(Wm):
        run(t)
    fence()
    # The K timed steps are launch-bound between kernels (~5 us of host gap per 220 us kernel): optionally capture
    # them once as a hipGraph and replay it.  Every node keeps its own action slice and observation buffers.
    piped = bool(getattr(sim, "pg_period", 0))       # pipelined staging passes: fork / join inside the capture (NavSim.pregen_join)
    if args.graph == "on" and getattr(sim, "pregen", False) and not piped:
        raise SystemExit("bench: --pregen without --pregen-pipeline waits for a pass of the previous step on the host's side of the "
                         "capture; it does not go into a hipGraph")
    # (auto: not for the pipelined passes -- as fork / join nodes of one graph they run at 3.5-5.1 M env-steps/s against 8.1 M
    #  with plain launches, profiles/r05_pipe/README.md; --graph on still captures them)
    if (args.graph == "on" or (args.graph == "auto" and regen and not getattr(sim, "pregen", False))) and gatherer is None:
        try:
            cur0 = sim.cur
            if piped:
                sim.pregen_sync()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for t in range(K):
                    run(Wm + t)
                if piped:
                    sim.pregen_join()
        except Exception as exc:                      # capture unsupported here: plain launches
            if args.graph == "on":
                raise
            graph = None
            sim.cur = cur0
            torch.cuda.synchronize()
            if rank == 0:
                print("bench: hipGraph capture failed (%s); timing plain launches" % type(exc).__name__, file=sys.stderr)

    elapsed, kernel_ms = timed()                     # THE measurement (`value`)
    more = [timed() for _ in range(max(args.repeats - 1, 0))] if not light else []
    with_gather, gather_error = None, None
    if gatherer is not None and not gather_on[0]:    # the same K steps, every step followed by the obs all-gather
        gather_on[0] = True
        try:
            for t in range(min(Wm, 3)):
                run(t)
            with_gather = timed()
        except Exception as exc:                     # the extra pass must not cost the run its line
            gather_error = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        gather_on[0] = False
    noise_off = None
    if args.noise_std > 0 and not args.no_noise_off_pass and not light:   # the same K steps without the per-beam Gaussian, beside it
        sim.cfg.add_scan_noise = 0
        noise_off = timed()
        sim.cfg.add_scan_noise = 1
    # the same K steps WITHOUT the spin-up, GPU in its idle power state, reported beside `value` as
    # value_no_spinup.  Taken LAST, after a few seconds of idle (round 3: taken first, it moved the simulation and its
    # launch-order schedule 25 steps on and cost the first timed window -- the one `value` is -- 1-3 %,
    # profiles/r03_lpt/driver_shape_first_repeat.txt), with W untimed steps in front like `value`.
    cold = None
    if args.spinup_ms > 0 and not args.no_cold_pass and not light:
        fence()
        time.sleep(args.cold_idle_s)
        for t in range(Wm):
            run(t)
        cold = timed()

    if rank == 0:
        import statistics
        from nav_gym_amd import abi, lib as _lib
        n_done = int(sim.t["episode"].sum().item())
        rects = "rect_table" in sim.t
        # A = H*W*s_map + 4B + 4(S*B + 11) + 96 (+ 96 N), s_map = "bytes per cell of the map representation
        # the kernel streams once per env-step".  What this kernel reads of a map: with rect records 16 B per 8x8-cell
        # tile = 0.25 B per cell (most probes never touch the field); the packed uint16 field alone 2; float32 4.
        lds_rows = rects and "rect_index" in sim.t and bool(sim.cfg.closed_maps) and sim.cfg.rect_lds != 1
        if lds_rows:
            row_bytes = int(sim.t["rect_index"].shape[1])
            s_map = row_bytes / float(cfg.map_h * cfg.map_w)
            s_map_of = ("index form of the two-rectangle tile records, copied to LDS once per arena-step: 256 rectangles x 8 B + 2 B per "
                        "8x8 cells = %d B per arena (nav-gym_amd/csrc/kernels_rect.hpp)" % row_bytes)
        elif rects:
            s_map, s_map_of = 0.25, "two-rectangle tile records: 16 B per 8x8 cells (nav-gym_amd/csrc/kernels_rect.hpp)"
        elif cfg.field_format == abi.FIELD_U16T:
            s_map, s_map_of = 2, "packed uint16 squared-distance field in 8x8 tiles"
        else:
            s_map, s_map_of = 4, "float32 distance field (what range_libc holds)"
        H, W, B, S = cfg.map_h, cfg.map_w, cfg.n_beams, cfg.n_scan_stack
        A = algorithmic_bytes_per_env_step(H, W, B, S, wl["peds"], s_map)
        A1 = algorithmic_bytes_per_env_step(H, W, B, S, wl["peds"], 1)
        achieved = A * E / (kernel_ms * 1e-3) / 1e9                # GB/s over the kernel's own duration
        frac = achieved / 8000.0
        prof, why_not = profiled_counters(args.workload, E, args.field, rects, args.indoor_ratio)
        tbytes = prof.get("hbm_bytes_per_launch") if prof else None
        all_values = [E_total * K / el for el, _ in [(elapsed, kernel_ms)] + more]
        value = E_total * K / elapsed
        tile_frac = None
        if rects:                           # tiles whose record reproduces the field (the others fall back to it)
            tile_frac = float(((sim.t["rect_table"][..., 0] & 0xFFFF) != 0x7FFF).double().mean().item())
        out = {
            "metric": "env steps/sec (whole node), 4096 envs x 1081-beam lidar",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world_size,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d arenas (%s scaling: %d on rank 0) x %d-beam lidar, %dx%d per-arena occupancy maps "
                            "(%s distance field%s), indoor ratio %.2g, %d pedestrians/arena, %s kinematics, %s, "
                            "scan_noise_std %.3g"
                            % (args.workload, E_total, scaling, E, B, H, W, args.field,
                               " + rect records" if rects else "", args.indoor_ratio, wl["peds"], wl.get("robot", "keti"),
                               ("new random map per episode (%s)" % (("worlds staged ahead, installed inside the step (navsim_step_install)" if getattr(sim, "pg_install", False) else "worlds staged ahead, navsim_regen_swap") if getattr(sim, "pregen", False) else "navsim_regen")) if regen else "auto-respawn in place",
                               args.noise_std),
                "envs_total": E_total, "envs_per_gpu": E, "n_beams": B, "map": [H, W],
                "pedestrians": wl["peds"], "obs_gather": args.gather, "episodes_finished_rank0": n_done,
                "regen_counters_rank0": (sim.counters() if regen else None),
                "regen_min_steps": int(cfg.regen_min_steps), "pregen_pipeline": int(getattr(sim, "pg_period", 0) or 0),
                "launch": "hipGraph replay of the K steps" if graph is not None else "one launch per step",
                "ranks": world_size, "collective_backend": (backend if world_size > 1 else None),
                "scan_noise_std": args.noise_std, "rect_table": rects, "rect_valid_tile_frac": tile_frac,
                "indoor_ratio": args.indoor_ratio,
                "gpu_spinup_ms": args.spinup_ms,     # untimed, before the warm-up steps, touches no simulator state
                "kernel_src_sha": _lib.source_hash(),
            },
            "repeats": {"n": len(all_values), "values": all_values, "median": statistics.median(all_values)},
            # without the spin-up: the K steps straight after set-up, GPU still in its idle power state
            "value_no_spinup": (E_total * K / cold[0]) if cold else None,
            "roofline": {
                # The contract's fraction: SURVEY.md 8d algorithmic bytes with the s_map of what the kernel reads, over
                # the kernel's own duration, against the HBM peak.  It is reported as asked; what actually limits the
                # kernel is vector issue on the dependent probe chain (issue_frac_profiled, profiles/README.md).
                "bound": "valu-issue (profiled); hbm frac reported per contract",
                # the same duration against the bytes of the representations the march could stream instead (each lossless, each
                # measured in an earlier round): what compressing the map bought, and why `frac` FELL while `value` ROSE
                "frac_by_representation": {
                    "index_rows_in_lds (round 4)": algorithmic_bytes_per_env_step(H, W, B, S, wl["peds"], (2048 + (2 * ((H + 7) // 8) * ((W + 7) // 8) + 15) // 16 * 16) / float(H * W)) * E / (kernel_ms * 1e-3) / 8.0e12,
                    "rect_records_16B_per_tile (rounds 2-3)": algorithmic_bytes_per_env_step(H, W, B, S, wl["peds"], 0.25) * E / (kernel_ms * 1e-3) / 8.0e12,
                    "packed_u16_field (round 1)": algorithmic_bytes_per_env_step(H, W, B, S, wl["peds"], 2) * E / (kernel_ms * 1e-3) / 8.0e12,
                    "occupancy_int8 (SURVEY 8d, s_map = 1)": A1 * E / (kernel_ms * 1e-3) / 8.0e12},
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": frac,
                "s_map": s_map, "s_map_of": s_map_of, "algorithmic_bytes_per_env_step": A,
                "algorithmic_bytes_per_launch": A * E,
                # the same bytes over the step's wall time (launch gaps, sort, regen included): never above `frac`
                "frac_of_ms_per_step": A * E_total / world_size / (elapsed / K) / 8.0e12,
                # HBM bytes per launch from the PMC counters of the committed profile of THESE sources (null otherwise)
                "traffic": tbytes,
                "traffic_source": (prof["source"] if prof else None), "traffic_unavailable": why_not,
                "hbm_frac_measured": (tbytes / (kernel_ms * 1e-3) / 8.0e12) if tbytes else None,
                "traffic_over_algorithmic": (tbytes / (A * E)) if tbytes else None,
                # SQ_ACTIVE_INST_VALU x 4 cycles / (SIMDs x kernel cycles), same profile
                "issue_frac_profiled": (prof.get("valu_issue_frac") if prof else None),
                "kernel": "navsim_step_kernel", "kernel_ms": kernel_ms,
                "kernel_ms_from": ("one HIP event pair per step launch" if (regen or (args.gather == "all" and gatherer is not None)) and graph is None else
                                   "one HIP event pair around the %d launches / %d (includes the gaps between launches)" % (K, K)),
                # SURVEY.md 8d also prints the formula with s_map = 1 (the occupancy grid as the reference stores it, int8,
                # once per arena-step).  This kernel does not read the grid, so that figure is not a bound on it:
                "frac_s_map_1": A1 * E / (kernel_ms * 1e-3) / 8.0e12, "algorithmic_bytes_s_map_1": A1,
                "note_s_map_1": "not a roofline for this kernel: it counts H*W bytes per arena-step that are never read "
                                "(the march reads a lossless 4x..8x smaller description, and only along the rays)",
            },
            # SURVEY.md 8d derived work counters
            "work": {
                "rays_per_s": value * B,
                "worst_case_probes_per_env_step": int(B * cfg.range_max / cfg.resolution),
            },
        }
        if frac > 1.0:
            out["roofline"]["note"] = ("algorithmic bytes / kernel time exceeds the HBM peak: on this workload even the s_map of "
                                       "the streamed representation over-counts (rays touch a fraction of the tiles); read "
                                       "hbm_frac_measured")
        if with_gather is not None or gather_on[0]:
            el_g = with_gather[0] if with_gather is not None else elapsed
            out["value_with_obs_gather"] = E_total * K / el_g
            out["obs_gather"] = {"ms_per_step": el_g / K * 1e3, "bytes_per_rank": int(sim.obs.numel() * 4),
                                 "bytes_total": int(gatherer.out.numel() * 4), "equal_shards": gatherer.equal,
                                 "collective": "all_gather_into_tensor (RCCL)" if gatherer.equal else "all_gather of padded rows (RCCL)"}
        elif world_size > 1:
            out["value_with_obs_gather"] = None
            if gather_error:
                out["obs_gather"] = {"error": gather_error}
        if noise_off is not None:
            out["noise_off"] = {"value": E_total * K / noise_off[0], "ms_per_step": noise_off[0] / K * 1e3,
                                "kernel_ms": noise_off[1]}
        if not args.no_cpu_baseline and world_size == 1 and not light:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            pr = out["cpu_baseline"].pop("probes_per_ray")
            out["work"].update(probes_per_ray_mean=pr["mean"], probes_per_ray_p50=pr["p50"], probes_per_ray_p99=pr["p99"],
                               probes_per_ray_from="oracle/navsim_ref.c trace_ray (calc_range, env.py:425): " + pr["sample"])
        # The bound the kernel actually has (round-4 verdict): vector issue.  From the committed PMC profile of THESE sources:
        # how busy the vector units are, how many vector instructions an arena-step issues, and how many of them an ideal
        # march would need -- every ray's probes (oracle histogram) x the probe round's instructions, 64 rays per instruction.
        valu = {"probe_round_valu_insts": PROBE_ROUND_VALU, "probe_round_from": PROBE_ROUND_FROM}
        if prof:
            cnt = prof.get("counters", {})
            valu["issue_frac"] = prof.get("valu_issue_frac")
            if cnt.get("SQ_INSTS_VALU"):
                valu["valu_insts_per_env_step"] = cnt["SQ_INSTS_VALU"] / float(E)
            if cnt.get("SQ_THREAD_CYCLES_VALU") and cnt.get("SQ_ACTIVE_INST_VALU"):
                # lanes the exec mask leaves on, per vector instruction (the march's own predication is finer: next figure)
                valu["exec_lane_frac"] = cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_ACTIVE_INST_VALU"] * 64.0)
            valu["source"] = prof["source"]
        else:
            valu["unavailable"] = why_not
        mean_probes = out["work"].get("probes_per_ray_mean")
        if mean_probes:
            ideal = B * mean_probes * PROBE_ROUND_VALU / 64.0
            valu["ideal_march_insts_per_env_step"] = ideal
            if valu.get("valu_insts_per_env_step"):
                valu["ideal_march_frac"] = ideal / valu["valu_insts_per_env_step"]
                # the rest: lanes idle inside probe rounds (a 64-beam chunk marches until its slowest ray), beam directions,
                # noise, second scans after a crash, observation packing
        out["roofline"]["valu"] = valu
        del sim
        torch.cuda.empty_cache()
        return out
    del sim
    return None


def dry_run(args, rank, world_size, backend):
    """Launcher / rendezvous / max-over-ranks / sharding control flow without a GPU (CPU test of `--gpus N`): every
    rank joins the process group, works out its shard of the workload exactly as the real run does, contributes a fake
    per-rank time and -- like the obs gather -- one row per owned arena holding that arena's GLOBAL index through
    sharding.RowGather; rank 0 prints the line shape.  Measures nothing."""
    dist = None
    elapsed = 1.0 + rank
    wl = dict(WORKLOADS[args.workload])
    if args.envs:
        wl["envs"] = args.envs
    if args.total_envs:
        wl["total"] = args.total_envs
    both = args.scaling == "auto" and world_size > 1            # value = strong, value_weak = weak, one invocation
    scaling = "strong" if both else ("weak" if args.scaling == "auto" else args.scaling)
    base, count = shard_of(wl, scaling, rank, world_size)
    E_total = wl["total"] if scaling == "strong" else world_size * wl["envs"]
    gathered_ok = None
    if world_size > 1:
        import torch
        import torch.distributed as dist
        from nav_gym_amd.sharding import RowGather
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if backend == "nccl" else backend)
        assert dist.get_world_size() == args.gpus
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if args.gather != "none":
            g = RowGather(E_total, (3,), torch.float32, "cpu", rank, world_size)
            rows = (torch.arange(base, base + count, dtype=torch.float32)[:, None] * torch.ones(3)).contiguous()
            got = g.run(rows)
            gathered_ok = bool(torch.equal(got[:, 0], torch.arange(E_total, dtype=torch.float32)))
    if os.environ.get("NAVSIM_BENCH_FAIL_RANK") == str(rank):      # failure-propagation test
        raise SystemExit(3)
    if rank == 0:
        from nav_gym_amd.sharding import shard_range
        shards = [list(shard_of(wl, scaling, r, world_size)) for r in range(world_size)]
        line = {"metric": "dry-run (no GPU work, control flow only)", "value": None, "n_gpus": world_size,
                "steps": args.steps, "warmup": args.warmup, "max_rank_time": elapsed, "dry_run": True,
                "scaling": scaling, "envs_total": E_total, "shards": shards, "gather_in_global_order": gathered_ok}
        line["value_strong" if scaling == "strong" else "value_weak"] = None
        if both:                                                    # the second pass of the real run: the weak curve
            line["value_weak"] = None
            line["weak"] = {"scaling": "weak", "envs_total": world_size * wl["envs"], "envs_per_gpu": wl["envs"],
                            "shards": [list(shard_of(wl, "weak", r, world_size)) for r in range(world_size)]}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
