#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused MI355X NavGym step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1: BASELINE.json configs[1] -- 4096 arenas, 1081-beam lidar, 500x500 static occupancy map,
diff-drive (KetiRobot kinematics), no pedestrians, auto-respawn of finished arenas in place.
N > 1: one rank per GPU -- either launched by torch.distributed.run (WORLD_SIZE set), or, when called plainly
as `python bench.py --gpus N`, this script spawns the N ranks itself before touching the GPU.  Every rank owns
its own 4096 arenas (weak scaling, arenas keyed by global env index); the step has no exchange, so there is
no data-path collective (`--gather all` adds the optional RCCL all_gather of observations).

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

WORKLOADS = {
    # name: (envs per GPU, beams, map size, pedestrians, ped model)
    "c1": dict(envs=1, beams=64, size=100, peds=0),
    "c2": dict(envs=4096, beams=1081, size=500, peds=0),
    "c3": dict(envs=4096, beams=1081, size=500, peds=20),
    "c4": dict(envs=2048, beams=1081, size=1000, peds=0),      # 16384 arenas over 8 GPUs
    # 4096 arenas over 8 GPUs, Husky, 20 pedestrians, a NEW random map at every episode end (navsim_regen)
    "c5": dict(envs=512, beams=1081, size=500, peds=20, robot="husky", regen=True),
}


def algorithmic_bytes_per_env_step(H, W, B, S, n_peds, s_map):
    """map streamed once + ranges written + packed obs written + scalar state."""
    return H * W * s_map + 4 * B + 4 * (S * B + 7 + 2 + 2) + 96 + 96 * n_peds


def build_sim(wl, rank, world_size, seed=1234, device="cuda:0"):
    import numpy as np
    import torch
    from nav_gym_amd import abi, lib, robots, sim, world
    E = wl["envs"]
    cfg = lib.default_config(n_envs=E, map_h=wl["size"], map_w=wl["size"], max_peds=max(wl["peds"], 1),
                             ped_model=abi.PED_SFM if wl["peds"] else abi.PED_NONE,
                             n_spawn=16, auto_reset=1, seed=seed, env_index_base=rank * E,
                             field_format={"f32": abi.FIELD_F32}.get(wl.get("field"), abi.FIELD_U16T))
    if wl["beams"] == 1081:
        world.lidar_1081(cfg)
    else:
        world.lidar_full_circle(cfg, wl["beams"])
    if wl.get("regen"):
        cfg.regen_cap = max(16, E // 16)          # arenas regenerated per step at most (c5: ~5 finish per step)
    occ = world.make_maps(E, wl["size"], seed, env_index_base=rank * E)
    goal = (10.0, 20.0) if wl["size"] >= 400 else (2.0, 4.0)
    arrays = world.make_world(cfg, occ, n_peds=wl["peds"], device=device, min_goal_dist=goal[0], max_goal_dist=goal[1],
                              robot_clearance=1.2 if wl["size"] >= 400 else 0.9,
                              # rect records: the march's shortcut around most field reads.  Not with navsim_regen every
                              # step: rebuilding the records of a handful of new maps is latency-bound (~0.1 ms per
                              # step) and costs more than it saves there (c5: 2.14 M env-steps/s without, 1.54 M with)
                              rect_table=wl.get("rects", False if wl.get("regen") else None))
    dev = torch.device(device)
    robot = wl.get("robot", "keti")
    cfg.axle_offset = robots.ROBOTS[robot]["axle_offset"]
    for i, v in enumerate(np.asarray(robots.ROBOTS[robot]["threshold_footprint"], dtype=np.float64).reshape(-1)):
        cfg.robot_seen_footprint[i] = float(v)
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = sim.scan_threshold(cfg, torch.from_numpy(robots.footprint_array(robot, name)).to(dev))
    s = sim.NavSim(cfg, arrays, device=device)
    s.reset_obs()
    return cfg, s, arrays, occ


def cpu_baseline(wl, seconds=15.0):
    """The CPU oracle (oracle/, kind "port": a restatement, not the reference binary -- the
    reference's own step() cannot execute) on a bounded sample of the same
    workload, all host cores, arenas split statically over threads."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from nav_gym_amd import abi, lib, robots, world
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref
    cores = os.cpu_count() or 1
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
    r = ref.RefSim(cfg, host)
    r.reset_obs()
    rng = np.random.default_rng(0)
    pool = ThreadPoolExecutor(cores)
    nthr = min(cores, E)

    def one():
        act = np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        r.step_threads(act, pool, nthr)
    one()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds and n < 2000:
        one()
        n += 1
    dt = time.perf_counter() - t0
    # SURVEY.md 8d extras, bounded to a few seconds each: the same oracle on ONE thread, and a
    # "reference-shaped" step (1 arena, 512 beams over 2*pi, 1000x1000 map, 10 pedestrians, every
    # pedestrian's own 512-beam scan computed each step as env.py:685-693 does)
    t1 = time.perf_counter()
    m = 0
    while time.perf_counter() - t1 < 3.0 and m < 2000:
        act = np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        r.step_threads(act, pool, 1)
        m += 1
    one_thread = E * m / (time.perf_counter() - t1)
    pool.shutdown()
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
    return dict(value=E * n / dt, unit="env-steps/s", cores=nthr, kind="port",
                sample="%d arenas x %d steps of the same workload (oracle/navsim_ref.c, %d threads, %.1f s)"
                       % (E, n, nthr, dt),
                value_1_thread=one_thread,
                reference_shaped_us_per_step=ref_shaped_us,
                reference_shaped="1 arena, 512 beams, 1000x1000 map, 10 pedestrians + their 512-beam scans, 1 thread "
                                 "(restatement, not the reference binary)")


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
    reader.join(5.0)
    sys.stdout.write(b"".join(out0).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %s" % ", ".join("rank %d exit %d" % rc for rc in bad))


def profiled_traffic(workload, E, field):
    """HBM bytes per launch of the step kernel from the committed PMC profile of this workload (rocprofv3 --pmc
    passes, profiles/pmc_pass.sh), NOT a reading of this run: counters cannot be collected inside the timed
    process.  -> (bytes or None, 'file@commit' or None)"""
    for rnd in ("r02", "r01"):
        tp = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, workload), "traffic.json")
        if not os.path.exists(tp):
            continue
        try:
            t = json.load(open(tp))
        except Exception:
            continue
        if t.get("envs_per_gpu", WORKLOADS[workload]["envs"]) != E or field != "u16t":
            return None, None                       # another launch shape: the stored figure does not apply
        return t.get("hbm_bytes_per_launch"), "profiles/%s_%s/traffic.json@%s" % (rnd, workload, t.get("commit", "?"))
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override arenas per GPU")
    ap.add_argument("--gather", default="none", choices=["none", "all"])
    ap.add_argument("--field", default="u16t", choices=["u16t", "f32"], help="distance-field storage")
    ap.add_argument("--noise-std", type=float, default=0.02,
                    help="scan_noise_std of every arena in the timed steps (SURVEY.md 8d: 0.02 for throughput runs)")
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of K steps (the first one is `value`)")
    ap.add_argument("--no-noise-off-pass", action="store_true",
                    help="skip the extra K steps timed with scan noise off (profiling runs: one kind of launch only)")
    ap.add_argument("--step-block", type=int, default=0, help="threads per arena (0 = library default)")
    ap.add_argument("--spinup-ms", type=float, default=500.0,
                    help="untimed GPU work before the warm-up steps (leaves the idle power state); 0 = none")
    ap.add_argument("--lpt-period", type=int, default=0, help="steps between launch-order sorts (0 = NavSim's default)")
    ap.add_argument("--ped-split", type=int, default=0, choices=[0, 1, 2],
                    help="navsim_config.ped_split: 0 library default, 1 pedestrians inside the step, 2 ped_update_kernel first")
    ap.add_argument("--no-rects", action="store_true",
                    help="march through the packed field only, without the two-rectangle tile records (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", default="off", choices=["auto", "on", "off"],
                    help="replay the K timed steps as one captured hipGraph (measured: c2 +0 %, c5 +2.5 %; off by default)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction control flow only, no GPU work (tests/test_host_logic.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
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
    if os.environ.get("NAVSIM_BENCH_ONE_GPU"):
        local_rank = 0
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    coll_dev = device if backend == "nccl" else "cpu"

    wl = dict(WORKLOADS[args.workload])
    if args.envs:
        wl["envs"] = args.envs
    wl["field"] = args.field
    if args.no_rects:
        wl["rects"] = False
    cfg, sim, arrays, _ = build_sim(wl, rank, world_size, device=device)
    if args.step_block:
        sim.cfg.step_block = args.step_block
    if args.ped_split:
        sim.cfg.ped_split = args.ped_split
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
    gather_buf = None
    if args.gather == "all" and dist is not None and backend == "nccl":
        gather_buf = torch.empty((world_size * E, sim.obs.shape[1]), dtype=torch.float32, device=device)

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
        if gather_buf is not None:
            dist.all_gather_into_tensor(gather_buf, sim.obs)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
    graph = None
    if args.graph != "off" and gather_buf is None:
        try:
            cur0 = sim.cur
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for t in range(K):
                    run(Wm + t)
        except Exception as exc:                      # capture unsupported here: plain launches
            if args.graph == "on":
                raise
            graph = None
            sim.cur = cur0
            torch.cuda.synchronize()
            if rank == 0:
                print("bench: hipGraph capture failed (%s); timing plain launches" % type(exc).__name__, file=sys.stderr)

    # HIP events on the launch stream.  Where the step launch is the only kernel of a step (c1-c4) ONE pair brackets
    # the K launches and kernel_ms = that span / K: an upper bound of the kernel's duration (it includes the 1-2 us
    # between consecutive launches), and nothing is inserted between the launches that are being timed -- an event
    # pair per step costs 4 % of the throughput at 120 us per step.  Where other kernels run between the steps
    # (navsim_regen, the obs gather) every step launch gets its own pair.
    events_per_step = (regen or gather_buf is not None) and graph is None

    def timed():
        """EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
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

    elapsed, kernel_ms = timed()                     # THE measurement (`value`)
    more = [timed() for _ in range(max(args.repeats - 1, 0))]
    noise_off = None
    if args.noise_std > 0 and not args.no_noise_off_pass:   # the same K steps without the per-beam Gaussian, beside it
        sim.cfg.add_scan_noise = 0
        noise_off = timed()
        sim.cfg.add_scan_noise = 1

    if rank == 0:
        import statistics
        n_done = int(sim.t["episode"].sum().item())
        # s_map = 1: the arena's occupancy grid as the reference stores it (int8 map_info['data'],
        # map_generator.py:136) read once per env-step -- BASELINE.md section 3's definition.  The
        # on-device distance-field encoding is an implementation choice, not algorithmic bytes.
        A = algorithmic_bytes_per_env_step(cfg.map_h, cfg.map_w, cfg.n_beams, cfg.n_scan_stack, wl["peds"], 1)
        achieved = A * E / (kernel_ms * 1e-3) / 1e9                # GB/s
        frac = achieved / 8000.0
        # the same formula with s_map = bytes per cell of the representation the kernel actually marches on
        # ("the map representation the kernel streams"): rect records are 16 B per 8x8 tile = 0.25,
        # the packed distance field 2, the float32 field 4
        from nav_gym_amd import abi
        s_streamed = 0.25 if "rect_table" in sim.t else (2 if cfg.field_format == abi.FIELD_U16T else 4)
        A_streamed = algorithmic_bytes_per_env_step(cfg.map_h, cfg.map_w, cfg.n_beams, cfg.n_scan_stack, wl["peds"], s_streamed)
        tbytes, tsrc = profiled_traffic(args.workload, E, args.field)
        all_values = [world_size * E * K / el for el, _ in [(elapsed, kernel_ms)] + more]
        out = {
            "metric": "env steps/sec (whole node), 4096 envs x 1081-beam lidar",
            "value": world_size * E * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world_size,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d arenas/GPU x %d-beam lidar, %dx%d per-arena occupancy maps (%s distance "
                            "field), %d pedestrians/arena, %s kinematics, %s, scan_noise_std %.3g"
                            % (args.workload, E, cfg.n_beams, cfg.map_h, cfg.map_w, args.field, wl["peds"],
                               wl.get("robot", "keti"),
                               "new random map per episode (navsim_regen)" if regen else "auto-respawn in place",
                               args.noise_std),
                "envs_per_gpu": E, "n_beams": cfg.n_beams, "map": [cfg.map_h, cfg.map_w],
                "pedestrians": wl["peds"], "obs_gather": args.gather, "episodes_finished_rank0": n_done,
                "launch": "hipGraph replay of the K steps" if graph is not None else "one launch per step",
                "ranks": world_size, "collective_backend": (backend if world_size > 1 else None),
                "scan_noise_std": args.noise_std, "rect_table": "rect_table" in sim.t,
                "gpu_spinup_ms": args.spinup_ms,     # untimed, before the warm-up steps, touches no simulator state
            },
            "repeats": {"n": len(all_values), "values": all_values, "median": statistics.median(all_values)},
            "roofline": {
                # SURVEY.md 8d figure: algorithmic bytes (whole occupancy grid once per arena-step) / kernel time.
                # A march touches only part of the grid, so on large maps the formula can exceed 1: then it says
                # nothing about the kernel and hbm_frac_measured (counter bytes) is the figure to read.
                "bound": "hbm",
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": frac,
                "note": (None if frac <= 1.0 else
                         "the 8d figure counts the whole occupancy grid once per arena-step; the kernel reads an 8x smaller "
                         "lossless description of it (rect records) and only along the rays, so achieved exceeds the peak: "
                         "the formula has stopped being a bound here, read hbm_frac_measured (counter bytes)"),
                "traffic": None,                # not measured in this process (PMC needs rocprofv3)
                "traffic_profiled": tbytes, "traffic_profile": tsrc,
                "hbm_frac_measured": (tbytes / (kernel_ms * 1e-3) / 8.0e12) if tbytes else None,
                "kernel": "navsim_step_kernel", "kernel_ms": kernel_ms,
                "kernel_ms_from": ("one HIP event pair per step launch" if events_per_step else
                                   "one HIP event pair around the %d launches / %d (includes the gaps between launches)" % (K, K)),
                "algorithmic_bytes_per_env_step": A, "s_map": 1,
                "s_map_streamed": s_streamed, "frac_s_map_streamed": A_streamed * E / (kernel_ms * 1e-3) / 8.0e12,
            },
        }
        if noise_off is not None:
            out["noise_off"] = {"value": world_size * E * K / noise_off[0], "ms_per_step": noise_off[0] / K * 1e3,
                                "kernel_ms": noise_off[1]}
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, rank, world_size, backend):
    """Launcher / rendezvous / max-over-ranks control flow without a GPU (CPU test of `--gpus N`): every rank
    joins the process group, contributes a fake per-rank time, rank 0 prints the line shape with n_gpus = the
    process group's size.  Measures nothing."""
    dist = None
    elapsed = 1.0 + rank
    if world_size > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if backend == "nccl" else backend)
        assert dist.get_world_size() == args.gpus
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if os.environ.get("NAVSIM_BENCH_FAIL_RANK") == str(rank):      # failure-propagation test
        raise SystemExit(3)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work, control flow only)", "value": None, "n_gpus": world_size,
                          "steps": args.steps, "warmup": args.warmup, "max_rank_time": elapsed, "dry_run": True}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
