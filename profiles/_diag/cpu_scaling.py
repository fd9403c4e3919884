"""How the CPU baseline (oracle/navsim_ref.c navsim_step_threads_cpu) scales with threads on this host, 8 arenas per thread (c2 shape)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ref
from nav_gym_amd import abi, lib, robots, world
cores = len(os.sched_getaffinity(0))
for nthr in [1, 4, 16, 64, 128, 256]:
    if nthr > cores: break
    E = 8 * nthr
    cfg = lib.default_config(n_envs=E, map_h=500, map_w=500, max_peds=1, ped_model=abi.PED_NONE, n_spawn=16, auto_reset=1, seed=1234)
    world.lidar_1081(cfg)
    occ = world.make_maps(E, 500, 1234)
    field = torch.from_numpy(ref.build_dt(occ))
    arrays = world.make_world(cfg, occ, n_peds=0, device="cpu", field=field)
    host = {k: v.numpy() for k, v in arrays.items()}
    host["scan_threshold"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    host["scan_discomfort"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    host["field"] = ref.field_local_copy(host["field"], nthr)
    r = ref.RefSim(cfg, host, keep=("field",)); r.reset_obs()
    rng = np.random.default_rng(0)
    acts = lambda n: np.stack([rng.uniform(0, 0.5, (n, E)), rng.uniform(-0.64, 0.64, (n, E))], axis=2)
    r.step_native_threads(acts(4), nthr)
    n = 100
    a = acts(n)
    t0 = time.perf_counter(); r.step_native_threads(a, nthr); dt = time.perf_counter() - t0
    print("threads %3d arenas %4d: %9.0f env-steps/s, %7.0f per thread, %.2f s" % (nthr, E, E * n / dt, E * n / dt / nthr, dt), flush=True)
