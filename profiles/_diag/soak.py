"""One-off soak (add "peds" as third argument for 20 social-force pedestrians per arena, both pedestrian-update forms): E arenas, the forced 256-thread kernel (parked rays, scalar-mask loop, table directions), many steps
of random actions with crash reverts and respawns, every output of every step compared with the oracle bit for bit.
   python profiles/_diag/soak.py [steps] [arenas]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from nav_gym_amd import abi, lib, robots, sim, world
import ref

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
size = 500
dev = torch.device("cuda:0")
peds = len(sys.argv) > 3 and sys.argv[3] == "peds"
for S, noise in ((1, 0), (2, 0)):
    cfg = lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=20 if peds else 1, n_scan_stack=S,
                             ped_model=abi.PED_SFM if peds else abi.PED_NONE, ped_split=S if peds else 0,
                             auto_reset=1, n_spawn=16, seed=2024 + S, field_format=abi.FIELD_U16T, step_block=256)
    world.lidar_1081(cfg)
    occ = world.make_maps(E, size, 2024 + S)
    arrays = world.make_world(cfg, occ, n_peds=20 if peds else 0, device=dev)
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = sim.scan_threshold(cfg, torch.from_numpy(robots.footprint_array("keti", name)).to(dev))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table")}
    host["field"] = ref.build_dt(occ)
    g = sim.NavSim(cfg, arrays); r = ref.RefSim(cfg, host)
    assert np.array_equal(g.reset_obs().cpu().numpy(), r.reset_obs())
    rng = np.random.default_rng(7)
    crashes = dones = 0
    t0 = time.time()
    for t in range(steps):
        act = np.stack([rng.uniform(0.0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        if t % 11 == 5: act[:, 0] = 0.5; act[:, 1] = 0.0
        go, gout = g.step(torch.from_numpy(act).to(dev)); ro, rout = r.step(act)
        if not np.array_equal(go.cpu().numpy(), ro):
            bad = np.argwhere(go.cpu().numpy() != ro)
            print("OBS MISMATCH step", t, "first", bad[:5], "count", len(bad)); sys.exit(1)
        for k in rout:
            if not np.array_equal(gout[k].cpu().numpy(), rout[k]):
                print("MISMATCH", k, "step", t); sys.exit(1)
        crashes += int(rout["is_crash"].sum()); dones += int(rout["done"].sum())
    print("S=%d: %d steps x %d arenas identical to the oracle (%d crashes, %d episodes ended, %.0f s)" % (S, steps, E, crashes, dones, time.time() - t0), flush=True)
