"""How many tiles of an arena's index row carry no record (the probe then reads the field), per arena: c2 / c3 / c4 worlds of bench.py."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
for name in os.environ.get("NAVSIM_WLS", "c2,c4,c5").split(","):
    wl = dict(bench.WORKLOADS[name]); wl["field"] = "u16t"
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, min(wl["envs"], 1024))
    rows = sim.t["rect_index"].cpu().numpy()
    H, W = cfg.map_h, cfg.map_w
    nt = ((H + 7) // 8) * ((W + 7) // 8)
    pair = rows[:, 2048:2048 + 2 * nt].reshape(rows.shape[0], nt, 2)
    inval = (pair[:, :, 0] == 255).sum(axis=1)
    print("%s: %d arenas, %d tiles each; tiles without a record per arena: mean %.1f, median %d, max %d; arenas with none: %d"
          % (name, rows.shape[0], nt, inval.mean(), np.median(inval), inval.max(), int((inval == 0).sum())))
    del sim
