"""How many wavefront probe rounds would parking save?  Takes the c2 bench world after 100 steps, copies the maps and robot
poses of 96 arenas to the host and counts every beam's probes with a NumPy march on SciPy's exact distance field
(t += max(0.999 d, 1), as the kernel).  Baseline: a wavefront marches 64 adjacent beams until the last one is done.
Parking: it stops when at most T rays are still marching, parks those and they are marched later 64 at a time.
   python profiles/_diag/park_model.py        (on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
from scipy import ndimage

wl = dict(bench.WORKLOADS["c2"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E, B, SIZE = cfg.n_envs, cfg.n_beams, cfg.map_h
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
acts = torch.rand((100, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for t in range(100):
    sim.io.action = acts[t].data_ptr(); sim.launch_step()
torch.cuda.synchronize()
pose = sim.t["robot_pose"].cpu().numpy()
ang = np.linspace(cfg.angle_min, cfg.angle_last, B)

def counts(d, i0, j0, th, max_range):
    dx, dy = np.cos(th + ang), np.sin(th + ang)
    t = np.zeros(B); n = np.zeros(B, int); act = np.ones(B, bool)
    while act.any():
        x = np.floor(i0 + t * dx).astype(int); y = np.floor(j0 + t * dy).astype(int)
        act &= (x >= 0) & (x < SIZE) & (y >= 0) & (y < SIZE) & (t < max_range)
        dd = np.where(act, d[np.clip(x, 0, SIZE - 1), np.clip(y, 0, SIZE - 1)], 0.0)
        n += act
        act &= dd > 0
        t = np.where(act, t + np.maximum(0.999 * dd, 1.0), t)
    return n

Ts = [0, 4, 8, 12, 16, 24]
tot = {T: 0.0 for T in Ts}; parked = {T: 0 for T in Ts}; ideal = 0.0; nscan = 0
for e in range(0, E, E // 96):
    occ = sim.occupancy(e)
    d = ndimage.distance_transform_edt(occ == 0)
    fi = (pose[e, 0] - cfg.origin_x) / cfg.resolution; fj = (pose[e, 1] - cfg.origin_y) / cfg.resolution
    n = counts(d, float(int(fi)), float(int(fj)), float(np.float32(pose[e, 2])), cfg.range_max / cfg.resolution + 4)
    ideal += n.sum() / 64.0; nscan += 1
    for T in Ts:
        rounds = 0; rem = []
        for c0 in range(0, B, 64):
            v = np.sort(n[c0:c0 + 64])[::-1]
            if T == 0 or len(v) <= T:
                rounds += v[0]
                continue
            r_exit = v[T]
            rounds += r_exit
            rem += [x - r_exit for x in v[:T] if x > r_exit]
        parked[T] += len(rem)
        for q in range(0, len(rem), 64):
            rounds += max(rem[q:q + 64])
        tot[T] += rounds
base = tot[0]
print("%d scans, %.1f rounds per 64-beam chunk, useful lanes %.3f" % (nscan, base / nscan / 17, ideal / base))
for T in Ts:
    print("T=%2d  rounds %.3f of baseline   parked rays per scan %.1f" % (T, tot[T] / base, parked[T] / nscan))
