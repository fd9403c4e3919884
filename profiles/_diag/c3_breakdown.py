"""Where does c3's step differ from c2's?  Same world, pedestrian variant of the kernel, with n_peds = 20 / 0, and
with pedestrians that cannot be seen (moved far away is not possible inside a map: has_legs off / on instead)."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, ROOT)
import torch
import bench

def timed(sim, acts, n=60):
    for t in range(10):
        sim.io.action = acts[t % len(acts)].data_ptr(); sim._reorder(); sim.launch_step(reorder=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for t in range(n):
        sim.io.action = acts[t % len(acts)].data_ptr(); sim._reorder(); sim.launch_step(reorder=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

wl = dict(bench.WORKLOADS["c3"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
acts = torch.rand((32, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
rows = {"c3 (20 pedestrians, split update)": timed(sim, acts)}
sim.cfg.ped_split = 1
rows["c3, pedestrian update inside the step"] = timed(sim, acts)
sim.cfg.ped_split = 0
n0 = sim.t["n_peds"].clone()
sim.t["n_peds"].zero_()
rows["pedestrian variant of the kernel, n_peds = 0"] = timed(sim, acts)
sim.t["n_peds"].copy_(n0)
sim.t["ped_has_legs"].zero_()
rows["20 pedestrians, all rectangles (no legs)"] = timed(sim, acts)
sim.t["ped_has_legs"].fill_(1)
rows["20 pedestrians, all legs"] = timed(sim, acts)
print(json.dumps(rows, indent=1))
