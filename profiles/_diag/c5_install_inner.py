"""Inside the install of navsim_step_install_kernel (c5): diagnostic build -DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME -DNAVSIM_STAMPS_INSTALL
-- an installing workgroup overwrites stamps 3..6 with: 3 install start, 4 all small rows loaded, 5 fifteen rows stored,
6 the waypoint row stored, 7 end (wavefront 0's view, ticks of 10 ns)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0, pregen=True, pipeline=4, install=True)
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
L = lib.load()
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
T = 200
acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * 2.0
rows = []
for t in range(T):
    slots0 = sim.t["map_slot"].clone()
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.float64)
    inst = (sim.t["map_slot"] != slots0).cpu().numpy()       # the arenas that installed in this launch
    sim.regen(); torch.cuda.synchronize()
    if t < 40:
        continue
    if inst.any():
        rows.append(np.diff(b[inst][:, 3:8], axis=1))
d = np.concatenate(rows)
print("%d installs; us: loads issued+arrived %.2f ; first 15 rows stored %.2f ; waypoint row (stores + 2 more rounds) %.2f ; rest incl. obs %.2f ; total %.2f"
      % (len(d), d[:, 0].mean() / 100, d[:, 1].mean() / 100, d[:, 2].mean() / 100, d[:, 3].mean() / 100, d.sum(1).mean() / 100))
