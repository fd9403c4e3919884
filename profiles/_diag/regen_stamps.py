"""Phase stamps of regen_maps_kernel (diagnostic build -DNAVSIM_STAMPS, s_memtime ticks = 10 ns):
   NAVSIM_LIB=build/libnavsim_stamps.so python profiles/_diag/regen_stamps.py
stamps of slice 7 of every live slot: 0 start, 1 after count / list, 2 after the obstacle set-up + barrier,
3 after the first item's distances (before its stores), 4 end (stores drained)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS["c5"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
L = lib.load()
buf = torch.zeros((cfg.n_envs, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
rows = []
for rep in range(30):
    sim.out["done"].zero_()
    sim.out["done"][torch.tensor([3 + rep, 77, 200 + rep, 301, 500 - rep], device="cuda:0")] = 1
    sim.t["episode"] += 1
    buf.zero_()
    # something big in between, like the step: evicts L2
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0").fill_(1.0)
    torch.cuda.synchronize()
    sim.regen()
    torch.cuda.synchronize()
    b = buf.cpu().numpy()[:5, :5].astype(np.float64)
    rows.append(np.diff(b, axis=1))
d = np.stack(rows)          # [rep, slot, phase]
print("ticks (10 ns) per phase, median over reps and slots: 0-1 count/list, 1-2 episode + draws + barrier, 2-3 first item's distances, 3-4 stores + rest:")
print(np.median(d.reshape(-1, 4), axis=0), " mean", d.reshape(-1, 4).mean(axis=0))
