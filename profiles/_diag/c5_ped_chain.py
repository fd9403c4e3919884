"""Where wavefront 0's pedestrian chain spends its time in c5's step (diagnostic build: -DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME
with the extra stamps 8..12 of profiles/_diag/tried/ped_chain_stamps.patch.txt applied; 16 stamp slots per workgroup)."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS[os.environ.get("NAVSIM_WL", "c5")]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
L = lib.load()
buf = torch.zeros((E, 16), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
T = 40
acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
seq = [0, 1, 8, 9, 10, 11, 12, 2, 3, 4, 5, 6]
names = ["0-1 before the barrier (loads, pop, staging | phase 0 on wavefront 1)", "1-8 pair terms (4 rounds)", "8-9 ped_sfm_step",
         "9-10 ped_finish", "10-11 primitives", "11-12 prim_in_range", "12-2 to the scan", "2-3 chunks left + merge + finish_beams",
         "3-4 flags", "4-5 reward (+ rescan)", "5-6 pack"]
rows = []
for t in range(T):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.float64)
    if wl.get("regen"):
        sim.regen(); torch.cuda.synchronize()
    if t >= 8:
        rows.append(np.diff(b[:, seq], axis=1))
d = np.concatenate(rows)
for i, nm in enumerate(names):
    print("%-75s mean %6.2f  median %6.2f  p99 %6.2f us" % (nm, d[:, i].mean() / 100, np.median(d[:, i]) / 100, np.percentile(d[:, i], 99) / 100))
