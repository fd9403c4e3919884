"""Active-workgroup timeline of one c2 step from the s_memtime stamps (diagnostic build):
   NAVSIM_LIB=.../libnavsim_stamps.so python profiles/_diag/tail_profile.py"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS[os.environ.get("NAVSIM_WL", "c2")]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
L = lib.load()
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((20, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for t in range(20):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
b = buf.cpu().numpy()
s, e = b[:, 0].astype(np.float64), b[:, 6].astype(np.float64)
t0, t1 = s.min(), e.max()
span = t1 - t0
grid = np.linspace(t0, t1, 201)
active = [(np.sum((s <= x) & (e > x))) for x in grid]
print("span ticks %.0f ; mean WG lifetime %.0f ; sum lifetimes / span = mean active WGs %.0f (of %d slots)" % (span, (e - s).mean(), (e - s).sum() / span, 256 * 8))
print("active WGs at 0,5,...,100 %% of the span:", [int(active[i]) for i in range(0, 201, 10)])
order = np.argsort(s)
print("start-time of WG by launch index quartiles:", [int(s[order[int(q * (E - 1))]] - t0) for q in (0, .25, .5, .75, 1)])
print("longest 5 lifetimes:", np.sort(e - s)[-5:].astype(int), " median", int(np.median(e - s)))
# phase durations (ticks of 10 ns): stamps 0 start, 1 after the scalars, 2 before scan A, 3 after scan A, 4 after the
# flag reduction, 5 after reward / rescan, 6 end
d = np.diff(b[:, :7].astype(np.float64), axis=1)
print("mean ticks per phase (0-1 scalars, 1-2 robot / lidar pose / first probe, 2-3 scan A, 3-4 reduce, 4-5 reward (+ rescan), 5-6 pack):",
      [int(x) for x in d.mean(axis=0)])
print("median:", [int(x) for x in np.median(d, axis=0)])
