"""Phases of the 1024-thread step kernel on c5 (512 arenas, one generation of workgroups: the launch lasts as long as its
slowest workgroup).  Diagnostic build with -DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME (ticks of 10 ns):
   NAVSIM_LIB=build/libnavsim_stamps.so python3 profiles/_diag/c5_step_phases.py"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
wl = dict(bench.WORKLOADS[os.environ.get("NAVSIM_WL", "c5")]); wl["field"] = "u16t"
if os.environ.get("NAVSIM_BEAMS"):            # what does the 17th chunk of 64 beams cost the 16 wavefronts? (1024 = no 17th)
    from nav_gym_amd import world
    nb = int(os.environ["NAVSIM_BEAMS"])
    def _lidar(cfg, _n=nb):
        cfg.n_beams = _n; cfg.angle_min = -0.75 * np.pi; cfg.angle_last = 0.75 * np.pi; return cfg
    world.lidar_1081 = _lidar
    print("beams", nb)
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
L = lib.load()
SLOTS = int(os.environ.get("NAVSIM_STAMP_SLOTS", "8"))      # 16 with profiles/_diag/tried/ped_chain_stamps.patch.txt
buf = torch.zeros((E, SLOTS), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
T = 40
acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
names = ["0-1 scalars", "1-2 pedestrians + robot + primitives", "2-3 scan A (+ merge)", "3-4 flags", "4-5 reward (+ rescan)", "5-6 pack"]
rows, spans, crit = [], [], []
for t in range(T):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.float64)
    if wl.get("regen"):
        sim.regen(); torch.cuda.synchronize()
    if t < 8:
        continue
    d = np.diff(b[:, :7], axis=1)
    rows.append(d)
    spans.append(b[:, 6].max() - b[:, 0].min())
    last = int(np.argmax(b[:, 6]))                       # the workgroup the launch waited for
    crit.append(np.concatenate([[b[last, 0] - b[:, 0].min()], d[last]]))
d = np.concatenate(rows)
print("launch span (first stamp 0 to last stamp 6), us: mean %.1f" % (np.mean(spans) / 100))
print("%-40s %8s %8s %8s %8s" % ("phase, us", "mean", "median", "p99", "max"))
for i, nm in enumerate(names):
    print("%-40s %8.1f %8.1f %8.1f %8.1f" % (nm, d[:, i].mean() / 100, np.median(d[:, i]) / 100, np.percentile(d[:, i], 99) / 100, d[:, i].max() / 100))
print("workgroup lifetime us: mean %.1f median %.1f p99 %.1f" % (d.sum(1).mean() / 100, np.median(d.sum(1)) / 100, np.percentile(d.sum(1), 99) / 100))
c = np.array(crit)
print("the LAST workgroup of each launch (mean over launches), us: start offset %.1f ; " % (c[:, 0].mean() / 100) +
      " ; ".join("%s %.1f" % (names[i], c[:, i + 1].mean() / 100) for i in range(6)))
