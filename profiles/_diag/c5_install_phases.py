"""Phases of navsim_step_install_kernel on c5 (512 arenas, one generation of 1024-thread workgroups: the launch lasts as long
as its slowest workgroup): which workgroup ends a launch -- one that installs a staged world? -- and what its tail is made of.
Diagnostic build with -DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME (ticks of 10 ns):
   NAVSIM_LIB=build/libnavsim_stamps.so python3 profiles/_diag/c5_install_phases.py [pipeline period]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import numpy as np, torch, bench
from nav_gym_amd import lib
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0, pregen=True, pipeline=P, install=True)
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
L = lib.load()
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
T = 240
acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * 2.0
names = ["0-1 scalars", "1-2 peds+robot+prims", "2-3 scan A", "3-4 flags", "4-5 reward (+rescan)", "5-6 pack", "6-7 install"]
spans, last_rows, inst_rows, other_rows, last_is_inst = [], [], [], [], []
ep_before = sim.t["episode"].clone()
for t in range(T):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.float64)
    served0 = sim.counters()["regen_served"]
    sim.regen(); torch.cuda.synchronize()
    if t < 40:
        continue
    d = np.diff(b, axis=1)                                   # [E, 7]
    spans.append(b[:, 7].max() - b[:, 0].min())
    last = int(np.argmax(b[:, 7]))
    inst = d[:, 6] > 100                                     # an install takes more than 1 us between stamps 6 and 7
    last_is_inst.append(bool(inst[last]))
    last_rows.append(np.concatenate([[b[last, 0] - b[:, 0].min()], d[last]]))
    if inst.any():
        inst_rows.append(d[inst])
    other_rows.append(d[~inst])
print("pipeline %d: launch span us mean %.1f; the last workgroup installs in %.0f %% of the launches; installs per launch %.2f"
      % (P, np.mean(spans) / 100, 100 * np.mean(last_is_inst), sum(len(r) for r in inst_rows) / len(spans)))
fmt = lambda a: " ; ".join("%s %.1f" % (names[i], a[i] / 100) for i in range(7))
print("workgroups that install (mean us):   " + fmt(np.concatenate(inst_rows).mean(0)) + "  | lifetime %.1f" % (np.concatenate(inst_rows).sum(1).mean() / 100))
print("workgroups that do not (mean us):    " + fmt(np.concatenate(other_rows).mean(0)) + "  | lifetime %.1f" % (np.concatenate(other_rows).sum(1).mean() / 100))
o = np.concatenate(other_rows).sum(1)
print("   ... their lifetime p50 %.1f p99 %.1f max %.1f" % (np.median(o) / 100, np.percentile(o, 99) / 100, o.max() / 100))
c = np.array(last_rows)
print("the LAST workgroup of a launch (mean):  start offset %.1f ; " % (c[:, 0].mean() / 100) + fmt(c[:, 1:].mean(0)))
