#!/usr/bin/env python3
"""Where does one arena's workgroup spend its time?  Needs the diagnostic build:
   NAVSIM_OUT=$PWD/gpurun_out/libnavsim_stamps.so NAVSIM_EXTRA_FLAGS=-DNAVSIM_STAMPS nav-gym_amd/csrc/build.sh
   NAVSIM_LIB=$PWD/gpurun_out/libnavsim_stamps.so python profiles/stamp_phases.py [envs] [variant]
Shares only -- the stamped build's run time is not quoted anywhere."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch
import bench
from nav_gym_amd import lib
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2: os.environ["NAVSIM_STEP_VARIANT"] = sys.argv[2]
wl = dict(bench.WORKLOADS[os.environ.get("NAVSIM_WL", "c2")]); wl["envs"] = E; wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
L = lib.load()
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda:0")
L.navsim_debug_set_stamps.argtypes = [C.c_void_p]
assert L.navsim_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0, "not a -DNAVSIM_STAMPS build"
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((20, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
tot = torch.zeros(6, dtype=torch.float64)
span = 0.0
for t in range(20):
    sim.io.action = acts[t].data_ptr(); sim.launch_step(); torch.cuda.synchronize()
    if t >= 5:
        b = buf.cpu().double()
        d = b[:, 1:7] - b[:, 0:6]
        tot += d.mean(0)
        span += float((b[:, 6].max() - b[:, 0].min()))
names = ["0 scalars+barrier", "1 peds/robot integrate", "2 scan A", "3 flag reduce", "4 reward/relocate (+scan B)", "5 pack obs/state"]
tot /= 15; span /= 15
print("E=%d variant=%s: mean cycles per arena workgroup by phase (s_memtime, 100 MHz ticks x ? -> shader cycles)" % (E, os.environ.get("NAVSIM_STEP_VARIANT", "default")))
for n, v in zip(names, tot.tolist()):
    print("  %-32s %10.0f  %5.1f %%" % (n, v, 100 * v / float(tot.sum())))
print("  workgroup lifetime %.0f ; launch span (first start .. last end) %.0f" % (float(tot.sum()), span))
