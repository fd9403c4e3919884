import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, ROOT)
import torch, bench
wl = dict(bench.WORKLOADS["c3"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
for _ in range(3): sim.ped_scans()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): sim.ped_scans()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("NAVSIM_LIB", "default"), "ped_scans %.3f ms" % (e0.elapsed_time(e1) / 10))
