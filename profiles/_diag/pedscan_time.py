import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
wl = dict(bench.WORKLOADS["c3"]); wl["field"]="u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, 1)
for _ in range(3): sim.ped_scans()
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): out=sim.ped_scans()
b.record(); torch.cuda.synchronize()
ms=a.elapsed_time(b)/20
print("ped scans c3: %.3f ms per call, %d x 20 x 512 rays -> %.2e rays/s"%(ms, cfg.n_envs, cfg.n_envs*20*512/ms*1e3))
