"""Does a hipGraph captured and replayed earlier in the process (bench.py other_workloads: c5's K steps as one graph) slow a later
pipelined environment down?  NAVSIM_FIRST=capture|plain|none"""
import os, sys, runpy
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
first = os.environ.get("NAVSIM_FIRST", "capture")
if first != "none":
    wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0)
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    def one():
        sim._reorder(); sim.launch_step(reorder=False); sim.regen()
    for t in range(30):
        one()
    torch.cuda.synchronize()
    if first == "capture":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(100):
                one()
        torch.cuda.synchronize()
        g.replay(); g.replay(); torch.cuda.synchronize()
        del g
    del sim, arrays
    torch.cuda.empty_cache()
runpy.run_path(os.path.join(ROOT, "profiles", "_diag", "gym_refdef_steps.py"), run_name="__main__")
