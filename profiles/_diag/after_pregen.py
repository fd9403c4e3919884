"""Does a world with pipelined pre-generation, built and dropped earlier in the process, slow a later environment down?
(bench.py's default run measures c5_pipelined before the gym-API windows.)"""
import os, sys, time, runpy
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
if os.environ.get("NAVSIM_FIRST", "1") == "1":
    wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0, pregen=True, pipeline=4, install=True)
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    for t in range(40):
        sim.launch_step(); sim.regen()
    torch.cuda.synchronize()
    if os.environ.get("NAVSIM_SYNC_SIDE"):
        sim.side.synchronize()
    del sim, arrays
    torch.cuda.empty_cache()
runpy.run_path(os.path.join(ROOT, "profiles", "_diag", "gym_refdef_steps.py"), run_name="__main__")
