#!/usr/bin/env python3
"""Kernel time vs number of arenas (how much is latency chain, how much is throughput)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch
import bench
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["256x1", "256x5"]
for E in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "256,1024,2048,4096,8192".split(","))]:
    wl = dict(bench.WORKLOADS["c2"]); wl["envs"] = E; wl["field"] = "u16t"
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    K = 30
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    acts = torch.rand((K, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    line = "E=%5d " % E
    for v in variants:
        os.environ["NAVSIM_STEP_VARIANT"] = v
        best = 1e9
        for r in range(3):
            for t in range(3):
                sim.io.action = acts[t].data_ptr(); sim.launch_step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for t in range(K):
                sim.io.action = acts[t].data_ptr(); sim.launch_step()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / K)
        line += " %s: %.1f us (%.2e/s)" % (v, best * 1e3, E / best * 1e3)
    print(line, flush=True)
    del sim, arrays
    torch.cuda.empty_cache()
