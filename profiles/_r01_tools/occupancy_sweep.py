#!/usr/bin/env python3
"""Kernel time vs resident workgroups per CU (LDS padding limits residency): latency- or throughput-bound?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch
import bench
wl = dict(bench.WORKLOADS["c2"]); wl["field"] = "u16t"
if len(sys.argv) > 1: wl["envs"] = int(sys.argv[1])
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs; K = 30
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((K, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for pad, wg in ((0, 8), (22000, 7), (26000, 6), (31000, 5), (39000, 4), (52000, 3), (79000, 2), (150000, 1)):
    os.environ["NAVSIM_LDS_PAD"] = str(pad)
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
    print("workgroups/CU <= %d (LDS pad %6d): %.1f us  -> %.2e env-steps/s" % (wg, pad, best * 1e3, E / best * 1e3), flush=True)
