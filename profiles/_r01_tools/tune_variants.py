#!/usr/bin/env python3
"""A/B of the fused step kernel's launch geometry in ONE process (interleaved rounds, rule 24 of
the CDNA guide): NAVSIM_STEP_VARIANT is re-read by the library at every launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else "c2"
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["256x1", "256x2", "256x3", "256x5", "384x3", "576x2", "192x6", "128x9", "1024x2"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
K = 40
wl = dict(bench.WORKLOADS[wl_name])
wl["field"] = os.environ.get("NAVSIM_FIELD", "u16t")
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
if os.environ.get("NAVSIM_SHARED_FIELD"):            # diagnostic: all arenas march arena 0's field (L2-resident)
    sim.cfg.shared_field = 1
E = cfg.n_envs
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((K, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        os.environ["NAVSIM_STEP_VARIANT"] = v
        for t in range(5):
            sim.io.action = acts[t].data_ptr(); sim.launch_step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(K):
            sim.io.action = acts[t].data_ptr(); sim.launch_step()
        b.record(); torch.cuda.synchronize()
        res[v].append(a.elapsed_time(b) / K)
print("workload %s, %d arenas, ms per step (median / min over %d rounds), env-steps/s at median" % (wl_name, E, rounds))
for v in variants:
    xs = sorted(res[v]); med = xs[len(xs) // 2]
    print("%-8s %.4f %.4f  %.3e" % (v, med, xs[0], E / med * 1e3))
