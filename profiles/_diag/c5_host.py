"""c5 with pipelined pre-generation + in-step install: is the loop host-bound?  Host enqueue time per step (the loop returns
before the GPU has finished) against the wall time per step, plain launches.  Args: pipeline period (default 4).  NAVSIM_NO_RULE=1: regen_min_steps 0, the fallback's navsim_regen after every step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0, pregen=bool(P), pipeline=P, install=True, no_rule=bool(os.environ.get('NAVSIM_NO_RULE')))
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
g = torch.Generator(device="cuda:0"); g.manual_seed(77)
K, Wm = 400, 50
acts = torch.rand((K + Wm, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
acts[..., 0] *= 1.0; acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * 2.0
sim.t["scan_noise_std"].fill_(0.01); sim.cfg.add_scan_noise = 1

def run(t):
    sim.io.action = acts[t].data_ptr()
    sim._reorder()
    sim.launch_step(reorder=False)
    sim.regen()

import contextlib
hi = torch.cuda.Stream(priority=-1) if os.environ.get("NAVSIM_DIAG_MAIN_HIGH") else None     # the steps on a high-priority stream
torch.cuda.synchronize()
ctx = torch.cuda.stream(hi) if hi is not None else contextlib.nullcontext()
ctx.__enter__()
for t in range(Wm):
    run(t)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for t in range(K):
        run(Wm + t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("pipeline %d: host enqueue %.1f us per step, wall %.1f us per step -> %.2f M env-steps/s; counters %s"
          % (P, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, E * K / (t2 - t0) / 1e6, sim.counters(reset=True)))
