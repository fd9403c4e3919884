"""What would rect records give c5's STEP kernel?  c5 world with the records on (navsim_regen then rebuilds them with the
verified builder -- slow, that is why c5 runs without) and off: HIP events around the step launch only."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench

def run(rects, rect_lds):
    wl = dict(bench.WORKLOADS["c5"]); wl["field"] = "u16t"; wl["rects"] = rects
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    sim.cfg.rect_lds = rect_lds
    E = cfg.n_envs
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    T = 80
    acts = torch.rand((T, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    step_us, regen_us = [], []
    for t in range(T):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        sim.io.action = acts[t].data_ptr()
        a.record(); sim.launch_step(); b.record(); sim.regen(); c.record()
        torch.cuda.synchronize()
        if t >= 20:
            step_us.append(a.elapsed_time(b) * 1e3); regen_us.append(b.elapsed_time(c) * 1e3)
    return dict(rects=rects, rect_lds=rect_lds, step_us=sum(step_us) / len(step_us), regen_us=sum(regen_us) / len(regen_us))

for rects, lds in ((False, 0), (True, 1), (True, 2), (False, 0), (True, 1), (True, 2)):
    print(json.dumps(run(rects, lds)))
