"""Which part of bench.py's default run slows the pipelined gym window down (1.6 M inside the run, 2.4 M in a fresh process)?
Runs bench.extras' pieces selectively: NAVSIM_PARTS = comma list of: main,c3,c4,c5,gym_c2,gym_c3,ref0   then the pipelined window."""
import os, sys, time, types
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench, nav_gym_env
parts = [p for p in os.environ.get("NAVSIM_PARTS", "").split(",") if p]
dev = "cuda:0"

def window(E, K, Wm, **kw):
    env = nav_gym_env.make("NavGym-v0", num_envs=E, device=dev, seed=1234, **kw)
    env.reset(); env.reset()
    g = torch.Generator(device=dev); g.manual_seed(78)
    acts = torch.rand((K + Wm, E, 2), generator=g, device=dev, dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    for t in range(Wm):
        env.step(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        env.step(acts[Wm + t])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    env.close(); del env; torch.cuda.empty_cache()
    return E * K / el / 1e6

def sim_window(name):
    wl = dict(bench.WORKLOADS[name]); wl["field"] = "u16t"; wl["indoor_ratio"] = 0.0
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    for t in range(230):
        sim._reorder(); sim.launch_step(reorder=False)
        if wl.get("regen"):
            sim.regen()
    torch.cuda.synchronize()
    if name == "c5":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(200):
                sim._reorder(); sim.launch_step(reorder=False); sim.regen()
        g.replay(); torch.cuda.synchronize(); del g
    del sim, arrays; torch.cuda.empty_cache()

c2_kw = dict(n_beams=1081, map_size=500, indoor_ratio=0.0)
for p in parts:
    if p in ("c2", "c3", "c4", "c5"):
        sim_window(p)
    elif p == "gym_c2":
        window(4096, 200, 30, pedestrian_model="none", num_humans=0, **c2_kw)
    elif p == "gym_c3":
        window(4096, 200, 30, pedestrian_model="sfm", num_humans=20, plan_paths=True, **c2_kw)
        window(4096, 200, 30, pedestrian_model="sfm", num_humans=20, plan_paths=False, **c2_kw)
    elif p == "ref0":
        window(1024, 100, 20, map_size="reference", randomize_maps=True, pedestrian_model="sfm", pregen_pipeline=0)
    elif p == "ref0_4096":
        window(4096, 100, 20, map_size="reference", randomize_maps=True, pedestrian_model="sfm", pregen_pipeline=0)
    print("done", p, flush=True)
print("parts %s -> pipelined window: %.3f M env-steps/s" % (parts, window(1024, 100, 20, map_size="reference", randomize_maps=True, pedestrian_model="sfm", pregen_pipeline=4)))
