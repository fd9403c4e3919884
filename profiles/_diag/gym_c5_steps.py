"""A c5-shaped world through the gym API (512 arenas, 1081 beams, 500 x 500 outdoor maps, 20 pedestrians, a new map per episode):
the env's default launch form against the pipelined reset path.  NAVSIM_PIPELINE=P, NAVSIM_PLAN=0|1, NAVSIM_ENVS"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, nav_gym_env
E = int(os.environ.get("NAVSIM_ENVS", "512"))
kw = dict(pregen_pipeline=int(os.environ.get("NAVSIM_PIPELINE", "0")), plan_paths=os.environ.get("NAVSIM_PLAN", "0") == "1")
env = nav_gym_env.make("NavGym-v0", num_envs=E, n_beams=1081, map_size=500, indoor_ratio=0.0, randomize_maps=True,
                       pedestrian_model="sfm", num_humans=20, device="cuda:0", seed=1234, **kw)
env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(78)
K, Wm = 300, 50
acts = torch.rand((K + Wm, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for t in range(Wm):
    env.step(acts[t])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(K):
    env.step(acts[Wm + t])
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("c5-shaped world through the gym API, %d arenas, plan_paths %s, pregen_pipeline %d, graphs %s: %.2f M env-steps/s, %.1f us per step; %s"
      % (E, kw["plan_paths"], env.pregen_pipeline, env._graphed, E * K / el / 1e6, el / K * 1e6, env.counters()))
