"""c3-shaped world through the gym API with planned pedestrian routes (NavGymEnv default plan_paths=True): 230 calls of step().
Under rocprofv3 --kernel-trace --stats this shows what navsim_replan's launches cost per step (profiles/r04_replan/)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, nav_gym_env
E = int(os.environ.get("NAVSIM_ENVS", "4096"))
env = nav_gym_env.make("NavGym-v0", num_envs=E, n_beams=1081, map_size=500, indoor_ratio=0.0, device="cuda:0", seed=1234,
                       pedestrian_model="sfm", num_humans=20, plan_paths=os.environ.get("NAVSIM_PLAN", "1") == "1",
                       use_graphs={"": None, "0": False, "1": True}[os.environ.get("NAVSIM_GRAPHS", "")])
if os.environ.get("NAVSIM_BIG_FIRST"):
    from nav_gym_amd import sim as _sim
    _sim.NavSim.overlap_big_first = os.environ["NAVSIM_BIG_FIRST"] == "1"
env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(78)
K, Wm = 200, 30
acts = torch.rand((K + Wm, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for t in range(Wm):
    env.step(acts[t])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(K):
    env.step(acts[Wm + t])
host = time.perf_counter() - t0          # the host's own time for the K calls (the GPU runs behind)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("host time per step() call: %.1f us" % (host / K * 1e6))
print("gym API, c3 world: %.2f M env-steps/s, %.4f ms per step; counters %s" % (E * K / el / 1e6, el / K * 1e3, env.counters()))
