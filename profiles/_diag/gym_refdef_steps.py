"""The reference's own configuration through the gym API (bench.py gym_api.reference_defaults): every registered default of
NavGym-v0, 512 beams, 1000 x 1000 arenas (corridor maps) / 400 x 400 outdoor maps, 5-15 pedestrians on planned routes, a new map
per episode.  Under rocprofv3 --kernel-trace --stats: what a step of that world is made of (profiles/r05_refdef/)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, nav_gym_env
E = int(os.environ.get("NAVSIM_ENVS", "1024"))
kw = {"pregen_pipeline": 0}                    # (the env's default for this world is 4: NAVSIM_PIPELINE chooses)
if os.environ.get("NAVSIM_GRAPHS"):
    kw["use_graphs"] = os.environ["NAVSIM_GRAPHS"] == "1"
if os.environ.get("NAVSIM_PIPELINE"):          # pipelined pre-generation: regen_min_steps = 4 P
    kw["pregen_pipeline"] = int(os.environ["NAVSIM_PIPELINE"]); kw["regen_min_steps"] = 4 * kw["pregen_pipeline"]
if os.environ.get("NAVSIM_STAGE_CAP"):
    kw["pregen_stage_cap"] = int(os.environ["NAVSIM_STAGE_CAP"])
if os.environ.get("NAVSIM_POLL"):               # the fallback's launches only when the step flagged somebody (round 6): 1 / 0 (default: the env's)
    kw["pregen_fallback_poll"] = os.environ["NAVSIM_POLL"] == "1"
if os.environ.get("NAVSIM_LANES"):              # staging passes alternating between two side streams (round 6): handed to NavSim.enable_pregen
    import nav_gym_amd.sim as _sim
    _orig = _sim.NavSim.enable_pregen
    def _ep(self, *a, _n=int(os.environ["NAVSIM_LANES"]), **k):
        k.setdefault("stage_lanes", _n)
        return _orig(self, *a, **k)
    _sim.NavSim.enable_pregen = _ep
if os.environ.get("NAVSIM_MIN_STEPS"):
    kw["regen_min_steps"] = int(os.environ["NAVSIM_MIN_STEPS"])
env = nav_gym_env.make("NavGym-v0", num_envs=E, map_size="reference", randomize_maps=True, device="cuda:0", seed=1234, **kw)
env.reset()
for _ in range(int(os.environ.get("NAVSIM_RESETS", "1")) - 1):
    env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(78)
K, Wm = int(os.environ.get("NAVSIM_STEPS", "100")), 20
acts = torch.rand((K + Wm, E, 2), generator=g, device="cuda:0", dtype=torch.float64)
acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
for t in range(Wm):
    env.step(acts[t])
torch.cuda.synchronize()
e0 = int(env.sim.t["episode"].sum().item())
t0 = time.perf_counter()
for t in range(K):
    env.step(acts[Wm + t])
host = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("host enqueue %.1f us per step, wall %.1f" % (host / K * 1e6, el / K * 1e6))
e1 = int(env.sim.t["episode"].sum().item())
print("gym API, reference defaults, %d arenas: %.3f M env-steps/s, %.4f ms per step, %.1f episodes ended per step; counters %s"
      % (E, E * K / el / 1e6, el / K * 1e3, (e1 - e0) / K, env.counters()))
