"""The reference's README usage loop (README.md:47-55) on this build, then the same loop batched.

    python examples/usage.py            # needs a MI355X; there is no CPU fallback
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nav-gym_amd"))
import nav_gym_env  # noqa: E402  registers 'NavGym-v0'

# ---- exactly the reference's loop ---------------------------------------------------------------------
env = nav_gym_env.make("NavGym-v0")
obs = env.reset()
done, steps = False, 0
while not done and steps < 200:
    action = env.action_space.sample()            # your agent code here
    obs, reward, done, info = env.step(action)
    steps += 1
print("single arena: %d steps, last reward %.3f, info %s" % (steps, reward, info))
img = env.render(mode="rgb_array")                # env.py:833-1050: float32 BGR [800, 800, 3] (no window here)
print("render:", img.shape, img.dtype)

# ---- 4096 arenas per call --------------------------------------------------------------------------------
import torch  # noqa: E402

E = 4096
benv = nav_gym_env.make("NavGym-v0", num_envs=E, n_beams=1081, map_size=500, pedestrian_model="sfm", num_humans=10)
obs = benv.reset()
actions = torch.rand((E, 2), dtype=torch.float64, device="cuda:0")
actions[:, 0] *= 0.5
actions[:, 1] = actions[:, 1] * 1.28 - 0.64
for _ in range(20):
    benv.step(actions)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
finished = 0
for _ in range(n):
    obs, reward, done, info = benv.step(actions)      # your batched agent here
    finished += done.sum()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d arenas: %.2f M env-steps/s, %d episodes finished, observation %s on %s"
      % (E, E * n / dt / 1e6, int(finished), tuple(obs["observation"].shape), obs["observation"].device))

# ---- the reference's own world, batched: every registered default, a new map at every episode end ----------
# (512 beams, 1000 x 1000 corridor / 400 x 400 outdoor maps, 5-15 pedestrians on planned routes).  For such worlds the
# env stages every arena's next world ahead of time on a side stream (pregen_pipeline=8 by default: the same rollout
# as pregen_pipeline=0, bit for bit, about 2.4 x the throughput); env.counters() says how many arenas were served.
E = 1024
renv = nav_gym_env.make("NavGym-v0", num_envs=E, map_size="reference", randomize_maps=True)
renv.reset()
actions = actions[:E].contiguous()
for _ in range(20):
    renv.step(actions)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    obs, reward, done, info = renv.step(actions)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("reference defaults, %d arenas: %.2f M env-steps/s (pregen_pipeline=%d); %s" % (E, E * 100 / dt / 1e6, renv.pregen_pipeline, renv.counters()))
