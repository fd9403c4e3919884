"""Where the first reset() of the reference's configuration spends its time (bench.py gym_api.reference_defaults reset_first_ms)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nav-gym_amd"))
import torch, nav_gym_env
E = int(os.environ.get("NAVSIM_ENVS", "4096"))
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
t0 = time.perf_counter()
env = nav_gym_env.make("NavGym-v0", num_envs=E, map_size="reference", randomize_maps=True, device="cuda:0", seed=1234)
t1 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
env.reset(); torch.cuda.synchronize()
pr.disable()
t2 = time.perf_counter()
print("make %.0f ms, first reset %.0f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
t3 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); print("second reset %.0f ms" % ((time.perf_counter() - t3) * 1e3))
