import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nav-gym_amd"))
import torch
import nav_gym_env
env = nav_gym_env.make('NavGym-v0', num_envs=4096, n_beams=1081, map_size=500, pedestrian_model='sfm',
                       num_humans=20, device='cuda:0', seed=0)
env.reset()
act = torch.zeros((4096, 2), dtype=torch.float64, device='cuda:0'); act[:, 0] = 0.3; act[:, 1] = 0.2
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
print("env.step          host %.3f ms  total %.3f ms" % timeit(lambda: env.step(act)))
print("sim.step(action)  host %.3f ms  total %.3f ms" % timeit(lambda: env.sim.step(act)))
print("sim.launch_step   host %.3f ms  total %.3f ms" % timeit(lambda: env.sim.launch_step()))
env.cfg.add_scan_noise = 0; env.sim.cfg.add_scan_noise = 0
print("launch, no noise  host %.3f ms  total %.3f ms" % timeit(lambda: env.sim.launch_step()))
