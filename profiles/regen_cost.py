"""Cost of navsim_regen inside env.step(): in-place respawn vs new map vs new map + path planning.
Run on the GPU box: python profiles/regen_cost.py   (writes gpurun_out/regen_cost.json)"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nav-gym_amd"))
import torch
import nav_gym_env

rows = []
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, kw in (("respawn", dict(randomize_maps=False)),
                 ("regen", dict(randomize_maps=True, plan_paths=False)),
                 ("regen+plan", dict(randomize_maps=True, plan_paths=True))):
    if only and name != only:
        continue
    env = nav_gym_env.make('NavGym-v0', num_envs=4096, n_beams=1081, map_size=500, pedestrian_model='sfm',
                           num_humans=20, device='cuda:0', seed=0, **kw)
    env.reset()
    act = torch.zeros((4096, 2), dtype=torch.float64, device='cuda:0')
    act[:, 0] = 0.5                                   # straight ahead: steady stream of crashes / successes
    done = 0
    for _ in range(20):
        _, _, d, _ = env.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        _, _, d, _ = env.step(act)
        done += d.sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    rows.append({"mode": name, "ms_per_step": dt * 1e3, "episodes_finished_per_step": float(done) / 200})
    print(rows[-1], flush=True)
    del env
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/regen_cost.json", "w"), indent=1)
