"""Does an environment that replays hipGraphs, run and dropped earlier in the process, slow a later pipelined environment down?
(bench.py's default run measures the graphed gym windows before the pipelined ones.)  NAVSIM_FIRST=c2|c3|refdef0|none"""
import os, sys, runpy
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, nav_gym_env
first = os.environ.get("NAVSIM_FIRST", "c3")
if first != "none":
    if first == "refdef0_4096":
        env = nav_gym_env.make("NavGym-v0", num_envs=4096, map_size="reference", randomize_maps=True, device="cuda:0", seed=1, pregen_pipeline=0)
    elif first == "refdef0":
        env = nav_gym_env.make("NavGym-v0", num_envs=1024, map_size="reference", randomize_maps=True, device="cuda:0", seed=1, pregen_pipeline=0)
    elif first == "refdef0_graphs":
        env = nav_gym_env.make("NavGym-v0", num_envs=1024, map_size="reference", randomize_maps=True, device="cuda:0", seed=1, pregen_pipeline=0, use_graphs=True)
    else:
        kw = dict(n_beams=1081, map_size=500, indoor_ratio=0.0)
        if first == "c2":
            kw.update(pedestrian_model="none", num_humans=0)
        else:
            kw.update(pedestrian_model="sfm", num_humans=20)
        env = nav_gym_env.make("NavGym-v0", num_envs=4096, device="cuda:0", seed=1, randomize_maps=(first == "c3r"), **kw)
    env.reset()
    a = torch.zeros((env.num_envs, 2), dtype=torch.float64, device="cuda:0")
    for t in range(60):
        env.step(a)
    torch.cuda.synchronize()
    print("first: %s, graphs %s" % (first, env._graphed))
    env.close(); del env
    torch.cuda.empty_cache()
runpy.run_path(os.path.join(ROOT, "profiles", "_diag", "gym_refdef_steps.py"), run_name="__main__")
