import os, sys, time, gc, cProfile, pstats
sys.path.insert(0, "/root/repo/nav-gym_amd")
import torch, nav_gym_env
def once(tag, prof=False):
    env = nav_gym_env.make("NavGym-v0", num_envs=4096, map_size="reference", randomize_maps=True, device="cuda:0", seed=1234)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if prof:
        pr = cProfile.Profile(); pr.enable()
    env.reset(); torch.cuda.synchronize()
    if prof:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    print(tag, "first reset %.0f ms" % ((time.perf_counter() - t1) * 1e3), "reserved GB", torch.cuda.memory_reserved() / 2**30)
    env.close(); del env
torch.zeros(1, device="cuda:0")
once("A")
t=time.perf_counter(); gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize(); print("free %.0f ms" % ((time.perf_counter()-t)*1e3), "reserved GB", torch.cuda.memory_reserved() / 2**30)
once("B", prof=True)
