"""Soak of the pipelined reset path: many steps, several periods; regen_late must stay 0 and every observation finite.
c5-shaped world (20 000 steps per period) and the reference-default world through the gym API (1 500 steps)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench, nav_gym_env
for P, no_rule in ((1, False), (2, False), (4, False), (8, False), (2, True), (4, True), (8, True)):
    wl = dict(bench.WORKLOADS["c5"]); wl.update(field="u16t", indoor_ratio=0.0, pregen=True, pipeline=P, install=True, no_rule=no_rule)
    cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
    E = cfg.n_envs
    g = torch.Generator(device="cuda:0"); g.manual_seed(7 + P)
    acts = torch.rand((512, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * 2.0
    t0 = time.perf_counter()
    N = 20000
    for t in range(N):
        sim.io.action = acts[t % 512].data_ptr()
        sim.launch_step(); sim.regen()
        if t % 5000 == 4999:
            torch.cuda.synchronize()
            assert torch.isfinite(sim.obs).all()
    torch.cuda.synchronize()
    c = sim.counters()
    print("c5, period %d%s: %d steps in %.1f s, %.2f M env-steps/s; served %d short %d late %d unserved %d"
          % (P, ", no rule" if no_rule else "", N, time.perf_counter() - t0, E * N / (time.perf_counter() - t0) / 1e6, c["regen_served"],
             c["regen_short"], c["regen_late"], c["regen_unserved"]))
    assert c["regen_late"] == 0 or no_rule
    assert c["regen_short"] == 0 or not no_rule
    del sim, arrays; torch.cuda.empty_cache()
for P, min_steps in ((2, 8), (4, 16), (4, 0), (8, 0)):
    env = nav_gym_env.make("NavGym-v0", num_envs=512, map_size="reference", randomize_maps=True, device="cuda:0", seed=99,
                           regen_min_steps=min_steps, pregen_pipeline=P)
    env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    acts = torch.rand((256, 512, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    for t in range(1500):
        obs, rew, done, info = env.step(acts[t % 256])
        if t == 700:
            env.reset()
    torch.cuda.synchronize()
    assert torch.isfinite(obs["observation"]).all()
    c = env.counters()
    print("reference defaults, 512 arenas, period %d, regen_min_steps %d: 1500 steps (+ a reset in the middle); %s" % (P, min_steps, c))
    assert c["regen_late"] == 0 or min_steps == 0
    env.close(); del env; torch.cuda.empty_cache()
print("soak ok")
