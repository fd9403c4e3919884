"""The pipelined reset path without a rule IS step + navsim_regen: two environments made with the same arguments, one with the
env's default (pregen_pipeline=8 for this world, fallback on), one with pregen_pipeline=0, are stepped side by side with the same
actions; observation, reward, done and the info flags are compared bit for bit at EVERY step.  NAVSIM_ENVS arenas (default 1024),
NAVSIM_STEPS steps (default 3000), a reset() of all arenas in the middle."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, nav_gym_env
E = int(os.environ.get("NAVSIM_ENVS", "1024")); N = int(os.environ.get("NAVSIM_STEPS", "3000"))
mk = lambda **kw: nav_gym_env.make("NavGym-v0", num_envs=E, map_size="reference", randomize_maps=True, device="cuda:0", seed=5, **kw)
for P in (None, 4, 2):
    a = mk() if P is None else mk(pregen_pipeline=P)
    b = mk(pregen_pipeline=0)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa["observation"], ob["observation"])
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    acts = torch.rand((256, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    bad = torch.zeros((), dtype=torch.int64, device="cuda:0")
    ends = torch.zeros((), dtype=torch.int64, device="cuda:0")
    t0 = time.perf_counter()
    for t in range(N):
        oa, ra, da, ia = a.step(acts[t % 256]); ob, rb, db, ib = b.step(acts[t % 256])
        bad += (oa["observation"] != ob["observation"]).any() | (ra != rb).any() | (da != db).any()
        for k in ("is_success", "is_crash"):
            if k in ia:
                bad += (torch.as_tensor(ia[k]) != torch.as_tensor(ib[k])).any()
        ends += da.sum()
        if t == N // 2:
            oa, ob = a.reset(), b.reset()
            bad += (oa["observation"] != ob["observation"]).any()
    torch.cuda.synchronize()
    ca, cb = a.counters(), b.counters()
    print("%d arenas, %d steps, pregen_pipeline=%s (env picked %d): steps that differ %d; %d episode ends; pipelined: regen_served %d, of them late "
          "(generated on the spot) %d, unserved %d; pregen_pipeline=0: regen_served %d; %.1f s"
          % (E, N, P, a.pregen_pipeline, int(bad), int(ends), ca["regen_served"], ca["regen_late"], ca["regen_unserved"],
             cb["regen_served"], time.perf_counter() - t0))
    assert int(bad) == 0 and ca["regen_unserved"] == 0
    a.close(); b.close(); del a, b; torch.cuda.empty_cache()
print("equal soak ok")
