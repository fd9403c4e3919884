"""GPU: what ends an episode in a batch (include/navsim.h NAVSIM_AUTORESET_*, ABI 6) -- the terminal observation of the
reference (env.py:700-728) under same-step auto-reset (io.final_obs), gymnasium's next-step auto-reset (io.reset_mask),
reset() of some arenas (navsim_restart) -- through the C ABI, against the reference's own traces and the CPU oracle."""
import numpy as np
import pytest

import ref
from helpers import load_trace
from nav_gym_amd import abi
from test_autoreset_oracle import trace_world, check_terminal_row

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from nav_gym_amd import lib, sim, world
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    lib.load()
    return type("G", (), dict(torch=torch, lib=lib, sim=sim, world=world, dev=torch.device("cuda:0")))


def _t(gpu, a, dtype=None):
    t = gpu.torch.from_numpy(np.ascontiguousarray(a)).to(gpu.dev)
    return t if dtype is None else t.to(dtype)


def _eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        diff = np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))
        raise AssertionError("%s: %d mismatches, max |diff| %.3e, first at %s" % (what, len(bad), diff, bad[0]))


def _np(d):
    return {k: v.cpu().numpy() for k, v in d.items()}


@pytest.mark.parametrize("name", ["crash_S3", "success_S2"])
@pytest.mark.parametrize("mode", [abi.AUTORESET_SAME_STEP, abi.AUTORESET_NEXT_STEP])
@pytest.mark.parametrize("E", [1, 5])
def test_terminal_observation_of_the_reference_traces_on_the_device(gpu, name, mode, E):
    """The rows with which the reference's crash / success traces end their episode -- what its step() returned with
    done = True, after the crash the re-scan at the reverted pose with the stack of S = 3 -- come back bit for bit as
    io.final_obs (same-step restart) resp. as the observation itself (next-step restart), alone and inside a batch whose
    other arenas drive differently (arena 2 of 5 replays the trace)."""
    tr = load_trace(name)
    cfg, arrays, occ = trace_world(tr, gpu.lib.default_config, lambda o: gpu.sim.build_dt(_t(gpu, o)).cpu().numpy(), mode)
    me = 2 if E > 1 else 0
    cfg.n_envs = E
    arrays = {k: (np.repeat(v, E, axis=0) if (isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1 and k not in ("scan_threshold", "scan_discomfort")) else v)
              for k, v in arrays.items()}
    arrays["scan_noise_std"] = np.zeros(E, np.float32)
    g = gpu.sim.NavSim(cfg, arrays, final_obs=True)
    host = dict(arrays); host["field"] = ref.build_dt(np.repeat(occ[None], E, axis=0))
    r = ref.RefSim(cfg, host)
    first = g.reset_obs().cpu().numpy()
    _eq(first, r.reset_obs(), "first observations")
    T = int(np.argmax(tr["done"] != 0)) + 1
    S, B = int(tr["S"]), int(tr["B"])
    rng = np.random.default_rng(3)
    for t in range(T + 2):
        act = np.stack([rng.uniform(0.0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        cmd = np.repeat(tr["ped_cmd"][min(t, T - 1)][None], E, axis=0)
        if t < T:
            act[me] = tr["actions"][t]
        g.set_ped_cmd(cmd); r.set_ped_cmd(cmd)
        obs, out = g.step(act)
        ro, rout = r.step(act)
        obs = obs.cpu().numpy(); out = _np(out)
        for k in rout:
            _eq(out[k], rout[k], "%s at step %d" % (k, t))
        _eq(obs, ro, "observations at step %d" % t)
        done = rout["done"] != 0
        if mode == abi.AUTORESET_SAME_STEP and done.any():
            fin = _np(g.final)
            _eq(fin["final_obs"][done], r.final["final_obs"][done], "terminal rows at step %d" % t)
            _eq(fin["final_goals"][done], r.final["final_goals"][done], "terminal goals at step %d" % t)
        if t < T:
            assert out["done"][me] == tr["done"][t] and out["is_crash"][me] == tr["is_crash"][t] and out["is_success"][me] == tr["is_success"][t], t
            assert abs(out["reward"][me] - tr["reward"][t]) < 1e-9, t
        if t == T - 1:
            if mode == abi.AUTORESET_SAME_STEP:
                fin = _np(g.final)
                check_terminal_row(tr, fin["final_obs"][me], fin["final_goals"][me], t)
                assert np.array_equal(obs[me, S * B:], first[me, S * B:])          # the row: the next episode's first observation
            else:
                check_terminal_row(tr, obs[me], None, t)
        if t == T and mode == abi.AUTORESET_NEXT_STEP:
            assert g.reset_flags[me].item() == 1 and out["done"][me] == 0 and out["reward"][me] == 0.0
            assert np.array_equal(obs[me, S * B:], first[me, S * B:])


def _pair(gpu, cfg, occ, n_peds, final_obs=False, **world_kw):
    from nav_gym_amd import robots
    arrays = gpu.world.make_world(cfg, occ, n_peds=n_peds, device=gpu.dev, **world_kw)
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)
    g = gpu.sim.NavSim(cfg, arrays, final_obs=final_obs)
    r = ref.RefSim(cfg, host)
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    return g, r


def _actions(rng, cfg, t):
    E = cfg.n_envs
    act = np.stack([rng.uniform(0.0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
    if t % 7 == 3:
        act[:, 0] = 0.5; act[:, 1] = 0.0          # bursts of straight driving provoke crashes
    return act


def _state_eq(g, r, what, skip=()):
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index", "counters") + tuple(skip):
            _eq(gs[k], v, "state %s %s" % (k, what))


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T, "u16t-rects-from-global-memory"])
@pytest.mark.parametrize("mode,ped_model,S,block", [(abi.AUTORESET_SAME_STEP, abi.PED_NONE, 3, 0), (abi.AUTORESET_SAME_STEP, abi.PED_SFM, 2, 0),
                                                    (abi.AUTORESET_NEXT_STEP, abi.PED_NONE, 2, 0), (abi.AUTORESET_NEXT_STEP, abi.PED_SFM, 3, 0),
                                                    (abi.AUTORESET_NEXT_STEP, abi.PED_EXTERNAL, 1, 256), (abi.AUTORESET_SAME_STEP, abi.PED_NONE, 2, 256),
                                                    (abi.AUTORESET_SAME_STEP, abi.PED_EXTERNAL, 1, 64)])
def test_autoreset_modes_vs_oracle(gpu, mode, ped_model, S, block, fmt):
    """48 arenas x 70 steps on 240 x 240 maps: every output, every observation row, the terminal rows of the arenas that
    finish (same-step: final_obs / final_goals; crashes with a stack of three among them) and every state array equal the
    oracle's bit for bit in both restart modes, for every field form and the 64 / 256-thread kernels (parked rays)."""
    rect_lds = 0
    if fmt == "u16t-rects-from-global-memory":
        fmt, rect_lds = abi.FIELD_U16T, 1
    E, size, N = 48, 240, 8
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=S, ped_model=ped_model,
                                 auto_reset=mode, n_spawn=8, seed=4343, field_format=fmt, rect_lds=rect_lds, step_block=block)
    if block == 64:
        gpu.world.lidar_full_circle(cfg, 64)
    else:
        gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 4343)
    g, r = _pair(gpu, cfg, occ, 6, final_obs=(mode == abi.AUTORESET_SAME_STEP), min_goal_dist=1.5, max_goal_dist=4.0)
    rng = np.random.default_rng(5)
    crashes = ends = crash_ends = resets = 0
    for t in range(70):
        act = _actions(rng, cfg, t)
        if ped_model == abi.PED_EXTERNAL:
            cmd = np.stack([rng.uniform(0, 0.6, (E, N)), rng.uniform(-0.6, 0.6, (E, N))], axis=2)
            g.set_ped_cmd(cmd); r.set_ped_cmd(cmd)
        go, gout = g.step(gpu.torch.from_numpy(act).to(gpu.dev))
        ro, rout = r.step(act)
        gout = _np(gout)
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go.cpu().numpy(), ro, "obs at step %d" % t)
        done = rout["done"] != 0
        if mode == abi.AUTORESET_SAME_STEP and done.any():
            fin = _np(g.final)
            _eq(fin["final_obs"][done], r.final["final_obs"][done], "terminal rows at step %d" % t)
            _eq(fin["final_goals"][done], r.final["final_goals"][done], "terminal goals at step %d" % t)
            # a terminal row that ends in success is what compute_terminals judges `done` (after a crash the reference returns
            # the re-scan at the reverted pose, whose flags are clear: SURVEY.md 9.2 item 8)
            won = done & (rout["is_crash"] == 0)
            if won.any():
                rd = ref.reward_done(r.cfg, r.final["final_obs"][won], r.final["final_goals"][won][:, 2:], r.a["scan_threshold"], r.a["scan_discomfort"])
                assert (rd["done"] != 0).all() and (rd["is_success"] != 0).all()
        if mode == abi.AUTORESET_NEXT_STEP:
            _eq(g.reset_flags.cpu().numpy(), r.reset_flags, "reset flags at step %d" % t)
            resets += int(r.reset_flags.sum())
        crashes += int(rout["is_crash"].sum()); ends += int(done.sum()); crash_ends += int((done & (rout["is_crash"] != 0)).sum())
        if t % 10 == 9:
            _state_eq(g, r, "at step %d" % t)
    assert ends > 10 and crash_ends > 2, (ends, crash_ends)
    if mode == abi.AUTORESET_NEXT_STEP:
        assert resets >= ends - E


@pytest.mark.parametrize("fmt,ped_model,plan,defer", [(abi.FIELD_U16T, abi.PED_SFM, 0, 0), (abi.FIELD_U16T, abi.PED_SFM, 0, 1),
                                                      (abi.FIELD_F32, abi.PED_NONE, 0, 1), (abi.FIELD_U16T, abi.PED_SFM, 1, 0)])
def test_next_step_autoreset_with_navsim_regen(gpu, fmt, ped_model, plan, defer):
    """Next-step auto-reset in a world that draws a new map per episode: the arenas that finished in call t are reset in call
    t + 1 -- the step skips them (zero outputs), navsim_regen keyed on the same flags gives them their new world and first
    observation (cfg.defer_reset_scan: also to those beyond regen_cap that restart in place).  Device == oracle bit for bit."""
    E, size, N = 40, 200 + 60 * plan, 6
    cap = 2 if defer else 5
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=ped_model, n_spawn=6,
                                 auto_reset=abi.AUTORESET_NEXT_STEP, seed=19, field_format=fmt, regen_cap=cap, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=plan, regen_indoor_ratio=0.5 if fmt == abi.FIELD_U16T else 0.0,
                                 defer_reset_scan=defer)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 19)
    g, r = _pair(gpu, cfg, occ, 5, plan_paths=bool(plan) and ped_model != abi.PED_NONE)
    rng = np.random.default_rng(6)
    B = cfg.n_beams * cfg.n_scan_stack
    regenerated = capped = 0
    for t in range(50):
        act = _actions(rng, cfg, t)
        go, gout = g.step(gpu.torch.from_numpy(act).to(gpu.dev))
        ro, rout = r.step(act)
        go = go.cpu().numpy(); gout = _np(gout)
        for k in ("reward", "done", "is_success", "is_crash", "distance"):
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        flags = r.reset_flags != 0
        _eq(g.reset_flags.cpu().numpy() != 0, flags, "reset flags at step %d" % t)
        keep = ~flags if defer else np.ones(E, bool)          # (deferred: the rows of the arenas being reset come from regen)
        _eq(go[keep], ro[keep], "obs at step %d" % t)
        go2 = g.regen().cpu().numpy()
        ro2 = r.regen()
        _eq(go2, ro2, "obs after regen at step %d" % t)
        for k in ("achieved_goal", "desired_goal"):
            _eq(g.out[k].cpu().numpy(), r.out[k], "%s after regen at step %d" % (k, t))
        n = int(flags.sum())
        regenerated += min(n, cap); capped += n > cap
        if n:
            _state_eq(g, r, "after regen at step %d" % t)
            if fmt == abi.FIELD_F32:
                _eq(g.numpy_state("field")["field"], r.a["field"], "field after regen at step %d" % t)
    assert regenerated > 5 and (capped > 0 or not defer), (regenerated, capped)


@pytest.mark.parametrize("period,min_steps,slow,E,block,beside", [(1, 0, False, 48, 0, False), (2, 0, True, 48, 0, False), (2, 8, False, 48, 0, False),
                                                                  (3, 0, True, 48, 256, False), (2, 0, False, 5, 0, False), (4, 0, False, 48, 64, False),
                                                                  (2, 0, True, 48, 0, True), (3, 0, True, 48, 256, True), (1, 0, False, 5, 0, True)])
def test_next_step_autoreset_with_the_pipelined_reset_path(gpu, monkeypatch, period, min_steps, slow, E, block, beside):
    """Next-step auto-reset with worlds staged ahead (navsim_step_install): the workgroup of an arena that is reset installs
    its staged world -- at the FRONT of the launch, in place of a step -- or, when the world is not staged yet, starts in
    place and flags the arena for the caller's navsim_regen (no rule) / restarts in place (cfg.regen_min_steps).  The rollout
    equals the oracle's step + synchronous navsim_regen_cpu keyed on the same flags, bit for bit, whatever the passes' timing.
    beside: navsim_step_install_next -- lateness decided when the episode ends, the late arenas regenerated on a third stream
    BESIDE the launch that resets them (NavSim.enable_pregen(late_beside=True); measured slower, kept as an entry point)."""
    size, N = 200, 6
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=abi.AUTORESET_NEXT_STEP, seed=31, field_format=abi.FIELD_U16T, regen_cap=E, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=0, regen_indoor_ratio=0.0, regen_min_steps=min_steps, step_block=block)
    if block == 64:
        gpu.world.lidar_full_circle(cfg, 64)
    else:
        gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 31)
    g, r = _pair(gpu, cfg, occ, 5)
    g.enable_pregen(pipeline=period, install=True, late_beside=beside)
    assert (g.late2 is not None) == beside
    if slow:
        stage = g.lib.navsim_regen_stage
        def delayed(*a, _stage=stage, _g=g):
            with gpu.torch.cuda.stream(_g.side):
                gpu.torch.cuda._sleep(200_000_000)
            return _stage(*a)
        monkeypatch.setattr(g.lib, "navsim_regen_stage", delayed)
    # ... and no pass at all is queued in two windows of the rollout: whoever ends two episodes inside one finds nothing staged
    # the second time, whatever the box's timing (the requests wait in mark[] for the first pass behind the window)
    hold = {"on": False}
    queue = g._queue_pass
    monkeypatch.setattr(g, "_queue_pass", lambda *a: None if hold["on"] else queue(*a))
    rng = np.random.default_rng(14)
    n_reset = 0
    for t in range(110):
        hold["on"] = slow and (28 <= t <= 36 or 58 <= t <= 66)
        act = _actions(rng, cfg, t)
        go, gout = g.step(gpu.torch.from_numpy(act).to(gpu.dev))
        ro, rout = r.step(act)
        go = go.cpu().numpy(); gout = _np(gout)
        for k in ("reward", "done", "is_success", "is_crash", "distance"):
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        flags = r.reset_flags != 0
        _eq(go[~flags], ro[~flags], "obs at step %d" % t)      # (an installed arena's row is already its new world's)
        n_reset += int(flags.sum())
        go2 = g.regen().cpu().numpy()
        ro2 = r.regen()
        _eq(go2, ro2, "obs after regen at step %d" % t)
        for k in ("achieved_goal", "desired_goal"):
            _eq(g.out[k].cpu().numpy(), r.out[k], "%s after regen at step %d" % (k, t))
        if flags.any():
            gpu.torch.cuda.synchronize()
            _state_eq(g, r, "after regen at step %d" % t, skip=("ped_waypoints",))
        if min_steps == 0 and E >= 8 and t in (30, 32, 60, 62):
            # two episode ends two calls apart (the robot put on its goal right after its reset): the second finds nothing staged
            idx = gpu.torch.arange(2, 8, device=gpu.dev)
            g.t["robot_goal"][idx] = g.t["robot_pose"][idx, :2]
            r.a["robot_goal"][2:8] = r.a["robot_pose"][2:8, :2]
    cg, cr = g.counters(), r.counters()
    assert n_reset > 8 or E < 8
    assert cg["regen_unserved"] == 0
    # which form served the late arenas: the arena's own workgroup (regen_lone: no fallback launch) wherever the library offers it
    assert g.lone == (min_steps == 0 and not beside and block != 64), (g.lone, block)
    # (a late arena is counted twice on the device: regen_late by the launch that found nothing staged, regen_served by the
    #  navsim_regen that then generated its world on the spot)
    assert cg["regen_short"] == cr["regen_short"] and cg["regen_served"] == cr["regen_served"], (cg, cr)
    if slow and E >= 8:
        assert cg["regen_late"] > 0


def test_reset_of_some_arenas_on_the_device(gpu):
    """NavSim.reset_arenas(mask) = navsim_restart + navsim_reset_obs(mask) (+ navsim_regen for worlds that draw maps): the
    masked arenas start a new episode, the others keep state and rows; device == oracle."""
    E, size = 24, 160
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=4, ped_model=abi.PED_SFM, n_spawn=6, n_scan_stack=2,
                                 auto_reset=abi.AUTORESET_NONE, seed=77)
    gpu.world.lidar_full_circle(cfg, 180)
    occ = gpu.world.make_maps(E, size, 77)
    g, r = _pair(gpu, cfg, occ, 3)
    rng = np.random.default_rng(8)
    for t in range(30):
        act = _actions(rng, cfg, t)
        go, gout = g.step(gpu.torch.from_numpy(act).to(gpu.dev))
        ro, rout = r.step(act)
        _eq(go.cpu().numpy(), ro, "obs at step %d" % t)
        if t % 6 == 5:
            mask = (rout["done"] != 0) | (rng.uniform(size=E) < 0.2)      # the finished arenas and a few others
            rows = g.reset_arenas(gpu.torch.from_numpy(mask).to(gpu.dev)).cpu().numpy()
            _eq(rows, r.restart(mask), "rows after the reset of some arenas at step %d" % t)
            _eq(g.out["done"].cpu().numpy(), r.out["done"], "done flags after the reset")
            _state_eq(g, r, "after the reset of some arenas at step %d" % t)


def test_env_autoreset_modes_and_reset_mask(gpu):
    """NavGymEnv(autoreset_mode=...): info['final_observation'] / info['final_mask'] (same-step), info['reset_mask'] and the
    terminal row as the observation (next-step), reset(mask); the two modes give every arena the same episodes (no
    pedestrians: an arena's rollout depends on its own actions only)."""
    torch = gpu.torch
    from nav_gym_amd import registry
    kw = dict(num_envs=16, map_size=160, n_beams=128, pedestrian_model="none", num_humans=0, seed=5, plan_paths=False,
              min_goal_dist=1.5, max_goal_dist=3.0)
    a = registry.make("NavGym-v0", **kw)
    b = registry.make("NavGym-v0", autoreset_mode="next_step", **kw)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa["observation"], ob["observation"])
    E, K = 16, 150
    rng = np.random.default_rng(4)
    acts = np.stack([rng.uniform(0.3, 0.5, (K, E)), rng.uniform(-0.3, 0.3, (K, E))], axis=2)
    rec = []
    for k in range(K):
        o, rew, done, info = a.step(acts[k])
        assert set(("final_observation", "final_mask")) <= set(info) and info["final_mask"].dtype == torch.bool
        rec.append((o["observation"].clone(), rew.clone(), done.clone(), info["final_observation"]["observation"].clone(),
                    info["final_observation"]["achieved_goal"].clone(), o["achieved_goal"].clone()))
    assert sum(int(x[2].sum()) for x in rec) >= E
    kb = np.zeros(E, int)
    awaiting = np.zeros(E, bool)
    for tau in range(K):
        act = np.stack([acts[min(kb[e], K - 1), e] for e in range(E)])
        o, rew, done, info = b.step(act)
        assert np.array_equal(info["reset_mask"].cpu().numpy(), awaiting) and "final_observation" not in info
        for e in range(E):
            if awaiting[e]:
                ra = rec[kb[e] - 1]
                assert torch.equal(o["observation"][e], ra[0][e]) and float(rew[e]) == 0.0 and not bool(done[e])
                continue
            if kb[e] >= K:
                continue
            ra = rec[kb[e]]
            assert float(rew[e]) == float(ra[1][e]) and bool(done[e]) == bool(ra[2][e])
            if bool(ra[2][e]):
                assert torch.equal(o["observation"][e], ra[3][e]) and torch.equal(o["achieved_goal"][e], ra[4][e])
            else:
                assert torch.equal(o["observation"][e], ra[0][e])
            kb[e] += 1
        awaiting = done.cpu().numpy().copy()
    # reset(mask): only those arenas start anew
    c = registry.make("NavGym-v0", auto_reset=False, **kw)
    c.reset()
    for k in range(5):
        o, _, done, info = c.step(acts[k])
    assert "final_observation" not in info
    before = o["observation"].clone()
    ep = c.sim.t["episode"].clone()
    mask = np.zeros(E, bool); mask[[1, 7, 8]] = True
    o2 = c.reset(mask)
    keep = torch.from_numpy(~mask).to(before.device)
    assert torch.equal(o2["observation"][keep], before[keep]) and not torch.equal(o2["observation"][~keep], before[~keep])
    assert torch.equal(c.sim.t["episode"].cpu(), ep.cpu() + torch.from_numpy(mask.astype(np.int64)))
    for e_ in (a, b, c):
        e_.close()


@pytest.mark.parametrize("name", ["c5", "c4"])
def test_eight_shards_equal_one_world_at_the_configured_totals(gpu, name):
    """SURVEY.md 8e's parity clause at BASELINE's own totals, on one GPU: c4's 16 384 arenas resp. c5's 4 096 as the EIGHT
    shards an 8-GPU job runs (2048 / 512 arenas each, env_index_base = 0, 2048, ... / 0, 512, ...), built and stepped one
    after the other exactly as bench.py builds a rank's share, against ONE world holding all of them: every output of
    every step and the final state, bit for bit, in global arena order -- the order in which sharding.RowGather delivers
    the shards' rows (tests/test_host_logic.py runs that gather with eight gloo ranks at these shard sizes)."""
    import bench
    torch = gpu.torch
    wl = dict(bench.WORKLOADS[name]); wl["field"] = "u16t"
    G, per, total = 8, wl["envs"], wl["total"]
    assert G * per == total
    regen = bool(wl.get("regen"))
    steps = 6
    g = torch.Generator(device=gpu.dev); g.manual_seed(21)
    acts = torch.rand((steps, total, 2), generator=g, device=gpu.dev, dtype=torch.float64)
    lin_hi, rot_hi = (1.0, 2.0) if wl.get("robot") == "husky" else (0.5, 0.64)
    acts[..., 0] *= lin_hi; acts[..., 1] = (acts[..., 1] * 2.0 - 1.0) * rot_hi
    skip = ("arena_cost", "launch_order", "counters", "regen_ws", "beam_table", "scan_threshold", "scan_discomfort")
    shared = ("scan_threshold", "scan_discomfort")
    big, rows, finals = {}, [[] for _ in range(steps)], {}
    cfg_big = None
    for r in range(G):                                           # the shards, one after the other
        cfg, sim, arrays, occ = bench.build_sim(wl, r * per, per)
        assert cfg.env_index_base == r * per and cfg.n_envs == per
        if r == 0:
            cfg_big = cfg.copy(); cfg_big.n_envs = total; cfg_big.env_index_base = 0
            if regen:
                cfg_big.regen_cap = G * cfg.regen_cap            # (a rank serves up to its own cap: the one world must serve them all)
        for k, v in sim.t.items():                               # its initial state -> its slice of the one world
            if k in skip and k not in shared:
                continue
            if k in shared:
                big.setdefault(k, v.clone())
                continue
            if k not in big:
                lead = v.numel() // per if v.dim() == 1 and k == "field" else None
                shape = (total * lead,) if lead else (total,) + tuple(v.shape[1:])
                big[k] = torch.empty(shape, dtype=v.dtype, device=v.device)
            if v.dim() == 1 and k == "field":
                n = v.numel()
                big[k][r * n:(r + 1) * n].copy_(v)
            else:
                big[k][r * per:(r + 1) * per].copy_(v)
        big.setdefault("obs0", torch.empty((total, sim.obs.shape[1]), dtype=sim.obs.dtype, device=gpu.dev))[r * per:(r + 1) * per].copy_(sim.obs)
        if regen and r == 0:
            on_goal = torch.as_tensor([2, 3, 100, 300, 511], device=gpu.dev)
        for t in range(steps):
            if regen and t in (0, 3):                            # c5: a few arenas of EVERY shard finish and get new worlds
                sim.t["robot_goal"][on_goal] = sim.t["robot_pose"][on_goal, :2]
            o, out = sim.step(acts[t, r * per:(r + 1) * per])
            if regen:
                assert int(out["done"].sum()) <= cfg.regen_cap
                o = sim.regen()
            rows[t].append((o.clone(), {k: v.clone() for k, v in sim.out.items()}))
        for k in ("robot_pose", "robot_goal", "prev_pose", "prev_action", "episode", "steps", "n_hist", "n_peds", "ped_pose", "ped_vel",
                  "ped_dist", "spawn_pose", "spawn_goal"):
            if k in sim.t:
                finals.setdefault(k, []).append(sim.t[k].clone())
        sim.close()
        del sim, arrays, occ
        torch.cuda.empty_cache()
    obs0 = big.pop("obs0")
    one = gpu.sim.NavSim(cfg_big, big)
    del big
    torch.cuda.empty_cache()
    one.obs_buf[one.cur].copy_(obs0)
    for t in range(steps):
        if regen and t in (0, 3):
            idx = torch.cat([on_goal + r * per for r in range(G)])
            one.t["robot_goal"][idx] = one.t["robot_pose"][idx, :2]
        o, out = one.step(acts[t])
        if regen:
            assert int(out["done"].sum()) <= cfg_big.regen_cap
            o = one.regen()
            out = one.out
        assert torch.equal(o, torch.cat([x[0] for x in rows[t]])), "observations at step %d" % t
        for k in out:
            assert torch.equal(out[k], torch.cat([x[1][k] for x in rows[t]])), "%s at step %d" % (k, t)
    for k, parts in finals.items():
        assert torch.equal(one.t[k], torch.cat(parts)), "state %s after %d steps" % (k, steps)
    if regen:
        assert int(one.t["episode"].sum().item()) >= 2 * G       # worlds WERE regenerated, in every shard
    one.close()
    del one
    torch.cuda.empty_cache()
