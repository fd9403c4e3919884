"""CPU: what ends an episode in a batch (include/navsim.h NAVSIM_AUTORESET_*, ABI 6), on the oracle.

The reference's step() returns the LAST observation of an episode together with done = True -- after a crash the
re-scan at the reverted pose (env.py:700-728) -- and leaves reset() to its caller (env.py:730).  A batch restarts
finished arenas itself; these tests pin what happens to that terminal observation against the reference's own
traces (tests/golden/golden_trace_crash_S3.npz, golden_trace_success_S2.npz end on `done`), and the two restart
modes against each other."""
import numpy as np
import pytest

import ref
from helpers import load_trace, trace_setup, outdoor_map
from nav_gym_amd import abi, robots


def trace_world(tr, default_config, build_dt, mode):
    """trace_setup + a one-entry spawn table (the trace's own start) so that the arena can restart."""
    cfg, arrays, occ = trace_setup(tr, default_config, build_dt)
    cfg.auto_reset = mode
    cfg.n_spawn = 1
    arrays["spawn_pose"] = tr["init_robot_pose"][None, None].copy()
    arrays["spawn_goal"] = tr["robot_goal"][None, None].copy()
    return cfg, arrays, occ


def check_terminal_row(tr, row, goals, t):
    S, B = int(tr["S"]), int(tr["B"])
    assert np.array_equal(row[: S * B], tr["obs_scan"][t]), "terminal scan stack"
    np.testing.assert_allclose(row[S * B:], tr["obs_tail"][t], rtol=0, atol=2e-6)
    if goals is not None:                                   # achieved_goal = the pose slots, desired_goal = the goal (env.py:455-461)
        np.testing.assert_allclose(goals[:2], tr["obs_tail"][t][2:4], rtol=0, atol=2e-6)
        np.testing.assert_allclose(goals[2:], tr["robot_goal"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["crash_S3", "success_S2"])
@pytest.mark.parametrize("mode", [abi.AUTORESET_SAME_STEP, abi.AUTORESET_NEXT_STEP])
def test_terminal_observation_of_the_reference_traces(name, mode):
    """The last rows of the reference's crash / success traces -- what its step() returned with done = True -- come back as
    final_obs under same-step auto-reset (the row itself then holds the new episode's first observation) and as the
    observation itself under next-step auto-reset (the reset happens in the following call)."""
    tr = load_trace(name)
    cfg, arrays, _ = trace_world(tr, ref.default_config, ref.build_dt, mode)
    sim = ref.RefSim(cfg, arrays)
    first = sim.reset_obs().copy()
    T = int(np.argmax(tr["done"] != 0)) + 1                   # up to the step that ends the episode (the reference's caller
    assert tr["done"][T - 1] and not tr["done"][: T - 1].any()   # steps on without a reset: every later row is `done` too)
    assert T > int(tr["S"])                                      # the stack is full by then
    for t in range(T):
        sim.set_ped_cmd(tr["ped_cmd"][t][None])
        obs, out = sim.step(tr["actions"][t][None])
        assert out["done"][0] == tr["done"][t] and out["is_crash"][0] == tr["is_crash"][t], t
        assert abs(out["reward"][0] - tr["reward"][t]) < 1e-9, t
    S, B = int(tr["S"]), int(tr["B"])
    if mode == abi.AUTORESET_SAME_STEP:
        check_terminal_row(tr, sim.final["final_obs"][0], sim.final["final_goals"][0], T - 1)
        # ... and the row is the first observation of the next episode: the trace's own start, scanned with the pedestrians
        # where they are now (none in these two traces moves into view)
        assert np.array_equal(obs[0, S * B:], first[0, S * B:])
        assert sim.a["episode"][0] == 1 and sim.a["steps"][0] == 0
    else:
        check_terminal_row(tr, obs[0], None, T - 1)
        assert sim.a["episode"][0] == 1 and sim.a["steps"][0] == 0          # the STATE has restarted ...
        np.testing.assert_array_equal(sim.a["robot_pose"][0], tr["init_robot_pose"])
        sim.set_ped_cmd(np.zeros_like(tr["ped_cmd"][0])[None])
        obs2, out2 = sim.step(np.array([[0.5, 0.3]]))                        # ... and the next call resets: action ignored
        assert out2["done"][0] == 0 and out2["reward"][0] == 0.0 and sim.reset_flags[0] == 1
        assert np.array_equal(obs2[0, S * B:], first[0, S * B:])
        np.testing.assert_array_equal(sim.a["robot_pose"][0], tr["init_robot_pose"])
        assert sim.a["steps"][0] == 0
        _, out3 = sim.step(tr["actions"][0][None])                          # and then the arena steps again
        assert sim.reset_flags[0] == 0 and sim.a["steps"][0] == 1


def _static_world(E, size, seed, mode, S=2, beams=90):
    from nav_gym_amd import lib, world
    import torch
    cfg = lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=1, ped_model=abi.PED_NONE, n_spawn=4,
                             auto_reset=mode, n_scan_stack=S, seed=seed)
    world.lidar_full_circle(cfg, beams)
    occ = world.make_maps(E, size, seed)
    arrays = world.make_world(cfg, occ, n_peds=0, device="cpu", field=torch.from_numpy(ref.build_dt(occ)), min_goal_dist=1.0,
                              max_goal_dist=2.5, robot_clearance=0.8)
    host = {k: v.numpy() for k, v in arrays.items()}
    host["scan_threshold"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    host["scan_discomfort"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    return cfg, host


def test_next_step_autoreset_is_the_same_step_rollout_with_the_reset_a_call_later():
    """Without pedestrians an arena's rollout depends on its own actions only: with next-step auto-reset every arena lives
    the SAME episodes as with same-step auto-reset -- same rewards, flags and observations step for step -- except that
    the call that ends an episode returns the terminal row (= same-step's final_obs) and the NEXT call, which ignores the
    arena's action, returns the new episode's first row (= what same-step returned at once)."""
    E, size, K = 12, 100, 260
    cfg_a, host = _static_world(E, size, 5, abi.AUTORESET_SAME_STEP)
    cfg_b = cfg_a.copy(); cfg_b.auto_reset = abi.AUTORESET_NEXT_STEP
    a, b = ref.RefSim(cfg_a, host), ref.RefSim(cfg_b, host)
    assert np.array_equal(a.reset_obs(), b.reset_obs())
    rng = np.random.default_rng(2)
    acts = np.stack([rng.uniform(0.2, 0.5, (K, E)), rng.uniform(-0.64, 0.64, (K, E))], axis=2)
    acts[::5, :, 1] = 0.0                                        # straight bursts: crashes and goals
    rec = []                                                     # per real step of A: what every arena saw
    for k in range(K):
        oa, outa = a.step(acts[k])
        rec.append((oa.copy(), {n: v.copy() for n, v in outa.items()}, a.final["final_obs"].copy(), a.final["final_goals"].copy()))
    assert sum(int(r[1]["done"].sum()) for r in rec) >= 2 * E, "the rollout must end many episodes"
    assert sum(int(r[1]["is_crash"].sum()) for r in rec) >= 4
    kb = np.zeros(E, int)                                        # B: real steps taken per arena
    awaiting = np.zeros(E, bool)                                 # B: arenas whose next call is their reset
    n_resets = 0
    for tau in range(K):
        act = np.stack([acts[min(kb[e], K - 1), e] for e in range(E)])
        act[awaiting] = [0.37, -0.11]                            # ignored
        ob, outb = b.step(act)
        assert np.array_equal(b.reset_flags != 0, awaiting)
        for e in range(E):
            if awaiting[e]:
                ra = rec[kb[e] - 1]                              # the step of A that ended the episode
                assert outb["done"][e] == 0 and outb["reward"][e] == 0.0 and outb["is_crash"][e] == 0 and outb["is_success"][e] == 0
                assert np.array_equal(ob[e], ra[0][e]), (tau, e)
                assert np.array_equal(outb["achieved_goal"][e], ra[1]["achieved_goal"][e])
                assert np.array_equal(outb["desired_goal"][e], ra[1]["desired_goal"][e])
                n_resets += 1
                continue
            if kb[e] >= K:
                continue
            ra = rec[kb[e]]
            for n in ("reward", "done", "is_success", "is_crash", "distance"):
                assert outb[n][e] == ra[1][n][e], (n, tau, e)
            if ra[1]["done"][e]:
                assert np.array_equal(ob[e], ra[2][e]), ("terminal row", tau, e)
                assert np.array_equal(outb["achieved_goal"][e], ra[3][e, :2]) and np.array_equal(outb["desired_goal"][e], ra[3][e, 2:])
            else:
                assert np.array_equal(ob[e], ra[0][e]), (tau, e)
                assert np.array_equal(outb["achieved_goal"][e], ra[1]["achieved_goal"][e])
            kb[e] += 1
        awaiting = outb["done"] != 0
    assert n_resets >= E


def test_restart_of_some_arenas():
    """navsim_restart + navsim_reset_obs(mask): reset() of SOME arenas (the reference's reset() is per environment,
    env.py:730-831) -- next start / goal pair, next episode number, first observation; the others are not touched."""
    E, size = 6, 100
    cfg, host = _static_world(E, size, 9, abi.AUTORESET_NONE, S=3)
    r = ref.RefSim(cfg, host)
    r.reset_obs()
    rng = np.random.default_rng(0)
    for _ in range(4):
        obs, _ = r.step(np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.6, 0.6, E)], axis=1))
    before = {k: v.copy() for k, v in r.a.items()}
    rows = obs.copy()
    mask = np.array([0, 1, 0, 0, 1, 1], np.uint8)
    got = r.restart(mask)
    for e in range(E):
        if mask[e]:
            assert r.a["episode"][e] == before["episode"][e] + 1 and r.a["steps"][e] == 0
            table = before["spawn_pose"][e]
            assert any(np.array_equal(r.a["robot_pose"][e], p) for p in table)
            assert np.array_equal(r.a["prev_action"][e], [0.0, 0.0]) and r.a["n_hist"][e] == 1
            B = cfg.n_beams
            assert np.array_equal(got[e, :B], got[e, B:2 * B]) and np.array_equal(got[e, :B], got[e, 2 * B:3 * B])   # stack filled
        else:
            for k in ("robot_pose", "robot_goal", "episode", "steps", "prev_action", "prev_pose", "n_hist"):
                assert np.array_equal(r.a[k][e], before[k][e]), k
            assert np.array_equal(got[e], rows[e])
