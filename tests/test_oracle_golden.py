"""CPU: the oracle against golden vectors captured from the reference (tests/golden/make_golden.py).

Pinned rows of SURVEY.md section 8a: a7 (leg odometry), a8/a9 (set_vel), a11 (obs packing /
stacking), a12/a13 (reward, terminals, info), a15 (xy->ij), and the orchestration of a1 through
full reset()/step() traces of the reference's own code."""
import os

import os

import numpy as np
import pytest

import ref
from helpers import load_trace, trace_setup
from nav_gym_amd import abi


@pytest.fixture(scope="module")
def units(golden_dir):
    d = np.load(os.path.join(golden_dir, "golden_units.npz"))
    return {k: d[k] for k in d.files}


def test_human_set_vel(units):
    inp = units["set_vel_in"]
    for dt in np.unique(inp[:, 5]):
        m = inp[:, 5] == dt
        pose, vel = ref.integrate(inp[m, 0:3], inp[m, 3:5], float(dt), 0.0)
        exp = units["human_set_vel_out"][m]
        np.testing.assert_allclose(pose, exp[:, 0:3], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vel, exp[:, 3:5], rtol=0, atol=1e-12)


def test_keti_set_vel(units):
    inp = units["set_vel_in"]
    for dt in np.unique(inp[:, 5]):
        m = inp[:, 5] == dt
        pose, vel = ref.integrate(inp[m, 0:3], inp[m, 3:5], float(dt), 0.14474)
        exp = units["keti_set_vel_out"][m]
        np.testing.assert_allclose(pose, exp[:, 0:3], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vel, exp[:, 3:5], rtol=0, atol=1e-12)


def test_set_vel_survey_goldens():
    # SURVEY.md 8a rows a8 / a9 [PROBE]
    p, _ = ref.integrate([[1.0, 2.0, 0.3]], [[0.5, 0.64]], 0.2, 0.0)
    assert p[0].tolist() == [1.0909797672793022, 2.0415052038400487, 0.428]
    p, _ = ref.integrate([[1.0, 2.0, 0.3]], [[0.5, 0.64]], 0.2, 0.14474)
    assert p[0].tolist() == [1.0975710555552805, 2.0242041665141244, 0.428]


@pytest.mark.parametrize("size", [100, 400, 500, 1000])
def test_xy_to_ij(units, size):
    got = ref.xy_to_ij(units["xy_%d" % size], (0.0, 0.0), 0.05, size, size)
    assert np.array_equal(got, units["ij_%d" % size])


@pytest.mark.parametrize("size", [100, 400, 500, 1000])
def test_xy_to_ij_float32_inputs(units, size):
    """env.py:419 passes float32 coordinates: NumPy >= 2 then divides in float32."""
    got = ref.xy_to_ij_f32(units["xyf32_%d" % size], (0.0, 0.0), 0.05, size, size)
    assert np.array_equal(got, units["ijf32_%d" % size])


def test_xy_to_ij_survey_golden():
    assert ref.xy_to_ij([[3.14, 7.77]], (0, 0), 0.05, 500, 500).tolist() == [[62, 155]]


@pytest.mark.parametrize("S,B", [(1, 128), (3, 64)])
def test_reward_terminals_info(units, S, B):
    tag = "rd_S%d_B%d_" % (S, B)
    obs = np.concatenate([units[tag + "scans"].astype(np.float64), units[tag + "tail"]], axis=1)
    cfg = ref.default_config(n_beams=B, n_scan_stack=S)
    out = ref.reward_done(cfg, obs, units[tag + "goals"], units[tag + "thr"], units[tag + "dthr"])
    np.testing.assert_allclose(out["reward"], units[tag + "reward"], rtol=0, atol=1e-9)
    assert np.array_equal(out["done"].astype(bool), units[tag + "done"].astype(bool))
    assert np.array_equal(out["is_success"], units[tag + "is_success"])
    assert np.array_equal(out["is_crash"], units[tag + "is_crash"])
    np.testing.assert_allclose(out["distance"], units[tag + "distance"], rtol=0, atol=1e-12)
    # the batch must exercise every branch
    assert out["is_crash"].any() and out["is_success"].any() and (out["is_crash"] == 0).any()


def test_angle_correction(units):
    got = ref.math_fn(4, units["angle_in"])
    np.testing.assert_allclose(got, units["angle_out"], rtol=0, atol=1e-12)


def test_constants(units):
    cfg = ref.default_config()
    amin, amax, ainc, rmax, n = units["keti_lidar"]
    assert cfg.angle_min == amin and cfg.angle_last == amax - ainc
    assert cfg.range_max == rmax and cfg.n_beams == int(n)
    kw = dict(zip(units["kwargs_keys"].tolist(), units["kwargs_vals"].tolist()))
    assert float(kw["time_step"]) == cfg.time_step
    assert float(kw["distance_threshold"]) == cfg.distance_threshold
    assert int(kw["num_scan_stack"]) == cfg.n_scan_stack
    for k in ("scale", "success_factor", "crash_factor", "progress_factor", "forward_factor",
              "rotation_factor", "discomfort_factor"):
        assert float(kw["reward_" + k]) == getattr(cfg, "reward_" + k)
    assert float(kw["min_turning_radius"]) == cfg.min_turning_radius


def test_scan_thresholds_match_reference_built(units):
    """Thresholds the reference code derived (through the oracle's polygon renderer) at init."""
    tr = load_trace("random_S1")
    cfg = ref.default_config()
    thr = ref.scan_threshold(cfg, units["keti_threshold_footprint"])
    dthr = ref.scan_threshold(cfg, units["keti_discomfort_footprint"])
    assert np.array_equal(thr, tr["scan_threshold"])
    assert np.array_equal(dthr, tr["scan_discomfort"])
    # closed form (keti_robot.py:18-23): front 0.6, sides 0.6, rear 0.7
    assert abs(thr[256] - 0.6) < 1e-6 and abs(thr[128] - 0.6) < 1e-6 and abs(thr[384] - 0.6) < 1e-6
    assert abs(thr[0] - 0.7) < 1e-6


@pytest.mark.parametrize("name", ["random_S1", "peds_S1", "crash_S3", "success_S2", "corridor_S1"])
def test_step_trace(name):
    """The oracle's step() reproduces the reference's own step() on the recorded traces (corridor_S1: the reference's
    own 1000 x 1000 corridor map, round 4)."""
    tr = load_trace(name)
    cfg, arrays, _ = trace_setup(tr, ref.default_config, ref.build_dt)
    sim = ref.RefSim(cfg, arrays)
    S, B = int(tr["S"]), int(tr["B"])
    first = sim.reset_obs()
    assert np.array_equal(first[0, : S * B], tr["first_obs"][: S * B].astype(np.float32))
    np.testing.assert_allclose(first[0, S * B:], tr["first_obs"][S * B:], rtol=0, atol=2e-6)
    T = tr["actions"].shape[0]
    for t in range(T):
        sim.set_ped_cmd(tr["ped_cmd"][t][None])
        obs, out = sim.step(tr["actions"][t][None])
        # flags bit-exact, reward/pose tight
        assert out["done"][0] == tr["done"][t], t
        assert out["is_crash"][0] == tr["is_crash"][t], t
        assert out["is_success"][0] == tr["is_success"][t], t
        assert abs(out["reward"][0] - tr["reward"][t]) < 1e-9, t
        assert abs(out["distance"][0] - tr["distance"][t]) < 1e-12, t
        assert np.array_equal(obs[0, : S * B], tr["obs_scan"][t]), t
        np.testing.assert_allclose(obs[0, S * B:], tr["obs_tail"][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose(sim.a["robot_pose"][0], tr["traj_robot_pose"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sim.a["ped_pose"][0], tr["traj_ped_pose"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sim.a["ped_vel"][0], tr["traj_ped_vel"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sim.a["ped_dist"][0], tr["traj_ped_dist"][t], rtol=0, atol=1e-10)
        if "ped_scan_steps" in tr and t in tr["ped_scan_steps"]:
            # env.py:685-693: the pedestrians' own scans (robot + other pedestrians as polygons)
            idx = int(np.where(tr["ped_scan_steps"] == t)[0][0])
            got = sim.ped_scans()[0, : tr["ped_scan"].shape[1]]
            assert np.array_equal(got, tr["ped_scan"][idx]), ("pedestrian scans", t)
            assert (got < 6.0).any()
    if name.startswith("crash"):
        assert tr["is_crash"].sum() > 0 and (tr["is_crash"] == 0).sum() > 0
    if name.startswith("success"):
        assert tr["is_success"].sum() > 0


def test_path_to_waypoints(units):
    """env.py:1261-1277 against the reference's outputs on 20 synthetic paths."""
    for k in range(units["wp_paths"].shape[0]):
        p = units["wp_paths"][k]; p = p[np.isfinite(p[:, 0])]
        exp = units["wp_out"][k]; exp = exp[np.isfinite(exp[:, 0])]
        got = ref.path_to_waypoints(p, float(units["wp_interval"][k]))
        assert got.shape == exp.shape and np.array_equal(got, exp), k
    # SURVEY.md 8a row a16 [PROBE]: straight path at 0.25 m spacing, interval 2 -> first waypoint x = 2.25
    straight = np.stack([np.arange(0, 10, 0.25), np.zeros(40)], 1)
    assert ref.path_to_waypoints(straight, 2)[0, 0] == 2.25


def _long_routes(golden_dir):
    d = np.load(os.path.join(golden_dir, "golden_long_routes.npz"))
    shape = tuple(int(x) for x in d["cost_shape"])
    cost = np.unpackbits(d["cost_packed"])[: shape[0] * shape[1]].reshape(shape)
    return d, cost


def test_long_routes_vs_reference(golden_dir):
    """Routes of FULL length (round 4; env.py:788-804, 1261-1277: the reference keeps every waypoint of a route).
    (1) path_to_waypoints on 16 paths of 33-64 waypoints equals the reference's own output, waypoint for waypoint;
    (2) 24 routes the reference's own _sample_start_goal_path drew on the costmap of its 1000 x 1000 corridor episode
        (half of them longer than the 16 waypoints rounds 1-3 kept): the reference's waypoints of the recorded path are
        reproduced exactly; the oracle's own planner joins the same start and goal with a path of the SAME number of
        cells (a shortest path: which one is the build-defined tie-break) and keeps all of its waypoints."""
    d, cost = _long_routes(golden_dir)
    assert d["wp_n"].min() >= 30 and d["wp_n"].max() >= 60
    for k in range(d["wp_paths"].shape[0]):
        p = d["wp_paths"][k][: int(d["wp_n_points"][k])]
        exp = d["wp_out"][k][: int(d["wp_n"][k])]
        got = ref.path_to_waypoints(p, 2.0, max_wp=128)
        assert got.shape == exp.shape and np.array_equal(got, exp), k
    cfg = ref.default_config()
    assert cfg.max_waypoints == 64
    res_c = float(d["cost_resolution"])
    n_long = 0
    for k in range(d["route_start"].shape[0]):
        n_exp = int(d["route_n_wp"][k])
        path = d["route_path"][k][: int(d["route_cells"][k])]
        got = ref.path_to_waypoints(path, 2.0, max_wp=128)
        assert np.array_equal(got, d["route_wp"][k][:n_exp]), k
        wp, n_wp, cells, plen = ref.plan(cost[None], d["route_start"][k][None], d["route_goal"][k][None], 2.0,
                                         max_wp=cfg.max_waypoints, res_c=res_c)
        assert cells[0] == d["route_cells"][k], (k, cells[0], d["route_cells"][k])
        assert abs(int(n_wp[0]) - n_exp) <= 2 and n_wp[0] < cfg.max_waypoints       # nothing cut at the default capacity
        assert np.array_equal(wp[0, n_wp[0] - 1], d["route_goal"][k])               # the list ends on the goal itself
        n_long += n_exp > 16
    assert n_long >= 12


def test_cut_route_length_counts_every_waypoint(golden_dir):
    """A route longer than max_wp is stored cut, but its path_distance (env.py:757-759) still runs over every waypoint
    of the path: the robot's `path_distance > 2 |goal - start|` rule must not depend on the capacity."""
    d, cost = _long_routes(golden_dir)
    res_c = float(d["cost_resolution"])
    k = int(np.argmax(d["route_n_wp"]))
    s, g = d["route_start"][k][None], d["route_goal"][k][None]
    wp_full, n_full, _, len_full = ref.plan(cost[None], s, g, 2.0, max_wp=128, res_c=res_c)
    wp_cut, n_cut, _, len_cut = ref.plan(cost[None], s, g, 2.0, max_wp=8, res_c=res_c)
    assert n_full[0] > 16 and n_cut[0] == 8
    assert np.array_equal(wp_cut[0], wp_full[0, :8]) and len_cut[0] == len_full[0]
    seg = np.diff(np.vstack([s, wp_full[0, : n_full[0]]]), axis=0)
    assert abs(np.linalg.norm(seg, axis=1).sum() - len_full[0]) < 1e-9


@pytest.mark.parametrize("name", ["random_S1", "peds_S1", "crash_S3", "success_S2", "corridor_S1"])
def test_policy_control_block_vs_reference_trace(name):
    """Row a10 (env.py:617-662 + human_policy.py:19-52): with the weights the traces were recorded with,
    the oracle's control block -- pedestrian scan -> clip / scale -> actor network -> clip -> * v_pref,
    local goal from the waypoint list -- reproduces the (v, omega) the reference handed to
    Human.set_vel and its prev_human_actions at every step, within 1e-5 (float32 network, different
    summation order than torch's kernels)."""
    from helpers import policy_weights
    tr = load_trace(name)
    cfg, arrays, occ = trace_setup(tr, ref.default_config, ref.build_dt)
    N = tr["init_ped_pose"].shape[0]
    nw = tr["init_ped_n_waypoints"]
    assert nw.max() <= cfg.max_waypoints
    wp = np.zeros((1, N, cfg.max_waypoints, 2))
    wp[0, :, :tr["init_ped_waypoints"].shape[1]] = tr["init_ped_waypoints"]
    arrays["ped_waypoints"] = wp
    arrays["ped_n_waypoints"] = nw[None].astype(np.int32)
    w = policy_weights(int(tr["policy_seed"]))
    r = ref.RefSim(cfg, arrays)
    r.reset_obs()
    T = tr["actions"].shape[0]
    checked = 0
    worst = 0.0
    replanned = np.zeros(N, bool)     # the reference drew this pedestrian a new random path (env.py:667-680):
    for t in range(T):                # its waypoints are no longer the recorded initial ones
        if t > 0:                                        # the speed input is the reference's own previous output
            r.prev_actions[0] = tr["ped_mean"][t - 1]
        cmd, mean = r.ped_policy(w)
        if t == 0 or not tr["is_crash"][t - 1]:          # after a crash the reference's pedestrians saw the
            ok = ~replanned                              # robot at its pre-revert pose (env.py:685-723)
            worst = max(worst, np.abs(cmd[0][ok] - tr["ped_cmd"][t][ok]).max(),
                        np.abs(mean[0][ok] - tr["ped_mean"][t][ok]).max())
            checked += int(ok.sum())
        r.set_ped_cmd(tr["ped_cmd"][t][None])
        r.step(tr["actions"][t][None])
        n_now = r.a["ped_n_waypoints"][0]
        last = r.a["ped_waypoints"][0, np.arange(N), n_now - 1]
        replanned |= np.linalg.norm(r.a["ped_pose"][0, :, :2] - last, axis=1) < 0.5
    assert checked >= 30 and worst < 1e-5, (checked, worst)
    assert replanned.sum() <= 1
    if name == "corridor_S1":           # the route of more than 32 m is walked from its full waypoint list
        assert nw.max() > 16 and (r.a["ped_wp_head"][0] > 0).any()       # (ABI 5: a pop advances the head, the list stays)


def _crowd_golden():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_crowd.npz"))
    shape = tuple(int(x) for x in d["map_shape"])
    maps = np.unpackbits(d["maps"])[: int(np.prod(shape))].reshape(shape)
    params = {str(k): float(v) for k, v in zip(d["param_names"], d["params"])}
    return d, maps, params


def test_crowd_check_vs_reference_step():
    """the collision / goal / reward block of the reference's own CrowdSim.step
    (crowd_sim.py:808-949) on 600 random situations: info class and done flag exact, reward and Danger
    distance to 1e-12."""
    d, maps, params = _crowd_golden()
    reward, done, info, md = ref.crowd_check(params, maps, d["robot"], d["agents"], d["global_time"])
    assert np.array_equal(info, d["info"]) and np.array_equal(done, d["done"])
    assert np.abs(reward - d["reward"]).max() < 1e-12
    danger = d["info"] == 5
    assert np.abs(md[danger] - d["dmin"][danger]).max() < 1e-12
    assert set(np.unique(info)) >= {0, 1, 2, 3, 5}


def _crowd_maps(golden_dir):
    d = np.load(os.path.join(golden_dir, "golden_crowd_maps.npz"))
    P = {str(k): float(v) for k, v in zip(d["param_names"], d["params"])}
    maps = np.unpackbits(d["maps"])[: int(np.prod(d["map_shape"]))].reshape(d["map_shape"])
    lmap = np.unpackbits(d["lmap"])[: int(np.prod(d["lmap_shape"]))].reshape(d["lmap_shape"])
    return d, P, maps, lmap


def test_crowd_angular_map_vs_reference(golden_dir):
    """CrowdSim.get_local_map_angular + calculate_angular_map_distances (crowd_sim.py:999-1102) on 400 situations
    recorded from the reference's own functions: same sectors touched, distances to 1e-12 (cos / sin / atan2 are the
    deterministic functions, within an ulp of the libm values the reference used)."""
    d, P, _, _ = _crowd_maps(golden_dir)
    got = ref.crowd_angular_map(P, d["robot"], d["verts"], d["n_obst"])
    assert np.array_equal(got < 1.0, d["amap"] < 1.0)
    np.testing.assert_allclose(got, d["amap"], rtol=0, atol=1e-12)
    assert (d["amap"] < 1.0).mean() > 0.2
    # ragged obstacle counts incl. none: all sectors at max range
    none = ref.crowd_angular_map(P, d["robot"][:3], d["verts"][:3], np.zeros(3, np.int32))
    assert (none == 1.0).all()
    raw = ref.crowd_angular_map(dict(P, normalize=0), d["robot"][:8], d["verts"][:8], d["n_obst"][:8])
    np.testing.assert_allclose(raw / P["angular_max_range"], got[:8], rtol=0, atol=1e-15)


def test_crowd_local_map_window_vs_reference(golden_dir):
    """CrowdSim.get_local_map (crowd_sim.py:1104-1166) window logic -- centre cell with Python's round-half-even,
    clipping at the map border, the exclusive slice ends, the 0.9 threshold -- on 200 situations recorded from the
    reference with the rotation replaced by the identity: exact."""
    d, P, maps, lmap = _crowd_maps(golden_dir)
    got = ref.crowd_local_map(P, maps, d["robot2"], rotate=False)
    assert np.array_equal(got, lmap)
    assert (lmap[:, -1, :] == 1).all() and (lmap[:, :, -1] == 1).all()       # the reference's exclusive slice ends
    assert (lmap == 0).any()


def test_crowd_local_map_rotation_properties(golden_dir):
    """rotate_grid_around_center (crowd_sim.py:1168-1186) is restated from OpenCV's warpAffine (cv2 is absent:
    UNPINNED).  What can be checked without it: heading pi/2 is a rotation by 0 degrees = the unrotated window;
    headings 0 and pi are quarter / half turns about (S/2, S/2) = exact index permutations with one border line."""
    d, P, maps, _ = _crowd_maps(golden_dir)
    rob = d["robot2"][:40].copy()
    plain = ref.crowd_local_map(P, maps[:40], rob, rotate=False)
    rob[:, 2] = np.pi / 2
    assert np.array_equal(ref.crowd_local_map(P, maps[:40], rob, rotate=True), plain)
    S = plain.shape[1]
    rob[:, 2] = -np.pi / 2                                   # angle = 180 degrees: dst[y, x] = src[S - y, S - x]
    half = ref.crowd_local_map(P, maps[:40], rob, rotate=True)
    exp = np.ones_like(plain)
    exp[:, 1:, 1:] = plain[:, :0:-1, :0:-1]
    assert np.array_equal(half, exp)


ORCA_P = dict(time_step=0.25, neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10)   # orca.py:62-65


def test_crowd_agent_step_vs_reference(golden_dir):
    """Agent.step with an ActionRot (crowd_sim/envs/utils/agent.py:108-141) on 500 states recorded from the
    reference's own method: pose and velocity to 1e-12 (deterministic cos / sin within an ulp of libm)."""
    d = np.load(os.path.join(golden_dir, "golden_crowd_agent.npz"))
    pose, vel = ref.crowd_agent_step(d["pose"], d["action"], float(d["time_step"]))
    np.testing.assert_allclose(pose, d["pose_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(vel, d["vel_out"], rtol=0, atol=1e-12)


def _orca_rollout(pos, goal, radius, v_pref, verts=None, steps=200):
    """CrowdSim's ORCA pedestrians in closed loop: every agent builds its own query (itself first), then all move."""
    n = len(pos)
    pos = np.array(pos, float); vel = np.zeros((n, 2)); goal = np.array(goal, float)
    closest, traj = np.inf, [pos.copy()]
    for _ in range(steps):
        ag = np.zeros((n, n, 6)); pv = np.zeros((n, 2))
        for i in range(n):
            for k, j in enumerate([i] + [j for j in range(n) if j != i]):
                ag[i, k] = [pos[j, 0], pos[j, 1], vel[j, 0], vel[j, 1], radius, v_pref]
            dvec = goal[i] - pos[i]; sp = np.linalg.norm(dvec)
            pv[i] = dvec / sp if sp > 1 else dvec                              # orca.py:116-120
        vel, _ = ref.crowd_orca(ORCA_P, ag, pv, verts)
        pos = pos + vel * ORCA_P["time_step"]
        traj.append(pos.copy())
        for i in range(n):
            for j in range(i + 1, n):
                closest = min(closest, float(np.linalg.norm(pos[i] - pos[j])))
    return pos, closest, np.array(traj)


def test_crowd_orca_properties():
    """rvo2 is absent (UNPINNED): what the restated RVO2 step must do regardless -- alone an agent takes its
    preferred velocity (clipped to its maximum speed); two agents walking at each other pass without their discs
    overlapping and reach their goals; eight agents crossing a circle never overlap; a wall in the way is never
    entered; the mirrored problem gives the mirrored answer."""
    v, a = ref.crowd_orca(ORCA_P, [[[0, 0, 0, 0, 0.3, 1.0]]], [[0.6, 0.0]], theta=[0.5])
    assert np.allclose(v, [[0.6, 0.0]], atol=1e-7) and np.allclose(a, [[0.6, -0.5]], atol=1e-7)
    v, _ = ref.crowd_orca(ORCA_P, [[[0, 0, 0, 0, 0.3, 1.0]]], [[3.0, 4.0]])
    assert abs(np.hypot(*v[0]) - 1.0) < 1e-6
    pos, closest, _ = _orca_rollout([[-3, 0], [3, 0.001]], [[3, 0], [-3, 0]], 0.3, 1.0)
    assert closest >= 0.6 - 1e-4 and np.abs(pos - [[3, 0], [-3, 0]]).max() < 0.05
    ang = np.arange(8) * 2 * np.pi / 8
    p0 = np.stack([4 * np.cos(ang), 4 * np.sin(ang)], 1)
    _, closest, _ = _orca_rollout(p0, -p0, 0.3, 1.0, steps=120)
    assert closest >= 0.6 - 1e-4
    box = np.array([[[[1, 1], [-1, 1], [-1, -1], [1, -1]]]], float)             # counter-clockwise
    _, _, traj = _orca_rollout([[-4, 0.2]], [[4, 0]], 0.3, 1.0, verts=box, steps=200)
    assert not ((np.abs(traj[:, 0, 0]) < 1.29) & (np.abs(traj[:, 0, 1]) < 1.29)).any()
    rng = np.random.default_rng(1)                                               # mirror symmetry (y -> -y)
    ag = np.zeros((50, 6, 6)); ag[..., :2] = rng.uniform(-3, 3, (50, 6, 2)); ag[..., 2:4] = rng.uniform(-1, 1, (50, 6, 2))
    ag[..., 4] = 0.3; ag[..., 5] = 1.0
    pv = rng.uniform(-1, 1, (50, 2))
    v1, _ = ref.crowd_orca(ORCA_P, ag, pv)
    m = ag.copy(); m[..., 1] *= -1; m[..., 3] *= -1
    v2, _ = ref.crowd_orca(ORCA_P, m, pv * [1, -1])
    assert np.allclose(v1 * [1, -1], v2, atol=1e-5)
