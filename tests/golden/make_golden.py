#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the importable parts of the reference (run in the build
container only: /root/reference does not exist on the GPU box and is never read by tests).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is captured:
  golden_units.npz  -- inputs/outputs of the reference's own pure-NumPy functions, imported
                       unmodified: Human.set_vel, KetiRobot.set_vel, batch_xy_to_ij,
                       compute_rewards / compute_terminals / compute_info, _stack_scan,
                       path_to_waypoints, angle_correction, transform_xys, registered kwargs.
  golden_trace_*.npz -- input/output traces of the reference's own NavGymEnv.reset()/step()
                       orchestration (env.py:591-831), executed unmodified.  The six pip packages it
                       imports are absent here, so import-only stand-ins are injected into
                       sys.modules: `range_libc`, `CMap2D` route the three L1 calls
                       (calc_range_many, render_contours_in_lidar, render_agents_in_lidar) to the
                       CPU oracle's primitives; `pose2d` is a NumPy restatement; `pyastar2d` is a
                       BFS; `cv2` is a NumPy/SciPy subset; `gym` is a 20-line stub; torch.load of
                       the missing human_policy.pth returns seeded random weights.
                       => the traces pin ORCHESTRATION (order of operations, obs packing, leg
                       odometry, reward/done/info, crash revert, scan stacking), NOT the L1
                       arithmetic, which stays "parity unpinned" (oracle/navsim_ref.h).
Only data (inputs and expected outputs) is written; no reference source text is stored.
"""
import os
import sys
import types
from collections import deque

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np  # noqa: E402
import scipy.ndimage as ndi  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/nav_gym/src"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref as oracle  # noqa: E402


# --------------------------------------------------------------------------------------------
# import-only stand-ins for the missing pip packages
# --------------------------------------------------------------------------------------------
def install_shims():
    # ---- gym ------------------------------------------------------------------------------
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class EzPickle(object):
        def __init__(self, *a, **k):
            pass

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Dict(object):
        def __init__(self, spaces):
            self.spaces = spaces
    gym.Env = Env
    gym.utils = types.ModuleType("gym.utils")
    gym.utils.EzPickle = EzPickle
    gym.spaces = types.ModuleType("gym.spaces")
    gym.spaces.Box = Box
    gym.spaces.Dict = Dict
    gym.envs = types.ModuleType("gym.envs")
    gym.envs.registration = types.ModuleType("gym.envs.registration")
    gym.registry = {}

    def register(id, kwargs=None, entry_point=None, **_):
        gym.registry[id] = dict(kwargs=kwargs, entry_point=entry_point)
    gym.envs.registration.register = register
    for name in ("gym", "gym.utils", "gym.spaces", "gym.envs", "gym.envs.registration"):
        sys.modules[name] = {"gym": gym, "gym.utils": gym.utils, "gym.spaces": gym.spaces,
                             "gym.envs": gym.envs, "gym.envs.registration": gym.envs.registration}[name]

    # ---- cv2 (reset path only) ---------------------------------------------------------------
    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "4.6.0"
    cv2.INTER_NEAREST = 0
    cv2.THRESH_BINARY = 0
    cv2.RETR_TREE = 3
    cv2.CHAIN_APPROX_SIMPLE = 2

    def resize(img, dsize, interpolation=0):
        w, h = dsize
        ys = np.minimum((np.arange(h) * (img.shape[0] / float(h))).astype(int), img.shape[0] - 1)
        xs = np.minimum((np.arange(w) * (img.shape[1] / float(w))).astype(int), img.shape[1] - 1)
        return img[np.ix_(ys, xs)]

    def filter2D(img, ddepth, kernel):
        out = ndi.correlate(img.astype(np.float64), kernel.astype(np.float64), mode="mirror")
        return np.clip(out, 0, 255).astype(img.dtype)

    def threshold(img, thresh, maxval, typ):
        return thresh, (img > thresh).astype(img.dtype) * maxval

    def dilate(img, kernel, iterations=1):
        return ndi.grey_dilation(img, footprint=kernel.astype(bool))

    def findContours(img, mode, method):
        return [], None          # static contours are only used by thresholds, which overwrite them
    cv2.resize, cv2.filter2D, cv2.threshold, cv2.dilate, cv2.findContours = \
        resize, filter2D, threshold, dilate, findContours
    sys.modules["cv2"] = cv2

    # ---- range_libc -> oracle a3/a4 ------------------------------------------------------------
    rl = types.ModuleType("range_libc")

    class PyOMap(object):
        def __init__(self, arr):
            self.occ = np.ascontiguousarray(arr).astype(np.uint8)

    class PyRayMarching(object):
        def __init__(self, omap, max_range):
            self.field = oracle.build_dt(omap.occ)
            self.max_range = float(max_range)

        def calc_range_many(self, ins, outs):
            assert ins.dtype == np.float32 and outs.dtype == np.float32
            outs[:] = oracle.cast_static(self.field, ins[None], self.max_range)[0]
    rl.PyOMap, rl.PyRayMarching = PyOMap, PyRayMarching
    sys.modules["range_libc"] = rl

    # ---- CMap2D -> oracle a5/a6 ------------------------------------------------------------------
    cm = types.ModuleType("CMap2D")

    def flatten_contours(contours):
        n = int(np.sum([len(c) for c in contours]))
        flat = np.zeros((n, 3), dtype=np.float32)
        v = 0
        for idx, contour in enumerate(contours):
            for vertex in contour:
                flat[v, :] = np.array([idx, vertex[0], vertex[1]])
                v += 1
        return flat

    def render_contours_in_lidar(ranges, angles, flat_contours, lidar_ij):
        ranges[:] = oracle.render_polys(ranges[None], np.asarray(angles, np.float64)[None],
                                        flat_contours[None], [len(flat_contours)],
                                        np.asarray(lidar_ij, np.float32)[None])[0]

    class CSimAgent(object):
        def __init__(self, pos, dist, vel):
            self.pos, self.dist, self.vel = pos, dist, vel

    class CMap2D_(object):
        def set_resolution(self, r):
            self.res = r

        def render_agents_in_lidar(self, ranges, angles, agents, lidar_ij):
            if not agents:
                return
            a = np.zeros((1, len(agents), 8), np.float32)
            for i, ag in enumerate(agents):
                a[0, i, 0:3] = ag.pos
                a[0, i, 3:6] = ag.dist
                a[0, i, 6:8] = ag.vel
            ranges[:] = oracle.render_legs(ranges[None], np.asarray(angles, np.float64)[None], a,
                                           [len(agents)], np.asarray(lidar_ij, np.float32)[None])[0]
    cm.flatten_contours, cm.render_contours_in_lidar = flatten_contours, render_contours_in_lidar
    cm.CMap2D, cm.CSimAgent = CMap2D_, CSimAgent
    sys.modules["CMap2D"] = cm

    # ---- pose2d (NumPy restatement of the published helpers) ------------------------------------
    p2 = types.ModuleType("pose2d")

    def rotate(x, th):
        rotmat = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        return np.matmul(rotmat, np.asarray(x).T).T

    def inverse_pose2d(p):
        inv_th = -p[2]
        inv_xy = rotate(np.array([-p[:2]]), inv_th)[0]
        return np.array([inv_xy[0], inv_xy[1], inv_th])

    def apply_tf_to_vel(vel, pose2d):
        xy = rotate(vel[:2], pose2d[2])
        return np.array([xy[0], xy[1], vel[2]])
    p2.inverse_pose2d, p2.apply_tf_to_vel = inverse_pose2d, apply_tf_to_vel
    sys.modules["pose2d"] = p2

    # ---- pyastar2d: 4-connected BFS ----------------------------------------------------------------
    pa = types.ModuleType("pyastar2d")

    def astar_path(grid, start, goal, allow_diagonal=False):
        free = np.isfinite(grid)
        s, g = (int(start[0]), int(start[1])), (int(goal[0]), int(goal[1]))
        if not (free[s] and free[g]):
            return None
        prev = -np.ones(grid.shape + (2,), dtype=np.int32)
        seen = np.zeros(grid.shape, bool)
        seen[s] = True
        dq = deque([s])
        while dq:
            c = dq.popleft()
            if c == g:
                break
            for d in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                nb = (c[0] + d[0], c[1] + d[1])
                if 0 <= nb[0] < grid.shape[0] and 0 <= nb[1] < grid.shape[1] and free[nb] and not seen[nb]:
                    seen[nb] = True
                    prev[nb] = c
                    dq.append(nb)
        if not seen[g]:
            return None
        path = [g]
        while path[-1] != s:
            path.append(tuple(prev[path[-1]]))
        return np.array(path[::-1])
    pa.astar_path = astar_path
    sys.modules["pyastar2d"] = pa


def import_reference():
    install_shims()
    sys.path.insert(0, REF_SRC)
    import nav_gym_env  # noqa: F401  (registers NavGym-v0 into the gym stub)
    from nav_gym_env import env as ref_env
    from nav_gym_env import human, keti_robot, utils as ref_utils, human_policy
    return ref_env, human, keti_robot, ref_utils, human_policy


# --------------------------------------------------------------------------------------------
# unit goldens
# --------------------------------------------------------------------------------------------
def make_units(ref_env, human, keti_robot, ref_utils):
    rng = np.random.default_rng(20261002)
    out = {}
    # (1) set_vel: 1000 random (px, py, theta, v, w, dt)
    n = 1000
    inp = np.stack([rng.uniform(0, 50, n), rng.uniform(0, 50, n), rng.uniform(0, 2 * np.pi, n),
                    rng.uniform(-0.2, 1.0, n), rng.uniform(-1.5, 1.5, n),
                    rng.choice([0.1, 0.2, 0.25], n)], axis=1)
    inp[:50, 2] = rng.uniform(-np.pi, np.pi, 50)         # wrapped yaws after a crash revert
    h_out = np.zeros((n, 5)); k_out = np.zeros((n, 5))
    for i in range(n):
        px, py, th, v, w, dt = inp[i]
        h = human.Human(px, py, th, 0., 0., dt); h.set_vel(v, w)
        h_out[i] = [h.px, h.py, h.theta, h.vx, h.vy]
        k = keti_robot.KetiRobot(px, py, th, 0., 0., dt); k.set_vel(v, w)
        k_out[i] = [k.px, k.py, k.theta, k.vx, k.vy]
    out.update(set_vel_in=inp, human_set_vel_out=h_out, keti_set_vel_out=k_out)
    # robot constants
    K = keti_robot.KetiRobot
    out.update(keti_footprint=np.array(K.footprint), keti_threshold_footprint=np.array(K.threshold_footprint),
               keti_discomfort_footprint=np.array(K.discomfort_threshold_footprint),
               keti_lidar=np.array([K.angle_min, K.angle_max, K.angle_increment, K.range_max, K.n_angles]),
               human_footprint=np.array(human.Human.footprint),
               human_lidar=np.array([human.Human.angle_min, human.Human.angle_max,
                                     human.Human.angle_increment, human.Human.range_max, human.Human.n_angles]))
    # (2) xy -> ij: float32-representable values carried in float64
    for size in (100, 400, 500, 1000):
        xy = rng.uniform(-1.0, size * 0.05 + 1.0, (1000, 2)).astype(np.float32).astype(np.float64)
        mi = dict(resolution=0.05, origin=(0, 0), height=size, width=size)
        out["xy_%d" % size] = xy
        out["ij_%d" % size] = ref_env.batch_xy_to_ij(xy, mi)
        # float32 inputs, as the scan origin passes them (env.py:419); grid-aligned values included
        xyf = xy.astype(np.float32)
        xyf[:200] = (rng.integers(0, size, (200, 2)) * 0.05 + rng.choice([0.0, 0.025, 0.05], (200, 2))).astype(np.float32)
        out["xyf32_%d" % size] = xyf
        out["ijf32_%d" % size] = ref_env.batch_xy_to_ij(xyf, mi)
    out["ij_to_xy_in"] = rng.integers(0, 400, (64, 2))
    out["ij_to_xy_out"] = ref_env.batch_ij_to_xy(out["ij_to_xy_in"], dict(resolution=0.05, origin=(0, 0)))
    # (3) reward / terminals / info on random obs batches with synthetic thresholds
    for S, B in ((1, 128), (3, 64)):
        env = object.__new__(ref_env.NavGymEnv)
        kw = sys.modules["gym"].registry["NavGym-v0"]["kwargs"]
        for k, v in kw.items():
            setattr(env, k, v)
        env.num_scan_stack = S
        env.robot = types.SimpleNamespace(n_angles=B)
        thr = rng.uniform(0.5, 0.9, B).astype(np.float32)
        dthr = (thr + rng.uniform(0.3, 0.6, B)).astype(np.float32)
        env.scan_threshold, env.scan_discomfort_threshold = thr, dthr
        nb = 96
        scans = rng.uniform(0.3, 6.0, (nb, S * B)).astype(np.float32)
        scans[: nb // 2] += 1.0                              # half the rows comfortably clear
        scans[: nb // 4] += 2.0
        tail = np.concatenate([rng.uniform(0, 20, (nb, 4)), rng.uniform(-0.7, 0.7, (nb, 2)),
                               rng.uniform(-np.pi, np.pi, (nb, 1))], axis=1)
        obs = np.concatenate([scans.astype(np.float64), tail], axis=1)
        goals = tail[:, 2:4] + rng.uniform(-3, 3, (nb, 2))
        goals[:16] = tail[:16, 2:4] + rng.uniform(-0.3, 0.3, (16, 2))   # successes
        # obs = concat(scans (float32 values) as float64, tail): stored split to keep the file small
        actions = rng.uniform(-1, 1, (nb, 2))
        od = dict(observation=obs, desired_goal=goals)
        tag = "rd_S%d_B%d_" % (S, B)
        out[tag + "scans"] = scans; out[tag + "tail"] = tail; out[tag + "goals"] = goals; out[tag + "thr"] = thr; out[tag + "dthr"] = dthr
        out[tag + "reward"] = env.compute_rewards(actions, od)
        out[tag + "done"] = env.compute_terminals(od)
        infos = [env.compute_info(dict(observation=obs[i], desired_goal=goals[i])) for i in range(nb)]
        out[tag + "is_success"] = np.array([i["is_success"] for i in infos], np.float32)
        out[tag + "is_crash"] = np.array([i["is_crash"] for i in infos], np.float32)
        out[tag + "distance"] = np.array([i["distance"] for i in infos], np.float64)
        # _stack_scan with queues of every fill level
        if S == 3:
            stack_in, stack_q, stack_out = [], [], []
            for fill in range(S):
                q = deque(maxlen=S - 1)
                prevs = []
                for _ in range(fill):
                    po = rng.uniform(0, 6, S * B + 7)
                    q.append(dict(observation=po)); prevs.append(po)
                cur = rng.uniform(0, 6, B + 7)
                res = env._stack_scan(dict(observation=cur), q, S, B)
                stack_in.append(cur)
                stack_q.append(np.stack(prevs + [np.zeros(S * B + 7)] * (S - 1 - fill)))
                stack_out.append(res["observation"])
            out["stack_cur"] = np.stack(stack_in); out["stack_queue"] = np.stack(stack_q)
            out["stack_out"] = np.stack(stack_out)
    # (4) path_to_waypoints on synthetic paths
    wp_paths, wp_out, wp_int = [], [], []
    for t in range(20):
        m = int(rng.integers(5, 120))
        steps = rng.choice([[0.25, 0], [0, 0.25], [-0.25, 0], [0, -0.25]], m, p=[0.4, 0.3, 0.15, 0.15])
        path = np.cumsum(np.vstack([[rng.uniform(0, 20, 2)], steps]), axis=0)
        interval = [2, 5][t % 2]
        w = ref_env.path_to_waypoints(path, interval=interval)
        pad = np.full((128, 2), np.nan); pad[: len(path)] = path
        wpad = np.full((64, 2), np.nan); wpad[: len(w)] = w
        wp_paths.append(pad); wp_out.append(wpad); wp_int.append(interval)
    out.update(wp_paths=np.stack(wp_paths), wp_out=np.stack(wp_out), wp_interval=np.array(wp_int))
    # (5) utils
    ang = rng.uniform(-20, 20, 256)
    out.update(angle_in=ang, angle_out=ref_utils.angle_correction(ang))
    xys = rng.uniform(-1, 1, (16, 2))
    T = ref_utils.translation_matrix_from_xyz(3.25, -1.5, 0)
    R = ref_utils.quaternion_matrix_from_yaw(0.7)
    out.update(tf_xys_in=xys, tf_xys_out=ref_utils.transform_xys(T, R, xys))
    # (6) registered kwargs
    kw = sys.modules["gym"].registry["NavGym-v0"]["kwargs"]
    flat = {k: v for k, v in kw.items() if k != "env_param_range"}
    out["kwargs_keys"] = np.array(sorted(flat.keys()))
    out["kwargs_vals"] = np.array([str(flat[k]) for k in sorted(flat.keys())])
    epr = kw["env_param_range"]
    out["env_param_keys"] = np.array(sorted(epr.keys()))
    out["env_param_vals"] = np.array([str(epr[k]) for k in sorted(epr.keys())])
    out["entry_point"] = np.array(sys.modules["gym"].registry["NavGym-v0"]["entry_point"])
    np.savez_compressed(os.path.join(HERE, "golden_units.npz"), **out)
    print("golden_units.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------
# step() traces
# --------------------------------------------------------------------------------------------
def make_env(ref_env, human_policy, S, seed, indoor=False):
    kw = dict(sys.modules["gym"].registry["NavGym-v0"]["kwargs"])
    kw["num_scan_stack"] = S
    kw["indoor_ratio"] = 1.0 if indoor else 0.0   # outdoor 400x400 maps keep the fixture small; indoor = the reference's
                                                  # 1000 x 1000 corridor map (create_indoor_map), for the long routes
    epr = dict(kw["env_param_range"])
    epr["scan_noise_std"] = ([0., 0.], "float")   # parity is defined without noise
    epr["num_humans"] = ([5, 7], "int")
    kw["env_param_range"] = epr
    torch.manual_seed(seed)
    weights = human_policy.HumanPolicy(frames=3, action_space=2).state_dict()
    real_load = torch.load
    torch.load = lambda *a, **k: weights          # human_policy.pth is a missing blob
    np.random.seed(seed)
    try:
        env = ref_env.NavGymEnv(**kw)
        env.reset()
    finally:
        torch.load = real_load
    return env


def rebuild_first_obs(env):
    """The tail of the reference's reset() (env.py:808-831) after poses were edited."""
    env.prev_action = np.array([0., 0.])
    env.prev_obs = None
    env.prev_obs_queue = deque(maxlen=env.num_scan_stack - 1)
    env.distances_travelled_in_base_frame = np.zeros((len(env.humans), 3))
    env.prev_humans_obs_queue = [deque(maxlen=2) for _ in env.humans]
    for i, h in enumerate(env.humans):
        others = [env.robot] + [x for x in env.humans if x != h]
        ho = env._convert_obs(h, others, env.prev_obs, env.prev_action, add_scan_noise=False, lidar_legs=False)
        ho = env._stack_scan(ho, env.prev_humans_obs_queue[i], 3, h.n_angles)
        env.prev_humans_obs_queue[i].append(ho)
    obs = env._convert_obs(env.robot, env.humans, env.prev_obs, env.prev_action, add_scan_noise=True, lidar_legs=True)
    obs = env._stack_scan(obs, env.prev_obs_queue, env.num_scan_stack, env.robot.n_angles)
    env.prev_obs = obs
    env.prev_obs_queue.append(obs)
    return obs


def snapshot(env):
    hs = env.humans
    return dict(
        robot_pose=np.array([env.robot.px, env.robot.py, env.robot.theta]),
        ped_pose=np.array([[h.px, h.py, h.theta] for h in hs]).reshape(-1, 3),
        ped_vel=np.array([[h.vx, h.vy] for h in hs]).reshape(-1, 2),
        ped_dist=np.array(env.distances_travelled_in_base_frame).reshape(-1, 3),
    )


def run_trace(name, ref_env, human, human_policy, S, seed, scenario, n_steps, ped_scan_every=0):
    env = make_env(ref_env, human_policy, S, seed, indoor=(scenario == "corridor"))
    if scenario == "corridor":
        # the reference's own reset() on its own 1000 x 1000 corridor map: keep the first seed whose episode holds a
        # pedestrian route of more than 16 waypoints (> 32 m: what the 16-waypoint cut of rounds 1-3 could not hold)
        while max(len(h.waypoints) for h in env.humans) <= 16:
            seed += 1
            env = make_env(ref_env, human_policy, S, seed, indoor=True)
    rng = np.random.default_rng(seed)
    B = env.robot.n_angles
    occ = (env.map_info["data"] >= 0.1)
    if scenario == "crash":
        # face the west border wall (5 cells = 0.25 m thick) from 0.5 m outside the crash zone
        field = oracle.build_dt(occ)[0]
        rows = np.where((field[:, 27] == 23.0) & (field[:, 45] >= 30.0))[0]     # nothing but the wall nearby
        env.robot.px, env.robot.py, env.robot.theta = 0.25 + 0.6 + 0.5, (rows[len(rows) // 2] + 0.5) * 0.05, np.pi
    if scenario == "success":
        env.robot.gx = env.robot.px + 0.9 * np.cos(env.robot.theta)
        env.robot.gy = env.robot.py + 0.9 * np.sin(env.robot.theta)
    if scenario in ("peds", "crash"):
        # one legged and one legless pedestrian in view of the robot
        layout = ((2.0, 0.4, True), (3.0, -0.5, False), (1.6, 2.4, True)) if scenario == "peds" else \
                 ((2.0, 2.5, True), (3.0, -2.4, False), (1.6, 3.0, True))
        for k, (dist, bearing, legs) in enumerate(layout):
            h = env.humans[k]
            h.px = env.robot.px + dist * np.cos(env.robot.theta + bearing)
            h.py = env.robot.py + dist * np.sin(env.robot.theta + bearing)
            h.has_legs = legs
    first = rebuild_first_obs(env)
    rec = dict(
        occ_packed=np.packbits(occ), occ_shape=np.array(occ.shape),
        S=np.array(S), B=np.array(B),
        scan_threshold=env.scan_threshold.copy(), scan_discomfort=env.scan_discomfort_threshold.copy(),
        robot_goal=np.array([env.robot.gx, env.robot.gy]),
        ped_v_pref=np.array([h.v_pref for h in env.humans]),
        ped_has_legs=np.array([h.has_legs for h in env.humans], np.uint8),
        first_obs=first["observation"].copy(),
        time_step=np.array(env.time_step),
    )
    for k, v in snapshot(env).items():
        rec["init_" + k] = v
    # what the pedestrian control block (env.py:617-662) reads besides the scans: the waypoint lists and
    # the seed the stand-in HumanPolicy weights were drawn with (torch.manual_seed(seed) + default init)
    wmax = max(len(h.waypoints) for h in env.humans)
    wps = np.zeros((len(env.humans), wmax, 2))
    for i, h in enumerate(env.humans):
        wps[i, :len(h.waypoints)] = np.asarray(h.waypoints)
    rec["init_ped_waypoints"] = wps
    rec["init_ped_n_waypoints"] = np.array([len(h.waypoints) for h in env.humans], np.int32)
    rec["policy_seed"] = np.array(seed)
    # record the (v, w) the reference hands to Human.set_vel (env.py:662)
    cmds = []
    orig = human.Human.set_vel

    def spy(self, v, w):
        cmds.append((float(v), float(w)))
        return orig(self, v, w)
    human.Human.set_vel = spy
    T = n_steps
    N = len(env.humans)
    acts = np.zeros((T, 2)); obs = np.zeros((T, S * B + 7)); rew = np.zeros(T); done = np.zeros(T, np.uint8)
    succ = np.zeros(T, np.float32); crash = np.zeros(T, np.float32); dist = np.zeros(T)
    ped_mean = np.zeros((T, N, 2), np.float32)
    ped_cmd = np.zeros((T, N, 2)); snaps = {k: [] for k in ("robot_pose", "ped_pose", "ped_vel", "ped_dist")}
    # pedestrian scans (env.py:685-693, the input of HumanPolicy): latest 512-beam scan of every
    # pedestrian on a few steps (kept small; steps with a crash are skipped because those scans saw
    # the robot at the pre-revert pose, which the returned state no longer holds)
    ped_scan_steps, ped_scans = [], []
    try:
        for t in range(T):
            if scenario in ("random", "corridor"):
                a = np.array([rng.uniform(0, 0.5), rng.uniform(-0.64, 0.64)])
            elif scenario == "crash":
                a = np.array([0.5, 0.0]) if t < 8 else np.array([rng.uniform(0, 0.5), rng.uniform(-0.64, 0.64)])
            else:
                a = np.array([0.5, rng.uniform(-0.1, 0.1)])
            del cmds[:]
            o, r, d, info = env.step(a.copy())
            acts[t] = a; obs[t] = o["observation"]; rew[t] = r; done[t] = d
            succ[t] = info["is_success"]; crash[t] = info["is_crash"]; dist[t] = info["distance"]
            ped_cmd[t] = np.array(cmds).reshape(N, 2)
            ped_mean[t] = env.prev_human_actions                   # clip(mean) of HumanPolicy (env.py:655-658)
            if ped_scan_every and t % ped_scan_every == 0 and not info["is_crash"]:
                ped_scan_steps.append(t)
                ped_scans.append(np.stack([env.prev_humans_obs_queue[i][-1]["observation"][2 * 512:3 * 512]
                                           for i in range(N)]).astype(np.float32))
            for k, v in snapshot(env).items():
                snaps[k].append(v)
    finally:
        human.Human.set_vel = orig
    rec.update(actions=acts, obs_scan=obs[:, : S * B].astype(np.float32), obs_tail=obs[:, S * B:],
               reward=rew, done=done, is_success=succ, is_crash=crash, distance=dist, ped_cmd=ped_cmd,
               ped_mean=ped_mean)
    assert np.array_equal(rec["obs_scan"].astype(np.float64), obs[:, : S * B])   # scans are float32 values
    for k, v in snaps.items():
        rec["traj_" + k] = np.stack(v)
    if ped_scans:
        rec["ped_scan_steps"] = np.array(ped_scan_steps)
        rec["ped_scan"] = np.stack(ped_scans)
    np.savez_compressed(os.path.join(HERE, "golden_trace_%s.npz" % name), **rec)
    print("golden_trace_%s.npz: T=%d N=%d crashes=%d successes=%d longest route %d waypoints (seed %d)"
          % (name, T, N, int(crash.sum()), int(succ.sum()), int(rec["init_ped_n_waypoints"].max()), seed))
    return env


def make_long_routes(ref_env, env):
    """golden_long_routes.npz (round 4: routes of FULL length, env.py:788-804, 1261-1277).
    (1) the reference's own path_to_waypoints on 16 synthetic paths of 70-150 m at the costmap's 0.25 m spacing:
        30-70 waypoints at the 2 m interval;
    (2) the reference's own _sample_start_goal_path (env.py:342-383; pedestrian rule: goal more than 10 m away, no
        upper bound) on the costmap of `env` -- the 1000 x 1000 corridor episode of the corridor trace -- 24 times:
        start, goal, the path the planner stand-in returned and the reference's waypoints of it.  The path's LENGTH is
        what any shortest-path planner must reproduce; which of the equally short paths is taken is the stand-in's."""
    rng = np.random.default_rng(20261004)
    out = {}
    paths, wps, n_pts, n_wps = [], [], [], []
    for t in range(16):
        m = int(rng.integers(340, 720))
        # a walk with momentum on the 4-connected 0.25 m grid: long straight runs, so that it really travels
        dirs = np.array([[0.25, 0], [0, 0.25], [-0.25, 0], [0, -0.25]])
        d = int(rng.integers(0, 4))
        steps = []
        for _ in range(m):
            if rng.random() < 0.08:
                d = (d + int(rng.choice([1, 3]))) % 4
            steps.append(dirs[d])
        path = np.cumsum(np.vstack([[rng.uniform(5, 45, 2)], np.array(steps)]), axis=0)
        w = ref_env.path_to_waypoints(path, interval=2)
        pad = np.full((768, 2), np.nan); pad[: len(path)] = path
        wpad = np.full((128, 2), np.nan); wpad[: len(w)] = w
        paths.append(pad); wps.append(wpad); n_pts.append(len(path)); n_wps.append(len(w))
    out.update(wp_paths=np.stack(paths), wp_out=np.stack(wps), wp_n_points=np.array(n_pts), wp_n=np.array(n_wps))
    cm = env.cost_map_info
    cost = (np.asarray(cm["data"]) > 0).astype(np.uint8)
    out["cost_packed"] = np.packbits(cost); out["cost_shape"] = np.array(cost.shape)
    out["cost_resolution"] = np.array(cm["resolution"])
    starts, goals, lens, cells, rwps, rn, rpaths = [], [], [], [], [], [], []
    np.random.seed(77)
    while len(starts) < 24:
        start, goal, path = env._sample_start_goal_path(cm, 10, np.inf)
        w = ref_env.path_to_waypoints(path, interval=2)
        if len(starts) < 12 and len(w) <= 16:          # half of them longer than the old cut
            continue
        starts.append(start); goals.append(goal); cells.append(len(path))
        wpad = np.full((128, 2), np.nan); wpad[: len(w)] = w
        rwps.append(wpad); rn.append(len(w))
        ppad = np.full((768, 2), np.nan); ppad[: len(path)] = path
        rpaths.append(ppad)
    out.update(route_start=np.array(starts), route_goal=np.array(goals), route_cells=np.array(cells),
               route_wp=np.stack(rwps), route_n_wp=np.array(rn), route_path=np.stack(rpaths))
    np.savez_compressed(os.path.join(HERE, "golden_long_routes.npz"), **out)
    print("golden_long_routes.npz: synthetic paths %d-%d waypoints; reference routes %d-%d waypoints"
          % (min(n_wps), max(n_wps), min(rn), max(rn)))



# --------------------------------------------------------------------------------------------
# CrowdSim-v0 termination / reward block (crowd_sim.py:808-945): SURVEY.md 8f #4
# --------------------------------------------------------------------------------------------
def make_crowd():
    """Runs the reference's own CrowdSim.step(update=False) on stub agents: the collision tests
    (point_to_segment_dist against every agent, occupancy-grid windows around the robot), goal test and
    reward / info selection for 600 random situations.  gym, rvo2, tensorflow and the policy factory are
    import-only stand-ins; nothing of them executes in that block."""
    import importlib
    from collections import namedtuple
    for name in ("gym", "gym.envs", "gym.envs.registration", "rvo2", "cv2", "tensorflow", "PIL", "matplotlib",
                 "crowd_nav", "crowd_nav.policy", "crowd_nav.policy.policy_factory"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["gym"], "Env"):
        sys.modules["gym"].Env = object
    if not hasattr(sys.modules["gym.envs.registration"], "register"):
        sys.modules["gym.envs.registration"].register = lambda **k: None
    sys.modules["crowd_nav.policy.policy_factory"].policy_factory = {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from crowd_sim.envs import crowd_sim as cs
    codes = {"Nothing": 0, "Timeout": 1, "ReachGoal": 2, "Collision": 3, "CollisionOtherAgent": 4, "Danger": 5}
    Action = namedtuple("Action", ["v", "r"])

    class Pol(object):
        name = "SARL"
        def human_state_in_FOV(self, robot, agent):
            return True

    class Rob(object):
        policy = Pol()
        def compute_velocity(self, a):
            return self.nv
        def compute_position(self, a, dt):
            return self.npos
        def get_goal_position(self):
            return self.goal

    class Hum(object):
        robot_visible = False
        def get_observable_state(self):
            return None
        def act(self, ob):
            return None
        def get_next_observable_state(self, a):
            return None

    rng = np.random.default_rng(21)
    n, A, G = 600, 4, 200
    P = dict(time_step=0.25, discomfort_dist=0.2, map_size_m=20.0, map_resolution=0.1, success_reward=1.0,
             collision_penalty=-0.25, discomfort_penalty_factor=0.5, rotation_penalty_factor=-0.01,
             timeout_penalty=-0.125, time_limit=25.0)
    maps = np.ones((n, G, G), np.uint8)
    robot = np.zeros((n, 10)); agents = np.zeros((n, A, 5)); gtime = np.zeros(n)
    reward = np.zeros(n); done = np.zeros(n, np.uint8); code = np.zeros(n, np.int32); dmin = np.full(n, np.inf)
    for k in range(n):
        for _ in range(rng.integers(0, 4)):
            x0, y0 = rng.integers(0, G - 30, 2); w, h = rng.integers(2, 30, 2)
            maps[k, x0:x0 + w, y0:y0 + h] = 0
        pos = rng.uniform(-9.5, 9.5, 2)
        if k % 7 == 0:                                   # next to an occupied block
            occ = np.argwhere(maps[k] == 0)
            if len(occ):
                c = occ[rng.integers(len(occ))]
                pos = (c + rng.uniform(-4, 4, 2)) * 0.1 - 10.0
        nv = rng.uniform(-1, 1, 2)
        npos = pos + nv * 0.25
        goal = npos + rng.uniform(-0.25, 0.25, 2) if k % 11 == 0 else rng.uniform(-9, 9, 2)
        radius = 0.3
        r_act = 0.0 if k % 3 == 0 else rng.uniform(-1, 1)
        robot[k] = [pos[0], pos[1], npos[0], npos[1], nv[0], nv[1], goal[0], goal[1], radius, r_act]
        for a in range(A):
            far = rng.random() < 0.5
            off = rng.uniform(-6, 6, 2) if far else rng.uniform(-0.9, 0.9, 2)
            agents[k, a] = [pos[0] + off[0], pos[1] + off[1], rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.2, 0.4)]
        if k % 13 == 5:
            agents[k, 1, 2:4] = nv                       # zero relative velocity: the point-distance branch
        gtime[k] = rng.uniform(0, 25.4)
        env = object.__new__(cs.CrowdSim)
        rb = Rob(); rb.px, rb.py, rb.radius = pos[0], pos[1], radius
        rb.nv, rb.npos, rb.goal = (nv[0], nv[1]), (npos[0], npos[1]), (goal[0], goal[1])
        hs = []
        for a in range(A):
            h = Hum(); h.px, h.py, h.vx, h.vy, h.radius = agents[k, a]
            hs.append(h)
        env.humans, env.other_robots, env.robot = hs, [], rb
        env.human_policy, env.use_grid_map, env.phase = "none", False, "train"
        env.map = maps[k].astype(np.float64)
        env.global_time = gtime[k]
        env.static_obstacles_as_pedestrians = []
        for kk, v in P.items():
            setattr(env, kk, v)
        _, _, rew, dn, info = cs.CrowdSim.step(env, Action(0.0, r_act), update=False, compute_local_map=False)
        reward[k], done[k], code[k] = rew, dn, codes[type(info).__name__]
        if code[k] == 5:
            dmin[k] = info.min_dist
    np.savez_compressed(os.path.join(HERE, "golden_crowd.npz"), maps=np.packbits(maps), map_shape=np.array(maps.shape),
                        robot=robot, agents=agents, global_time=gtime, reward=reward, done=done, info=code, dmin=dmin,
                        params=np.array([P[k] for k in sorted(P)]), param_names=np.array(sorted(P)))
    print("golden_crowd.npz: n=%d" % n, {k: int((code == v).sum()) for k, v in codes.items()})


# --------------------------------------------------------------------------------------------
# CrowdSim-v0 local maps (crowd_sim.py:999-1186): SURVEY.md 8f #4
# --------------------------------------------------------------------------------------------
def make_crowd_maps():
    """get_local_map_angular (+ calculate_angular_map_distances) of the reference itself on 400 random situations
    (1-5 axis-aligned obstacles with the reference's vertex order, crowd_sim.py:250-258, robots anywhere incl.
    next to and inside obstacles' bounding boxes), and get_local_map's window logic (centre cell, clipping at the
    map border, exclusive slice ends, 0.9 threshold) on 200 situations with rotate_grid_around_center replaced by
    the identity (cv2 is not installed; the rotation is restated from OpenCV's documented algorithm and is
    UNPINNED: oracle/navsim_ref.c)."""
    import importlib
    from collections import namedtuple
    for name in ("gym", "gym.envs", "gym.envs.registration", "rvo2", "cv2", "tensorflow", "PIL", "matplotlib",
                 "crowd_nav", "crowd_nav.policy", "crowd_nav.policy.policy_factory"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["gym"], "Env"):
        sys.modules["gym"].Env = object
    if not hasattr(sys.modules["gym.envs.registration"], "register"):
        sys.modules["gym.envs.registration"].register = lambda **k: None
    sys.modules["crowd_nav.policy.policy_factory"].policy_factory = {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from crowd_sim.envs import crowd_sim as cs
    State = namedtuple("State", ["px", "py", "theta", "radius"])
    rng = np.random.default_rng(33)
    n, O, dim = 400, 5, 72
    P = dict(angular_min=-np.pi, angular_max=np.pi, angular_max_range=6.0, angular_dim=dim, normalize=1,
             map_size_m=14.0, map_resolution=0.1, submap_size_m=6.0)          # test_soadrl_static.config [map]
    robot = np.zeros((n, 4)); verts = np.zeros((n, O, 4, 2)); n_obst = np.zeros(n, np.int32)
    amap = np.zeros((n, dim))
    for k in range(n):
        no = int(rng.integers(1, O + 1))
        n_obst[k] = no
        obst = []
        for o in range(no):
            cx, cy = rng.uniform(-5, 5, 2); hx, hy = rng.uniform(0.15, 1.6, 2)
            vs = [(cx + hx, cy + hy), (cx - hx, cy + hy), (cx - hx, cy - hy), (cx + hx, cy - hy)]
            verts[k, o] = vs
            obst.append(vs)
        pos = rng.uniform(-6, 6, 2)
        if k % 5 == 0:                                       # close to an obstacle corner
            pos = verts[k, 0, rng.integers(4)] + rng.uniform(-0.6, 0.6, 2)
        robot[k] = [pos[0], pos[1], rng.uniform(-np.pi, np.pi), rng.uniform(0.2, 0.4)]
        env = object.__new__(cs.CrowdSim)
        env.angular_map_max_range, env.angular_map_dim = P["angular_max_range"], dim
        env.angular_map_min_angle, env.angular_map_max_angle = P["angular_min"], P["angular_max"]
        env.obstacle_vertices = obst
        env.local_maps_angular = []
        amap[k] = cs.CrowdSim.get_local_map_angular(env, State(*robot[k]))
    # get_local_map window logic, identity rotation
    m, G = 200, 140
    S = int(round(P["submap_size_m"] / P["map_resolution"]))
    maps = np.ones((m, G, G), np.uint8); rob2 = np.zeros((m, 4)); lmap = np.zeros((m, S, S), np.uint8)
    for k in range(m):
        for _ in range(rng.integers(1, 6)):
            x0, y0 = rng.integers(0, G - 20, 2); w, h = rng.integers(2, 25, 2)
            maps[k, x0:x0 + w, y0:y0 + h] = 0
        pos = rng.uniform(-6.9, 6.9, 2) if k % 2 else rng.uniform(-3.9, 3.9, 2)     # odd k: windows clipped by the border
        if k % 10 == 0:
            pos = np.round(pos, 1) + 0.05                    # ties of Python's round-half-even
        rob2[k] = [pos[0], pos[1], rng.uniform(-np.pi, np.pi), 0.3]
        env = object.__new__(cs.CrowdSim)
        env.map = maps[k].astype(np.float64)
        env.map_size_m, env.map_resolution, env.submap_size_m = P["map_size_m"], P["map_resolution"], P["submap_size_m"]
        env.local_maps = []
        env.rotate_grid_around_center = lambda grid, angle: grid          # stand-in: cv2 absent
        lmap[k] = cs.CrowdSim.get_local_map(env, State(*rob2[k]))
    np.savez_compressed(os.path.join(HERE, "golden_crowd_maps.npz"), robot=robot, verts=verts, n_obst=n_obst, amap=amap,
                        params=np.array([P[k] for k in sorted(P)]), param_names=np.array(sorted(P)),
                        maps=np.packbits(maps), map_shape=np.array(maps.shape), robot2=rob2, lmap=np.packbits(lmap),
                        lmap_shape=np.array(lmap.shape))
    print("golden_crowd_maps.npz: angular n=%d (%.1f %% of sectors below max range), windows m=%d (%.1f %% occupied)"
          % (n, 100 * (amap < 1.0).mean(), m, 100 * (lmap == 0).mean()))


def make_crowd_reset():
    """CrowdSim.reset (crowd_sim.py:626-722) -- the reference's own method, configured from its own config file
    (crowd_nav/config/test_soadrl_static.config: 5 humans, ORCA pedestrians, 14 m map at 0.1 m, up to 10 circles and 10 walls,
    randomize_attributes), for phases 'test' (square_crossing) and 'val' (circle_crossing), which seed NumPy's global stream
    with counter_offset + case number (crowd_sim.py:651-657): robot, humans, obstacle outlines, the occupancy map and the static
    obstacles as pedestrians of 12 + 12 cases.  gym / rvo2 / tensorflow are import-only stand-ins; the humans' policy object is
    the reference's ORCA class, which reset() only resets."""
    import configparser, importlib
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    for name in ("gym", "gym.envs", "gym.envs.registration", "rvo2", "cv2", "tensorflow", "PIL", "matplotlib",
                 "crowd_nav", "crowd_nav.policy", "crowd_nav.policy.policy_factory"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["gym"], "Env"):
        sys.modules["gym"].Env = object
    if not hasattr(sys.modules["gym.envs.registration"], "register"):
        sys.modules["gym.envs.registration"].register = lambda **k: None
    import warnings
    if not hasattr(sys.modules["crowd_nav.policy.policy_factory"], "policy_factory"):
        sys.modules["crowd_nav.policy.policy_factory"].policy_factory = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from crowd_sim.envs.policy.orca import ORCA
        sys.modules["crowd_nav.policy.policy_factory"].policy_factory["orca"] = ORCA
        from crowd_sim.envs import crowd_sim as cs
        cs.policy_factory = {"orca": ORCA}
        from crowd_sim.envs.utils.robot import Robot
    config = configparser.RawConfigParser()
    config.read(os.path.join(REF_SRC, "crowd_nav", "config", "test_soadrl_static.config"))

    class RobotPolicy(object):                      # any learning policy: reset() only reads its name
        name = "SARL"
        time_step = None

    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for phase in ("test", "val"):
            for case in list(range(10)) + [123, 499 if phase == "test" else 49]:
                env = cs.CrowdSim()
                env.configure(config, silent=True)
                robot = Robot(config, "robot")
                robot.set_policy(RobotPolicy())
                env.set_robot(robot)
                ob, local_map = env.reset(phase=phase, test_case=case)
                r = env.robot
                rows.append(dict(
                    phase=phase, case=case,
                    robot=np.array([r.px, r.py, r.gx, r.gy, r.vx, r.vy, r.theta, r.radius, r.v_pref]),
                    humans=np.array([[h.px, h.py, h.gx, h.gy, h.vx, h.vy, h.theta, h.radius, h.v_pref, float(h.robot_visible)]
                                     for h in env.humans]).reshape(-1, 10),
                    verts=np.array(env.obstacle_vertices, dtype=np.float64).reshape(-1, 4, 2),
                    map=np.asarray(env.map, dtype=np.uint8),
                    static=np.array([[s.px, s.py, s.vx, s.vy, s.radius] for s in env.static_obstacles_as_pedestrians]).reshape(-1, 5),
                    local_map=np.asarray(local_map, dtype=np.float64).reshape(-1),
                    circle_radius=np.float64(env.last_circle_radius)))
    out = {"n": np.int64(len(rows)), "phase": np.array([r["phase"] for r in rows]), "case": np.array([r["case"] for r in rows])}
    for k, r in enumerate(rows):
        for key in ("robot", "humans", "verts", "static", "local_map", "circle_radius"):
            out["%s_%d" % (key, k)] = r[key]
        out["map_%d" % k] = np.packbits(r["map"]); out["map_shape_%d" % k] = np.array(r["map"].shape)
    np.savez_compressed(os.path.join(HERE, "golden_crowd_reset.npz"), **out)
    print("golden_crowd_reset.npz: %d resets; humans per case %s; obstacles per case %s" % (
        len(rows), [len(r["humans"]) for r in rows], [len(r["verts"]) for r in rows]))


def make_crowd_agent():
    """Agent.step with an ActionRot (crowd_sim/envs/utils/agent.py:108-141), the reference's own method on 500 random
    states: new pose, velocity."""
    import importlib
    for name in ("gym", "gym.envs", "gym.envs.registration", "rvo2", "cv2", "tensorflow", "PIL", "matplotlib",
                 "crowd_nav", "crowd_nav.policy", "crowd_nav.policy.policy_factory"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["gym"], "Env"):
        sys.modules["gym"].Env = object
    if not hasattr(sys.modules["gym.envs.registration"], "register"):
        sys.modules["gym.envs.registration"].register = lambda **k: None
    sys.modules["crowd_nav.policy.policy_factory"].policy_factory = {}
    sys.path.insert(0, REF_SRC)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from crowd_sim.envs.utils.agent import Agent
    from crowd_sim.envs.utils.action import ActionRot
    rng = np.random.default_rng(44)
    n = 500
    pose = np.concatenate([rng.uniform(-8, 8, (n, 2)), rng.uniform(0, 2 * np.pi, (n, 1))], axis=1)
    action = np.stack([rng.uniform(0, 1.2, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
    out = np.zeros((n, 3)); vel = np.zeros((n, 2))
    for k in range(n):
        a = object.__new__(Agent)
        a.px, a.py, a.theta, a.time_step = pose[k, 0], pose[k, 1], pose[k, 2], 0.25
        a.vx = a.vy = 0.0
        Agent.step(a, ActionRot(action[k, 0], action[k, 1]))
        out[k] = [a.px, a.py, a.theta]; vel[k] = [a.vx, a.vy]
    np.savez_compressed(os.path.join(HERE, "golden_crowd_agent.npz"), pose=pose, action=action, time_step=np.float64(0.25),
                        pose_out=out, vel_out=vel)
    print("golden_crowd_agent.npz: n=%d" % n)


from make_reset import make_reset  # noqa: E402


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "crowd_maps":      # only this fixture (the others stay byte-identical)
        return make_crowd_maps()
    if len(sys.argv) > 1 and sys.argv[1] == "crowd_agent":
        return make_crowd_agent()
    if len(sys.argv) > 1 and sys.argv[1] == "crowd_reset":
        return make_crowd_reset()
    ref_env, human, keti_robot, ref_utils, human_policy = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "reset":            # only golden_reset.npz
        return make_reset(ref_env)
    if len(sys.argv) > 1 and sys.argv[1] == "routes":           # only the round-4 fixtures (the others stay byte-identical)
        env = run_trace("corridor_S1", ref_env, human, human_policy, S=1, seed=21, scenario="corridor", n_steps=100, ped_scan_every=20)
        return make_long_routes(ref_env, env)
    make_units(ref_env, human, keti_robot, ref_utils)
    run_trace("random_S1", ref_env, human, human_policy, S=1, seed=11, scenario="random", n_steps=40, ped_scan_every=5)
    run_trace("peds_S1", ref_env, human, human_policy, S=1, seed=12, scenario="peds", n_steps=30, ped_scan_every=3)
    run_trace("crash_S3", ref_env, human, human_policy, S=3, seed=13, scenario="crash", n_steps=24)
    run_trace("success_S2", ref_env, human, human_policy, S=2, seed=14, scenario="success", n_steps=12)
    env = run_trace("corridor_S1", ref_env, human, human_policy, S=1, seed=21, scenario="corridor", n_steps=100, ped_scan_every=20)
    make_long_routes(ref_env, env)
    make_reset(ref_env)
    make_crowd()
    make_crowd_maps()
    make_crowd_agent()
    make_crowd_reset()


if __name__ == "__main__":
    main()
