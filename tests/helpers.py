"""Shared test helpers: build simulator inputs from golden traces and synthetic generators."""
import os

import numpy as np

from nav_gym_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_trace(name):
    d = np.load(os.path.join(GOLDEN, "golden_trace_%s.npz" % name))
    return {k: d[k] for k in d.files}


def trace_setup(tr, default_config, build_dt):
    """Returns (cfg, arrays) reproducing the initial state of a golden trace (E = 1).

    default_config / build_dt come from whichever implementation is under test."""
    H, W = [int(x) for x in tr["occ_shape"]]
    occ = np.unpackbits(tr["occ_packed"])[: H * W].reshape(H, W).astype(np.uint8)
    N = tr["init_ped_pose"].shape[0]
    S, B = int(tr["S"]), int(tr["B"])
    cfg = default_config(n_envs=1, n_beams=B, map_h=H, map_w=W, max_peds=max(N, 1), n_scan_stack=S,
                         ped_model=abi.PED_EXTERNAL, lidar_legs=1, time_step=float(tr["time_step"]))
    arrays = dict(
        field=build_dt(occ[None]),
        scan_threshold=tr["scan_threshold"].astype(np.float32),
        scan_discomfort=tr["scan_discomfort"].astype(np.float32),
        scan_noise_std=np.zeros(1, np.float32),
        robot_pose=tr["init_robot_pose"][None].copy(),
        robot_goal=tr["robot_goal"][None].copy(),
        prev_action=np.zeros((1, 2)), prev_pose=np.zeros((1, 3)),
        n_hist=np.zeros(1, np.int32), episode=np.zeros(1, np.int64), steps=np.zeros(1, np.int64),
        n_peds=np.array([N], np.int32),
        ped_pose=tr["init_ped_pose"][None].copy(),
        ped_vel=tr["init_ped_vel"][None].copy(),
        ped_prev_yaw=np.zeros((1, N)), ped_dist=tr["init_ped_dist"][None].copy(),
        ped_v_pref=tr["ped_v_pref"][None].copy(),
        ped_has_legs=tr["ped_has_legs"][None].astype(np.uint8),
        ped_waypoints=np.zeros((1, N, cfg.max_waypoints, 2)),
        ped_n_waypoints=np.ones((1, N), np.int32),
        ped_cmd=np.zeros((1, N, 2)),
    )
    # waypoints are irrelevant to the dynamics under PED_EXTERNAL; park them far away
    arrays["ped_waypoints"][...] = 1.0e6
    return cfg, arrays, occ


def outdoor_map(rng, size, n_obstacles=10, width_range=(0.3, 1.0)):
    """Restatement of create_outdoor_map (map_generator.py:126-143) scaled to size x size:
    5-cell border wall, `n_obstacles` squares of half-width int(10*U[width_range]) cells."""
    m = np.ones((size, size), np.uint8)
    m[5:size - 5, 5:size - 5] = 0
    hw = int(10 * rng.uniform(*width_range))
    for _ in range(n_obstacles):
        cx = rng.integers(hw + 2, size - hw - 1)
        cy = rng.integers(hw + 2, size - hw - 1)
        m[cx - hw:cx + hw + 1, cy - hw:cy + hw + 1] = 1
    return np.flipud(m).copy()


def finished_world(cfg, occ, field, n_peds, thresholds):
    """Host arrays for a RefSim whose robots already stand on their goals (every arena finishes in
    the first step), for exercising navsim_regen without a rollout.  thresholds = (thr, dthr)."""
    E, N, K = cfg.n_envs, cfg.max_peds, max(cfg.n_spawn, 1)
    res = cfg.resolution
    pose = np.zeros((E, 3))
    for e in range(E):
        j, i = np.unravel_index(np.argmax(field[e]), field[e].shape)
        pose[e, :2] = ((i + 0.5) * res + cfg.origin_x, (j + 0.5) * res + cfg.origin_y)
    return dict(
        field=field, scan_threshold=thresholds[0], scan_discomfort=thresholds[1],
        scan_noise_std=np.zeros(E, np.float32),
        robot_pose=pose, robot_goal=pose[:, :2].copy(),
        prev_action=np.zeros((E, 2)), prev_pose=np.zeros((E, 3)),
        n_hist=np.zeros(E, np.int32), episode=np.zeros(E, np.int64), steps=np.zeros(E, np.int64),
        n_peds=np.full(E, n_peds, np.int32),
        ped_pose=np.full((E, N, 3), 1.0e6), ped_vel=np.zeros((E, N, 2)),
        ped_prev_yaw=np.zeros((E, N)), ped_dist=np.zeros((E, N, 3)),
        ped_v_pref=np.ones((E, N)), ped_has_legs=np.ones((E, N), np.uint8),
        ped_waypoints=np.full((E, N, cfg.max_waypoints, 2), 1.0e6),
        ped_n_waypoints=np.ones((E, N), np.int32), ped_cmd=np.zeros((E, N, 2)),
        spawn_pose=np.tile(pose[:, None, :], (1, K, 1)), spawn_goal=np.tile(pose[:, None, :2], (1, K, 1)),
    )


def policy_weights(seed):
    """The actor weights of the stand-in HumanPolicy the golden traces were recorded with:
    tests/golden/make_golden.py seeds torch and takes the default initialisation of the reference's
    module (human_policy.py:19-37).  The actor's layers are created first, in this order, so the same
    seed and the same layer shapes reproduce them without importing the reference."""
    import torch
    torch.manual_seed(int(seed))
    cv1 = torch.nn.Conv1d(3, 32, kernel_size=5, stride=2, padding=1)
    cv2 = torch.nn.Conv1d(32, 32, kernel_size=3, stride=2, padding=1)
    fc1 = torch.nn.Linear(128 * 32, 256)
    fc2 = torch.nn.Linear(256 + 2 + 2, 128)
    a1 = torch.nn.Linear(128, 1)
    a2 = torch.nn.Linear(128, 1)
    f = lambda t: t.detach().numpy().astype(np.float32).copy()
    return dict(cv1_w=f(cv1.weight), cv1_b=f(cv1.bias), cv2_w=f(cv2.weight), cv2_b=f(cv2.bias),
                fc1_w=f(fc1.weight), fc1_b=f(fc1.bias), fc2_w=f(fc2.weight), fc2_b=f(fc2.bias),
                a1_w=f(a1.weight).reshape(-1), a1_b=f(a1.bias), a2_w=f(a2.weight).reshape(-1), a2_b=f(a2.bias))
