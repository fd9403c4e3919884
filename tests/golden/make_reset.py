"""golden_reset.npz -- the reset path "draws supplied" (VERDICT r2 item 2).  Part of make_golden.py
(`python tests/golden/make_golden.py reset`); runs in the build container only.

The reference's own generators and spawn loops are replayed on a recorded sequence of uniforms:
np.random.random / choice / uniform are replaced by functions that map the NEXT supplied uniform u the way the
build maps its hash-keyed ones:  random() -> u;  uniform(a, b) -> a + (b - a) u;  choice(seq) -> seq[int(u len(seq))].
Only the random number generator stays build-defined; everything the reference computes FROM the draws is pinned:
  part A  _sample_env_param (env.py:281-292), the map-kind coin (env.py:295), create_indoor_map / create_outdoor_map
          (map_generator.py:97-143) and the costmap (env.py:312-332) on 8 supplied draw tapes (NAVSIM_DRAW_* layout);
  part B  every candidate (start, goal) the reference's reset() drew in 6 episodes and what its
          _sample_start_goal_path / robot loop (env.py:342-383, 748-806) did to it.
Only data is written.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class Feed(object):
    """Patches np.random.{random, choice, uniform} to consume `draws` in order (a list, or a callable u())."""

    def __init__(self, draws, log=None):
        self.next_u = draws if callable(draws) else iter(draws).__next__
        self.n = 0
        self.log = log                       # event list: every uniform() call leaves a ("uniform",) mark

    def u(self):
        self.n += 1
        return float(self.next_u())

    def __enter__(self):
        self.saved = (np.random.random, np.random.choice, np.random.uniform)
        np.random.random = lambda *a: self.u()

        def uniform(lo=0.0, hi=1.0, size=None):
            if self.log is not None:
                self.log.append(("uniform",))
            return lo + (hi - lo) * self.u()
        np.random.uniform = uniform

        def choice(seq, *a, **k):
            seq = list(seq) if isinstance(seq, range) else np.asarray(seq)
            return seq[int(self.u() * len(seq))]
        np.random.choice = choice
        return self

    def __exit__(self, *exc):
        np.random.random, np.random.choice, np.random.uniform = self.saved


# NAVSIM_DRAW_* of include/navsim.h
N_SCENES = 16
D_KIND, D_NOBS, D_NHUM, D_NOISE, D_OWID, D_CWID, D_ITER, D_MAP, D_N = 0, 1, 2, 3, 4, 5, 6, 8, 464


def make_reset(ref_env):
    kw = dict(sys.modules["gym"].registry["NavGym-v0"]["kwargs"])
    epr = dict(kw["env_param_range"])
    epr["obstacle_number"] = ([6, 14], "int")        # the registered [10, 10] never exercises the 'int' draw
    rng = np.random.default_rng(20261003)
    out = {"env_param_keys": np.array(list(epr.keys())), "indoor_ratio": np.float64(kw["indoor_ratio"]),
           "env_param_lo": np.array([epr[k][0][0] for k in epr], np.float64),
           "env_param_hi": np.array([epr[k][0][1] for k in epr], np.float64)}
    # ---- part A: _sample_env_param + _sample_map (create_indoor_map / create_outdoor_map) on supplied draws ----------
    n_a = 8
    tapes = rng.random((n_a, D_N))
    tapes[:, D_KIND] = [0.1, 0.9, 0.3, 0.7, 0.49999, 0.5, 0.2, 0.95]     # < 0.5 indoor; exactly 0.5 is OUTDOOR (strict <)
    tapes[2, D_NOBS] = 0.999999; tapes[3, D_NOBS] = 0.0                  # both ends of the 'int' interval
    tapes[4, D_CWID] = 0.999999; tapes[6, D_ITER] = 0.0; tapes[0, D_ITER] = 0.999999
    params = np.zeros((n_a, len(epr)))
    kinds = np.zeros(n_a, np.int32)
    maps, grids, costs = [], [], []
    order = {"num_humans": D_NHUM, "corridor_width": D_CWID, "iterations": D_ITER, "obstacle_number": D_NOBS,
             "obstacle_width": D_OWID, "scan_noise_std": D_NOISE}
    for a in range(n_a):
        T = tapes[a]
        env = object.__new__(ref_env.NavGymEnv)
        env.env_param_range, env.indoor_ratio = epr, kw["indoor_ratio"]
        with Feed([T[order[k]] for k in epr]) as f:                     # dict order = the order the reference draws in
            env.env_param = env._sample_env_param()
            assert f.n == len(epr)
        params[a] = [env.env_param[k] for k in epr]
        with Feed([T[D_KIND]] + list(T[D_MAP:])) as f:                  # the kind coin, then the generator's own draws
            env._sample_map()
            used = f.n - 1
        data = env.map_info["data"]
        size = data.shape[0]
        kinds[a] = int(size == 1000)
        assert (size == 1000) == (T[D_KIND] < kw["indoor_ratio"])
        assert used == (3 * env.env_param["iterations"] if kinds[a] else 2 * env.env_param["obstacle_number"]), used
        occ = (data >= 0.1)
        if kinds[a]:                                                    # 1000 x 1000 = every coarse cell 10 x 10 times
            g = occ[::10, ::10]
            assert np.array_equal(np.kron(g, np.ones((10, 10), bool)), occ)
            grids.append(np.packbits(g))
        else:
            maps.append(np.packbits(occ))
        costs.append(np.packbits(env.cost_map_info["data"] > 0))
        print("  reset A%d: %s map, params %s" % (a, "indoor 1000" if kinds[a] else "outdoor 400",
                                                  {k: round(float(v), 4) for k, v in env.env_param.items()}))
    out.update(tapes=tapes, params=params, kinds=kinds, indoor_grids=np.stack(grids), outdoor_maps=np.stack(maps),
               costmaps_packed=np.concatenate(costs), costmap_sizes=np.array([200 if k else 80 for k in kinds]))

    # ---- part B: the acceptance decisions of _sample_start_goal_path and of reset()'s robot loop ---------------------
    # Full reset() of the reference on uniforms from a seeded generator; every candidate it draws is logged through
    # wrappers around the functions it calls (ij_to_xy, np.linalg.norm, pyastar2d.astar_path), and WHAT HAPPENED to a
    # candidate is read off the control flow (which call follows), never re-derived from the values.
    kw2 = dict(kw)
    e2 = dict(kw["env_param_range"]); e2["scan_noise_std"] = ([0., 0.], "float"); e2["num_humans"] = ([6, 8], "int")
    kw2["env_param_range"] = e2
    ev = []
    real_ij_to_xy, real_norm, real_astar = ref_env.ij_to_xy, np.linalg.norm, sys.modules["pyastar2d"].astar_path
    real_ssgp = ref_env.NavGymEnv._sample_start_goal_path
    real_ptw = ref_env.path_to_waypoints

    def w_ij_to_xy(ij, mi):
        r = real_ij_to_xy(ij, mi)
        ev.append(("xy", np.array(r, np.float64)))
        return r

    mute = [False]

    def w_norm(x, *a, **k):
        r = real_norm(x, *a, **k)
        if not mute[0]:
            ev.append(("norm", np.array(x, np.float64).copy(), float(r)))
        return r

    def w_astar(grid, s_, g_, allow_diagonal=False):
        r = real_astar(grid, s_, g_, allow_diagonal=allow_diagonal)
        ev.append(("astar", r is not None))
        return r

    def w_ssgp(self, map_info, dmin, dmax, start=None, robot_pose=None):
        ev.append(("call", None if robot_pose is None else np.array(robot_pose, np.float64), float(dmin), float(dmax)))
        r = real_ssgp(self, map_info, dmin, dmax, start=start, robot_pose=robot_pose)
        ev.append(("ret", r[0] is not None))
        return r

    def w_ptw(path, interval):
        ev.append(("ptw", float(interval)))
        mute[0] = True                               # path_to_waypoints takes norms of its own (env.py:1265)
        try:
            return real_ptw(path, interval)
        finally:
            mute[0] = False
    torch.manual_seed(5)
    from nav_gym_env import human_policy as hp
    weights = hp.HumanPolicy(frames=3, action_space=2).state_dict()
    real_load = torch.load
    torch.load = lambda *a, **k: weights
    cands = []          # (scene, kind, sx, sy, gx, gy, rx, ry, code, path_distance / (2 |goal - start|))
    scene_cost, scene_size = [], []
    g2 = np.random.default_rng(77)
    try:
        with Feed(lambda: g2.random(), log=ev):
            env = ref_env.NavGymEnv(**kw2)          # __init__ runs reset() twice (env.py:163, 173)
            ref_env.ij_to_xy, np.linalg.norm = w_ij_to_xy, w_norm
            sys.modules["pyastar2d"].astar_path = w_astar
            ref_env.pyastar2d.astar_path = w_astar
            ref_env.NavGymEnv._sample_start_goal_path = w_ssgp
            ref_env.path_to_waypoints = w_ptw
            for scene in range(N_SCENES):
                del ev[:]
                env.reset()
                scene_cost.append(np.packbits(env.cost_map_info["data"] > 0))
                scene_size.append(env.cost_map_info["data"].shape[0])
                cands += parse_spawn_events(ev, scene)
            # ---- a pair that reset() drops for its PATH (env.py:756-762): a 400 x 400 map split by a wall whose only gap is
            # at the far end, handed to the reference's _sample_map in place of create_outdoor_map; the first pair reset()
            # draws faces the wall 19 m apart (the way round is > 80 m), the second one is easy
            wall_map = np.zeros((400, 400), np.int8)
            wall_map[:5, :] = 100; wall_map[-5:, :] = 100; wall_map[:, :5] = 100; wall_map[:, -5:] = 100
            wall_map[:340, 198:203] = 100                                # data[j, i]: a wall along y at x = 10 m, gap at the top
            real_com = ref_env.create_outdoor_map
            ref_env.create_outdoor_map = lambda n_obs, width: dict(data=wall_map.copy(), origin=(0, 0), resolution=0.05,
                                                                   width=400, height=400)
            try:
                with Feed([0.9]):
                    env._sample_map()                                    # dry run: the costmap reset() will build again
                rs, cs = np.where(env.cost_map_info["data"].T == 0)
                index = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(rs, cs))}
                pick = lambda i, j: (index[(i, j)] + 0.5) / len(rs)
                crafted = [0.5] * 6 + [0.9] + [pick(10, 12), pick(70, 12), pick(12, 12), pick(12, 60)]

                def crafted_then_random(it=iter(crafted)):       # env_param, kind coin, two robot pairs; then whatever
                    v = next(it, None)
                    return g2.random() if v is None else v
                del ev[:]
                with Feed(crafted_then_random, log=ev):
                    env.reset()
            finally:
                ref_env.create_outdoor_map = real_com
            scene_cost.append(np.packbits(env.cost_map_info["data"] > 0))
            scene_size.append(env.cost_map_info["data"].shape[0])
            got = parse_spawn_events(ev, N_SCENES)
            print("  reset B walled map: robot codes %s" % [int(q[8]) for q in got if q[1] == 0])
            cands += got
            # ---- hand-picked candidates through the same wrapped method of the reference: the equalities of its three
            # inequalities (a start EXACTLY 4.0 m from the robot is kept; a goal EXACTLY 10.0 m or 20.0 m from the start
            # is dropped) and pairs in range that no path joins (a wall across the map)
            for scene, wall in ((N_SCENES + 1, False), (N_SCENES + 2, True)):
                data = np.zeros((200, 200), np.uint8)
                data[:2, :] = 100; data[-2:, :] = 100; data[:, :2] = 100; data[:, -2:] = 100
                if wall:
                    data[:, 100] = 100                                   # data[j, i]: column i = 100 blocked for every j
                cmi = dict(data=data, origin=(0, 0), resolution=0.25, width=200, height=200)
                rs, cs = np.where(data.T == 0)                           # the reference's own enumeration (env.py:356)
                index = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(rs, cs))}
                pick = lambda i, j: (index[(i, j)] + 0.5) / len(rs)      # the uniform that makes choice() take cell (i, j)
                del ev[:]
                if not wall:
                    # robot pairs: |goal - start| = 10.0 (dropped), 20.0 (dropped), 10.25 (kept)
                    with Feed([pick(20, 20), pick(60, 20), pick(20, 30), pick(100, 30), pick(20, 40), pick(61, 40)]):
                        env._sample_start_goal_path(cmi, 10, 20)
                    # pedestrians around a robot in the centre of cell (20, 20): 3.75 m (dropped), 4.0 m (kept) with a
                    # goal 10.0 m away (dropped); 4.0 m the other way (kept) with a goal 10.25 m away (kept)
                    robot = np.array([20.5 * 0.25, 20.5 * 0.25])
                    with Feed([pick(35, 20), pick(36, 20), pick(76, 20), pick(20, 36), pick(20, 77)]):
                        env._sample_start_goal_path(cmi, 10, np.inf, robot_pose=robot)
                else:
                    with Feed([pick(60, 50), pick(120, 50), pick(90, 150), pick(150, 150), pick(30, 60), pick(90, 60)]):
                        env._sample_start_goal_path(cmi, 10, 20)
                    robot = np.array([20.5 * 0.25, 20.5 * 0.25])
                    with Feed([pick(70, 120), pick(130, 120), pick(110, 30), pick(160, 30)]):
                        env._sample_start_goal_path(cmi, 10, np.inf, robot_pose=robot)
                scene_cost.append(np.packbits(data > 0))
                scene_size.append(200)
                got = parse_spawn_events(ev, scene)
                print("  reset B hand-picked scene %d: codes %s" % (scene, [int(g[8]) for g in got]))
                cands += got
    finally:
        torch.load = real_load
        ref_env.ij_to_xy, np.linalg.norm = real_ij_to_xy, real_norm
        sys.modules["pyastar2d"].astar_path = real_astar
        ref_env.pyastar2d.astar_path = real_astar
        ref_env.NavGymEnv._sample_start_goal_path = real_ssgp
        ref_env.path_to_waypoints = real_ptw
    c = np.array(cands, np.float64)
    out.update(cand=c, scene_costmaps_packed=np.concatenate(scene_cost), scene_costmap_sizes=np.array(scene_size))
    codes, counts = np.unique(c[:, 8].astype(int), return_counts=True)
    print("  reset B: %d candidates over %d scenes, codes %s" % (len(c), len(scene_size), dict(zip(codes.tolist(), counts.tolist()))))
    np.savez_compressed(os.path.join(HERE, "golden_reset.npz"), **out)
    print("golden_reset.npz:", len(out), "arrays")


def parse_spawn_events(ev, scene):
    """Event log of one reset() -> candidates with what the reference did to them.  Inside one
    _sample_start_goal_path call an iteration is   xy(start) [norm(robot - start)] xy(goal) norm(start - goal) [astar];
    an iteration that ends early is recognised by what comes next.  code: 0 kept, 1 start too close to the robot,
    2 goal distance outside the interval, 3 no path, 4 (robot) reset() dropped the pair: path_distance > 2 |goal - start|
    (env.py:761; path_distance / (2 |goal - start|) is recorded beside every planned robot pair), 5 (robot) kept by all
    of these, then dropped because the first scan at the drawn pose is inside the discomfort zone (env.py:776-781)."""
    out = []
    i = 0
    n = len(ev)
    while i < n:
        if ev[i][0] != "call":
            i += 1
            continue
        robot = ev[i][1]
        kind = 0 if robot is None else 1
        j = i + 1
        last = None
        while ev[j][0] != "ret":
            assert ev[j][0] == "xy", ev[j][0]
            start = ev[j][1]; j += 1
            if kind == 1:
                assert ev[j][0] == "norm" and np.allclose(ev[j][1], robot - start)
                j += 1
                if ev[j][0] == "xy" and ev[j + 1][0] == "norm" and np.allclose(ev[j + 1][1], robot - ev[j][1]):
                    out.append([scene, kind, start[0], start[1], np.nan, np.nan, robot[0], robot[1], 1, 0.0])
                    continue                                        # the next iteration drew a new start
            goal = ev[j][1]; j += 1
            assert ev[j][0] == "norm" and np.allclose(ev[j][1], start - goal)
            j += 1
            rx, ry = (np.nan, np.nan) if robot is None else (robot[0], robot[1])
            if ev[j][0] == "astar":
                found = ev[j][1]; j += 1
                out.append([scene, kind, start[0], start[1], goal[0], goal[1], rx, ry, 0 if found else 3, 0.0])
                last = len(out) - 1 if found else None
            else:
                out.append([scene, kind, start[0], start[1], goal[0], goal[1], rx, ry, 2, 0.0])
        assert ev[j][1] is True or last is None
        # reset()'s own test of the robot's pair (env.py:756-762): path_to_waypoints(path, 5), the norms of the
        # polyline, then EITHER another robot call (dropped) OR the heading draw (kept)
        if kind == 0 and last is not None and j + 1 < n and ev[j + 1][0] == "ptw":    # (not after a bare method call)
            k = j + 1
            assert ev[k][1] == 5.0
            k += 1
            norms = []
            while k < n and ev[k][0] == "norm":
                norms.append(ev[k][2]); k += 1
            straight = norms[-1]                                   # the last norm is |goal - start|
            plen = sum(norms[:-1])
            out[last][9] = plen / (2.0 * straight)
            if k < n and ev[k][0] == "call" and ev[k][1] is None:
                out[last][8] = 4                                   # `continue` at env.py:762: straight into the next try
            else:
                # kept by the path test: the heading is drawn (env.py:763).  If the robot loop then runs AGAIN, the first
                # scan at that pose found an obstacle inside the discomfort zone (env.py:776-781): code 5 -- kept by every
                # spawn RULE, dropped by the scan
                assert ev[k][0] == "uniform", ev[k][0]
                k += 1
                if k < n and ev[k][0] == "call" and ev[k][1] is None:
                    out[last][8] = 5
        i = j + 1
    return out
