"""CrowdSim-v0 `step()` for E envs at once (nav_gym/src/crowd_sim/envs/crowd_sim.py:724-997), composed from the
device entry points of include/navsim.h:

    pedestrians' actions      navsim_crowd_orca        ORCA.predict per pedestrian (policy/orca.py:85-135)
    collision / goal / reward navsim_crowd_check       crowd_sim.py:808-949
    agent update              navsim_crowd_agent_step  Agent.step (utils/agent.py:108-141), crowd_sim.py:951-958
    robot's local map         navsim_crowd_angular_map / navsim_crowd_local_map   crowd_sim.py:999-1186

Host side is torch plumbing only (building each pedestrian's query: itself first, then the agents it sees).  The
reference's learning policies (SARL / CADRL / SDOADRL) are out of scope: the robot's action comes from the caller.
Pedestrians see everything (the default field of view of orca.py:67-68 is the full circle).
"""
import math

import numpy as np

from . import sim

# The values of the reference's one environment file, crowd_nav/config/test_soadrl_static.config (sections env / reward /
# map / humans / robot), as keyword defaults of CrowdSimEnv; CrowdSim.configure (crowd_sim.py:89-146) reads the same names.
CROWD_DEFAULTS = dict(
    time_limit=35, time_step=0.2, val_size=50, test_size=500, randomize_attributes=True,
    train_val_sim="circle_crossing", test_sim="square_crossing", square_width=10.0, circle_radius_min=4.0, human_num=5,
    success_reward=1.0, collision_penalty=-0.25, discomfort_dist=0.2, discomfort_penalty_factor=0.5, timeout_penalty=0.0,
    rotation_penalty_factor=-0.003,
    use_grid_map=False, map_size_m=14.0, submap_size_m=6.0, map_resolution=0.1, angular_map_max_range=6.0, num_circles=10,
    num_walls=10, angular_map_dim=72, angle_min=-1.0, angle_max=1.0,
    human_radius=0.3, human_v_pref=1.0, robot_radius=0.3, robot_v_pref=1.0, robot_visible=True)
CROWD_CASE_CAPACITY = dict(train=int(np.iinfo(np.uint32).max) - 2000, val=1000, test=1000)      # crowd_sim.py:107-111


def _rint(x):
    return int(round(x))                                # the reference's int(round(.)): Python rounds half to even


def _norm2(x, y):
    return float(np.linalg.norm((x, y)))                # numpy.linalg.norm of a pair, as the reference calls it


def crowd_reset_scenario(cfg, phase, case, seed=None):
    """One episode's initial world exactly as CrowdSim.reset draws it (crowd_sim.py:626-722 with
    generate_random_human_position :407-440, generate_circle_crossing :442-497, generate_square_crossing :499-547,
    generate_static_map_input :194-370, create_observation_from_static_obstacles :372-405): the same draws from NumPy's legacy
    stream in the same order, so phases 'val' and 'test' -- which seed it with counter_offset + case (crowd_sim.py:645-657) --
    reproduce the reference's own cases (tests/golden/golden_crowd_reset.npz).  'train' seeds from `seed` (the reference:
    from the OS).  Host side, NumPy only: reset is not on the hot path.
    -> dict(robot [9] px py gx gy vx vy theta radius v_pref, humans [n,10] (the same + robot_visible), verts [m,4,2],
            free_map [G,G] uint8 (1 = free, indexed [x][y]), static [k,5] px py vx vy radius, circle_radius)."""
    assert phase in ("train", "val", "test")
    offset = {"train": CROWD_CASE_CAPACITY["val"] + CROWD_CASE_CAPACITY["test"], "val": 0, "test": CROWD_CASE_CAPACITY["val"]}
    rs = np.random.RandomState(seed if phase == "train" else offset[phase] + int(case))
    rnd = rs.random_sample
    dd = float(cfg["discomfort_dist"])
    r_vpref, r_radius = float(cfg["robot_v_pref"]), float(cfg["robot_radius"])
    if cfg["randomize_attributes"]:                     # Agent.sample_random_attributes (agent.py:39-44)
        r_vpref = rs.uniform(0.8, 1.2); r_radius = rs.uniform(0.3, 0.5)
    cr = float(cfg["circle_radius_min"]) * min(r_vpref * 5, 1) * (1 + rnd() * 2)
    cr = 9 if cr > 9 else cr
    robot = np.array([0.0, -cr, 0.0, cr, 0.0, 0.0, np.pi / 2, r_radius, r_vpref])
    n_h = _rint(cfg["human_num"] * (0.5 + rnd()))
    rnd()                                               # other_robots_num = round(robot_num * (0.5 + random())): robot_num = 0
    rule = cfg["train_val_sim"] if phase in ("train", "val") else cfg["test_sim"]
    if rule not in ("square_crossing", "circle_crossing"):
        raise ValueError("Rule doesn't exist")
    placed = [robot]                                    # [robot] + humans: rows px py gx gy ... radius at [7]
    humans = []
    for _ in range(n_h):
        v_pref, radius = float(cfg["human_v_pref"]), float(cfg["human_radius"])
        if cfg["randomize_attributes"]:
            v_pref = rs.uniform(0.8, 1.2); radius = rs.uniform(0.3, 0.5)
        if rule == "circle_crossing":
            while True:
                angle = rnd() * np.pi * 2
                px_noise = (rnd() - 0.5) * v_pref
                py_noise = (rnd() - 0.5) * v_pref
                c = float(cfg["circle_radius_min"]) * (1 + rnd() * 1.5)
                px = c * np.cos(angle) + px_noise
                py = c * np.sin(angle) + py_noise
                if not any(_norm2(px - o[0], py - o[1]) < radius + o[7] + dd or _norm2(px - o[2], py - o[3]) < radius + o[7] + dd
                           for o in placed):
                    break
            gx, gy = -px, -py
        else:
            sign = -1 if rnd() > 0.5 else 1
            sw = float(cfg["square_width"])
            while True:
                px = rnd() * sw * 0.5 * sign
                py = (rnd() - 0.5) * sw
                if not any(_norm2(px - o[0], py - o[1]) < radius + o[7] + dd for o in placed):
                    break
            while True:
                gx = rnd() * sw * 0.5 * -sign
                gy = (rnd() - 0.5) * sw
                if not any(_norm2(gx - o[2], gy - o[3]) < radius + o[7] + dd for o in placed):
                    break
        sees = bool(rnd() > 0.5) and bool(cfg["robot_visible"])
        row = np.array([px, py, gx, gy, 0.0, 0.0, 0.0, radius, v_pref, float(sees)])
        humans.append(row); placed.append(row)
    # static obstacles (crowd_sim.py:194-370)
    res = float(cfg["map_resolution"])
    n_circ = _rint(rnd() * cfg["num_circles"])
    n_wall = _rint(rnd() * cfg["num_walls"])
    G = _rint(float(cfg["map_size_m"]) / res)
    fmap = np.ones((G, G))
    half = _rint(G) / 2.0
    infl = 1 if phase == "test" else 1.25
    reach = r_radius + dd
    obst, verts = [], []                                # obst: (cell x, cell y, dim x, dim y)
    for _ in range(n_circ):
        while True:
            lx = rs.randint(-half, half); ly = rs.randint(-half, half)
            rad = (rnd() + 0.5) * 0.7
            xm, ym = lx * res, ly * res
            if not (_norm2(xm - robot[0], ym - robot[1]) < rad + reach or _norm2(xm - robot[2], ym - robot[3]) < rad + reach):
                break
        d = _rint(2 * rad / res)
        obst.append((_rint(lx + G / 2.0), _rint(ly + G / 2.0), d, d))
        ri = infl * rad
        verts.append([(xm + ri, ym + ri), (xm - ri, ym + ri), (xm - ri, ym - ri), (xm + ri, ym - ri)])
    for _ in range(n_wall):
        while True:
            lx = rs.randint(-half, half); ly = rs.randint(-half, half)
            if rnd() > 0.5:
                xd, yd = rs.randint(2, 4), 1
            else:
                yd, xd = rs.randint(2, 4), 1
            xm, ym = lx * res, ly * res
            near = lambda qx, qy: abs(xm - qx) < xd / 2.0 + reach and abs(ym - qy) < yd / 2.0 + reach
            if not (near(robot[0], robot[1]) or near(robot[2], robot[3])):
                break
        obst.append((_rint(lx + G / 2.0), _rint(ly + G / 2.0), _rint(xd / res), _rint(yd / res)))
        xi, yi = infl * xd, infl * yd
        verts.append([(xm + xi / 2.0, ym + yi / 2.0), (xm - xi / 2.0, ym + yi / 2.0), (xm - xi / 2.0, ym - yi / 2.0),
                      (xm + xi / 2.0, ym - yi / 2.0)])
    for ox, oy, dx, dy in obst:                         # paint: whole patch when it lies inside, else cell by cell
        if ox > dx / 2.0 and ox < G - dx / 2.0 and oy > dy / 2.0 and oy < G - dy / 2.0:
            sx, sy = _rint(ox - dx / 2.0), _rint(oy - dy / 2.0)
            fmap[sx:sx + dx, sy:sy + dy] = 0
        else:
            for ix in range(dx):
                for iy in range(dy):
                    cx, cy = _rint(ox + (ix - dx / 2.0)), _rint(oy + (iy - dy / 2.0))
                    if 0 < cx < G and 0 < cy < G:       # the reference's strict bounds: row / column 0 is never painted here
                        fmap[cx, cy] = 0
    static = []                                         # crowd_sim.py:372-405: discs along every obstacle
    for (ox, oy, dx, dy), v in zip(obst, verts):
        if dx == dy:
            px = (v[0][0] + v[2][0]) / 2.0; py = (v[0][1] + v[2][1]) / 2.0
            static.append((px, py, 0, 0, (v[0][0] - px) * np.sqrt(2)))
        elif dx > dy:
            py = (v[0][1] + v[2][1]) / 2.0
            rad = (v[0][1] - py) * np.sqrt(2)
            px = v[1][0] + rad
            while px < v[0][0]:
                static.append((px, py, 0, 0, rad)); px = px + 2 * rad
        else:
            px = (v[0][0] + v[2][0]) / 2.0
            rad = (v[0][0] - px) * np.sqrt(2)
            py = v[2][1] + rad
            while py < v[0][1]:
                static.append((px, py, 0, 0, rad)); py = py + 2 * rad
    return dict(robot=robot, humans=np.array(humans, dtype=np.float64).reshape(-1, 10),
                verts=np.array(verts, dtype=np.float64).reshape(-1, 4, 2), free_map=fmap.astype(np.uint8),
                static=np.array(static, dtype=np.float64).reshape(-1, 5), circle_radius=float(cr))


class CrowdSimStepper(object):
    """State: humans [E,H,9] and robot [E,9] = px, py, vx, vy, radius, v_pref, gx, gy, theta (float64 CUDA tensors);
    obstacle polygons verts [E,O,4,2] (counter-clockwise, crowd_sim.py:250-258) with n_obst [E]; free_map [E,G,G]
    uint8 (1 = free, indexed [x][y] like CrowdSim.map); global_time [E]."""

    def __init__(self, humans, robot, verts, n_obst, free_map, params, orca_params=None, map_params=None,
                 safety_space=0.0, robot_visible=True, use_grid_map=False, n_humans=None, sees_robot=None):
        """n_humans [E] (optional): env e has the humans 0 .. n_humans[e]-1, the other rows are padding that is neither
        seen nor moved (CrowdSim.reset draws round(human_num * (0.5 + u)) humans per episode, crowd_sim.py:668-671);
        sees_robot [E,H] (optional): Human.robot_visible, drawn per human at reset (crowd_sim.py:489-493, 539-543)."""
        import torch
        self.torch = torch
        self.h = humans.to(torch.float64).contiguous().clone()
        self.r = robot.to(torch.float64).contiguous().clone()
        self.verts = verts.to(torch.float64).contiguous()
        self.n_obst = n_obst.to(torch.int32).contiguous()
        self.free_map = free_map.to(torch.uint8).contiguous()
        self.params = dict(params)                      # navsim_crowd_params (time_step, penalties, map size, time limit)
        self.orca = dict(orca_params or dict(neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10))
        self.orca["time_step"] = self.params["time_step"]
        self.map_params = map_params
        self.safety_space = float(safety_space)
        self.robot_visible = bool(robot_visible)
        self.use_grid_map = bool(use_grid_map)
        E, H = self.h.shape[0], self.h.shape[1]
        self.n_humans = None if n_humans is None else n_humans.to(device=self.h.device, dtype=torch.int64).contiguous()
        self.sees_robot = None if sees_robot is None else sees_robot.to(device=self.h.device, dtype=torch.bool).contiguous()
        self.global_time = torch.zeros(E, dtype=torch.float64, device=self.h.device)
        # query (e, h) lists pedestrian h first, then the other pedestrians in index order, then the robot
        idx = torch.arange(H, device=self.h.device)
        others = torch.stack([torch.cat([idx[:k], idx[k + 1:]]) for k in range(H)]) if H > 1 else idx.new_zeros((H, 0))
        self.order = torch.cat([idx[:, None], others], dim=1)                       # [H, H]

    def human_queries(self):
        """agents [E*H, A, 6] and pref_vel [E*H, 2] exactly as ORCA.predict hands them to rvo2 (orca.py:101-124)."""
        torch = self.torch
        E, H = self.h.shape[0], self.h.shape[1]
        hs = self.h[:, self.order]                                                  # [E, H(query), H(agent), 9]
        ag = torch.stack([hs[..., 0], hs[..., 1], hs[..., 2], hs[..., 3],
                          hs[..., 4] + 0.01 + self.safety_space,                    # orca.py:103, 111
                          hs[:, :, :1, 5].expand(E, H, H)], dim=-1)                 # every agent gets agent 0's v_pref
        if self.robot_visible:
            rb = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 2], self.r[:, 3],
                              self.r[:, 4] + 0.01 + self.safety_space, self.r[:, 5]], dim=-1)
            rb = rb[:, None, None, :].expand(E, H, 1, 6).clone()
            rb[..., 5] = ag[:, :, :1, 5]
            ag = torch.cat([ag, rb], dim=2)
            if self.n_humans is not None:                # the robot follows the LIVE humans: slot n_humans[e] of every list
                ag[torch.arange(E, device=ag.device), :, self.n_humans] = rb[:, :, 0]
        vel = self.h[..., 6:8] - self.h[..., 0:2]                                   # orca.py:116-120
        speed = torch.sqrt(vel[..., :1] * vel[..., :1] + vel[..., 1:] * vel[..., 1:])     # separate ops: no fused multiply-add
        pref = torch.where(speed > 1, vel / speed, vel)
        return ag.reshape(E * H, ag.shape[2], 6).contiguous(), pref.reshape(E * H, 2).contiguous()

    def query_counts(self):
        """agents per query [E*H] (None: every list is full) and the live mask [E,H]."""
        torch = self.torch
        E, H = self.h.shape[0], self.h.shape[1]
        if self.n_humans is None and self.sees_robot is None:
            return None, None
        n = self.n_humans if self.n_humans is not None else torch.full((E,), H, dtype=torch.int64, device=self.h.device)
        live = torch.arange(H, device=self.h.device)[None, :] < n[:, None]
        sees = self.sees_robot if self.sees_robot is not None else torch.ones_like(live)
        cnt = n[:, None] + (sees & self.robot_visible).to(torch.int64)
        return torch.where(live, cnt, torch.ones_like(cnt)).reshape(-1).to(torch.int32), live

    def step(self, action, compute_local_map=True):
        """action [E,2] = ActionRot (v, r) of the robot.  -> dict(reward, done, info, min_dist, local_map)."""
        torch = self.torch
        E, H = self.h.shape[0], self.h.shape[1]
        dt = float(self.params["time_step"])
        action = action.to(device=self.h.device, dtype=torch.float64).reshape(E, 2)
        ag, pref = self.human_queries()
        obst_set = torch.arange(E, device=self.h.device, dtype=torch.int32).repeat_interleave(H)
        n_ag, live = self.query_counts()
        _, h_act = sim.crowd_orca(self.orca, ag, pref, self.verts, n_ag, self.n_obst, obst_set,
                                  self.h[..., 8].reshape(-1))
        rpose = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 8]], dim=1)
        npose, nvel = sim.crowd_agent_step(rpose, action, dt)                       # compute_position / compute_velocity
        robot10 = torch.stack([self.r[:, 0], self.r[:, 1], npose[:, 0], npose[:, 1], nvel[:, 0], nvel[:, 1],
                               self.r[:, 6], self.r[:, 7], self.r[:, 4], action[:, 1]], dim=1)
        agents5 = self.h[..., 0:5].contiguous()
        reward, done, info, min_dist = sim.crowd_check(self.params, self.free_map, robot10, agents5, self.global_time,
                                                       None if self.n_humans is None else self.n_humans)
        # crowd_sim.py:951-958: update all agents
        self.r[:, 0:2] = npose[:, 0:2]; self.r[:, 2:4] = nvel; self.r[:, 8] = npose[:, 2]
        hpose = torch.stack([self.h[..., 0], self.h[..., 1], self.h[..., 8]], dim=-1).reshape(E * H, 3)
        hp, hv = sim.crowd_agent_step(hpose, h_act, dt)
        if live is not None:                             # padding rows stay where they are
            keep = ~live.reshape(-1, 1)
            hp = torch.where(keep, hpose, hp); hv = torch.where(keep, self.h[..., 2:4].reshape(E * H, 2), hv)
        self.h[..., 0:2] = hp[:, 0:2].reshape(E, H, 2); self.h[..., 2:4] = hv.reshape(E, H, 2)
        self.h[..., 8] = hp[:, 2].reshape(E, H)
        self.global_time += dt
        local_map = None
        if compute_local_map and self.map_params is not None:
            rb4 = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 8], self.r[:, 4]], dim=1)
            local_map = (sim.crowd_local_map(self.map_params, self.free_map, rb4) if self.use_grid_map
                         else sim.crowd_angular_map(self.map_params, rb4, self.verts, self.n_obst))
        return dict(reward=reward, done=done, info=info, min_dist=min_dist, local_map=local_map,
                    human_actions=h_act.reshape(E, H, 2))


CROWD_INFO = ("Nothing", "Timeout", "ReachGoal", "Collision", "CollisionOtherAgent", "Danger")   # navsim.h NAVSIM_CROWD_*; info.py


class CrowdSimEnv(object):
    """'CrowdSim-v0' (crowd_sim/__init__.py:3-6 -> crowd_sim/envs/crowd_sim.py CrowdSim) for num_envs episodes at once.

    reset(phase, test_case) draws every episode's world on the host exactly as the reference does (crowd_reset_scenario:
    phases 'val' / 'test' reproduce the reference's own numbered cases), uploads it, and returns (ob, local_map) -- with a
    robot policy named 'ORCA' (ob, obstacle_vertices, local_map), crowd_sim.py:717-722.  step(action) is CrowdSimStepper
    (ORCA pedestrians, collision / goal / reward block, agent update, the robot's local map on the device) and returns
    (ob, local_map, reward, done, info) like crowd_sim.py:996.  Batched forms of the reference's lists:
      ob         dict(humans [E,H,5] px py vx vy radius, n_humans [E]; static [E,K,5], n_static [E] -- the static obstacles
                 as pedestrians, appended by the reference unless the robot's policy is SDOADRL or ORCA, crowd_sim.py:613-614)
      info       int32 [E] codes (CROWD_INFO), info_min_dist [E] for Danger
      action     [E,2] ActionRot (v, r) of the robot (the learning policies that would choose it are out of scope)
    Configuration: keyword arguments named like the reference's config file entries (CROWD_DEFAULTS).  Finished episodes are
    NOT restarted by step(): like the reference, the caller resets."""
    metadata = {"render.modes": ["human"]}

    def __init__(self, num_envs=1, device="cuda:0", seed=0, robot_policy_name="SARL", safety_space=0.0, **config):
        unknown = set(config) - set(CROWD_DEFAULTS)
        if unknown:
            raise TypeError("unknown CrowdSim configuration entries: %s" % sorted(unknown))
        self.cfg = dict(CROWD_DEFAULTS); self.cfg.update(config)
        self.num_envs, self.device, self.seed = int(num_envs), device, int(seed)
        self.robot_policy_name = robot_policy_name
        self.safety_space = float(safety_space)
        self.case_size = dict(train=CROWD_CASE_CAPACITY["train"], val=int(self.cfg["val_size"]), test=int(self.cfg["test_size"]))
        self.case_counter = dict(train=0, test=0, val=0)
        self.phase, self.stepper, self.scenarios, self._resets = None, None, None, 0

    # -- the parameter blocks of the device entry points, from the configuration
    def _params(self):
        c = self.cfg
        return dict(time_step=c["time_step"], discomfort_dist=c["discomfort_dist"], map_size_m=c["map_size_m"],
                    map_resolution=c["map_resolution"], success_reward=c["success_reward"],
                    collision_penalty=c["collision_penalty"], discomfort_penalty_factor=c["discomfort_penalty_factor"],
                    rotation_penalty_factor=c["rotation_penalty_factor"], timeout_penalty=c["timeout_penalty"],
                    time_limit=c["time_limit"])

    def _map_params(self):
        c = self.cfg
        return dict(map_size_m=c["map_size_m"], map_resolution=c["map_resolution"], submap_size_m=c["submap_size_m"],
                    angular_max_range=c["angular_map_max_range"], angular_dim=c["angular_map_dim"], normalize=1,
                    angular_min=c["angle_min"] * np.pi, angular_max=c["angle_max"] * np.pi)      # crowd_sim.py:140-145

    def reset(self, phase="test", test_case=None, compute_local_map=True):
        import torch
        assert phase in ("train", "val", "test")
        self.phase = phase
        if test_case is not None:
            self.case_counter[phase] = int(test_case)
        E = self.num_envs
        cases = [(self.case_counter[phase] + e) % self.case_size[phase] for e in range(E)]
        self.case_counter[phase] = (self.case_counter[phase] + E) % self.case_size[phase]       # crowd_sim.py:685-687, E at a time
        self.scenarios = [crowd_reset_scenario(self.cfg, phase, cases[e], seed=(self.seed + 7919 * self._resets + e) & 0x7FFFFFFF)
                          for e in range(E)]
        self._resets += 1
        sc = self.scenarios
        H = max(1, max(len(s["humans"]) for s in sc)); O = max(1, max(len(s["verts"]) for s in sc))
        K = max(1, max(len(s["static"]) for s in sc)); G = sc[0]["free_map"].shape[0]
        humans = np.zeros((E, H, 9)); sees = np.zeros((E, H), bool); nh = np.zeros(E, np.int32)
        robot = np.zeros((E, 9)); verts = np.zeros((E, O, 4, 2)); no = np.zeros(E, np.int32)
        static = np.zeros((E, K, 5)); ns = np.zeros(E, np.int32); fmap = np.ones((E, G, G), np.uint8)
        humans[..., 0] = 1e6; humans[..., 1] = 1e6 + 100.0 * np.arange(H)[None, :]     # padding rows: far away, apart
        for e, s in enumerate(sc):
            n = len(s["humans"]); nh[e] = n
            h = s["humans"]                        # px py gx gy vx vy theta radius v_pref sees
            humans[e, :n] = np.stack([h[:, 0], h[:, 1], h[:, 4], h[:, 5], h[:, 7], h[:, 8], h[:, 2], h[:, 3], h[:, 6]], axis=1)
            sees[e, :n] = h[:, 9] > 0.5
            r = s["robot"]
            robot[e] = [r[0], r[1], r[4], r[5], r[7], r[8], r[2], r[3], r[6]]
            m = len(s["verts"]); no[e] = m; verts[e, :m] = s["verts"]
            k = len(s["static"]); ns[e] = k; static[e, :k] = s["static"]
            fmap[e] = s["free_map"]
        t = lambda a, dt=None: torch.as_tensor(a, device=self.device) if dt is None else torch.as_tensor(a, device=self.device).to(dt)
        self.static, self.n_static = t(static), t(ns)
        self.stepper = CrowdSimStepper(t(humans), t(robot), t(verts), t(no), t(fmap), self._params(),
                                       map_params=self._map_params(), safety_space=self.safety_space,
                                       robot_visible=bool(self.cfg["robot_visible"]), use_grid_map=bool(self.cfg["use_grid_map"]),
                                       n_humans=t(nh), sees_robot=t(sees))
        local_map = self._local_map() if compute_local_map else None
        if self.robot_policy_name == "ORCA":
            return self._ob(), self.stepper.verts, local_map
        return self._ob(), local_map

    def _ob(self):
        st = self.stepper
        import torch
        ob = dict(humans=st.h[..., 0:5].clone(), n_humans=st.n_humans.to(torch.int32))
        if self.robot_policy_name not in ("SDOADRL", "ORCA"):
            ob["static"], ob["n_static"] = self.static, self.n_static
        return ob

    def _local_map(self):
        import torch
        st = self.stepper
        rb4 = torch.stack([st.r[:, 0], st.r[:, 1], st.r[:, 8], st.r[:, 4]], dim=1)
        return (sim.crowd_local_map(st.map_params, st.free_map, rb4) if st.use_grid_map
                else sim.crowd_angular_map(st.map_params, rb4, st.verts, st.n_obst))

    def step(self, action, compute_local_map=True):
        if self.stepper is None:
            raise RuntimeError("reset() first")
        out = self.stepper.step(action, compute_local_map=compute_local_map)
        self.info_min_dist = out["min_dist"]
        return self._ob(), out["local_map"], out["reward"], out["done"].bool(), out["info"]

    def close(self):
        self.stepper = None
