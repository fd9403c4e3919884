"""NavGymEnv: the reference's gym.Env API (nav_gym_env/env.py:30-831) over E batched arenas.

Same constructor keywords as the registered NavGym-v0 (nav_gym_env/__init__.py:6-38), same methods
(reset, step, compute_reward(s), compute_terminals, compute_done, compute_info,
_override_reward_factor), same observation dict keys and layout.  Build additions are keyword-only
and default to the reference's behaviour for num_envs == 1:

    num_envs        arenas stepped per call (leading axis of every returned array)
    n_beams, lidar  lidar geometry (default: KetiRobot's 512 beams over 2*pi)
    map_size        cells per side of every arena (default 400, the reference's outdoor size)
    randomize_maps  True: an arena that finishes an episode restarts on a NEW random outdoor map,
                    generated on the device inside step() (navsim_regen); False: it respawns in place
    plan_paths      True (default): pedestrians follow the waypoints of planned shortest paths and re-plan
                    at their goal (env.py:667-680, 788-804; navsim_replan after every step); with
                    randomize_maps new starts / goals are sampled on the costmap and kept only when the
                    planner joins them (env.py:342-383, 756-762).  False: straight-line goals.
                    Maps above 1000 cells per side fall back to False (the search lives in LDS).
    pedestrian_model 'policy' (the reference's HumanPolicy actor on the device, env.py:617-662; needs
                    policy_weights = the state_dict of human_policy.pth, a path to it, or a dict of arrays),
                    'sfm' (build-defined social force), 'external' (caller supplies (v, w) per
                    pedestrian -- the slot the reference fills with HumanPolicy) or 'none'
    action_kind     'twist' (default): action = (v, omega) like the reference (env.py:591); 'wheels': action = the
                    angular speeds (left, right) of a skid-steer base's wheel pairs in rad/s, converted on the device
                    with the robot's wheel radius / track (robots.py; Husky: husky.urdf.xacro:61-67)
    clip_actions    True: the twist is clamped to linvel_range x rotvel_range on the device (the reference only prints
                    a warning and never clips, env.py:606-613: default False)
    regen_min_steps, pregen_pipeline   (randomize_maps) pregen_pipeline = P > 0 (default: 4; for worlds of corridor maps with
                    planned starts the passes alternate between two side streams; 0 where the pipeline is not available): the next world of every arena is staged ahead
                    of time on a side stream (a pass every P steps) and installed inside the step's own launch.  With
                    regen_min_steps = 0 (default) the rollout is EXACTLY the one without the pipeline: an arena that finishes
                    before its world is staged is generated on the spot (counters()['regen_late']).  regen_min_steps >= 4 P
                    drops that fallback's launches (fastest): an episode shorter than regen_min_steps restarts on its old map,
                    which the reference never does (it draws a map at every reset) -- counted in counters()['regen_short'].
                    The on-the-spot fallback serves at most max(8, num_envs // 128) arenas per step: more late arenas than
                    that in ONE step keep their old map for the next episode and are counted in counters()['regen_unserved']
                    (not seen in the soaks of profiles/r05_pipe/: a pass every P steps leaves a handful of late arenas per step)
    pregen_fallback_poll   (pregen_pipeline, regen_min_steps < 4 P) True: step() reads one byte back from the device -- "did this step
                    leave an arena without its staged world?" -- and launches the on-the-spot generation only then, instead of
                    enqueueing its launches blind behind every step.  A host wait per step; default (None): on for worlds of
                    corridor maps or planned starts, whose fallback is ~18 launches (the reference's configuration: 2.74 -> 3.25 M env-steps/s at 1024 arenas, 3.74 -> 4.07 M at 4096),
                    off otherwise.  Same rollout either way.
    use_graphs      replay a step's launches (navsim_step, navsim_regen, navsim_replan) as one captured hipGraph; None
                    (default) = when randomize_maps makes a step several launches (c5: +6 %) and navsim_regen does not fork
                    (corridor maps with planned starts: plain launches are as fast or faster); results are identical
    max_waypoints   waypoints kept per pedestrian route (default 64 = 128 m at the 2 m interval; the reference keeps
                    all of them, env.py:788-804); longer routes are stored cut, counted, and continued to the same goal
    device, seed, env_index_base (global index of arena 0: sharding), auto_reset, field_format
    autoreset_mode  (auto_reset) 'same_step' (default): the step() that ends an arena's episode also restarts it -- the returned
                    observation row is the FIRST of the new episode, and the observation the reference's step() returns with
                    done = True (env.py:700-728: the last one of the episode, after a crash the re-scan at the reverted pose)
                    is info['final_observation'] (same keys as the observation dict; rows where info['final_mask'] = done);
                    'next_step' (gymnasium's next-step mode): step() returns that terminal observation itself, like the
                    reference, and the arena is reset by the NEXT step() -- its action is ignored, reward 0, done False, the
                    observation is the new episode's first one, info['reset_mask'] marks it.  final_observation=False drops
                    the terminal rows of 'same_step' (and their second scan after a crash).
    reset(mask)     reset() of SOME arenas (the reference's reset() is per environment, env.py:730-831): a bool / 0-1 array
                    [num_envs]; the others keep their state and their rows.  Not with the pipelined reset path.

`env.counters()` reports what the caps of the device-side reset path left unserved (arenas beyond regen_cap,
pedestrians beyond replan_cap, routes cut at max_waypoints) since the last call.

num_envs == 1 returns NumPy float64 arrays with the reference's shapes; num_envs > 1 returns torch
tensors on `device` (float32 observations).  Everything on the step() path runs in the HIP library;
there is no CPU fallback.

Lifetime of what step() returns (num_envs > 1): the observation dict, reward, done and info are views of device buffers
the simulator keeps in PAIRS -- they stay intact through the next step() and are overwritten by the one after
(`obs = next_obs`, `dones.append(done)` across one step are safe; keep a `.clone()` of anything needed longer).

Pickling (env.py:30, 56-78: the reference is an EzPickle): `pickle.dumps(env)` stores the constructor's keyword
arguments; the copy is a fresh environment that has not been reset().  `state_dict()` / `load_state_dict()` move the
simulation state itself.  When `gym` is importable the class is a gym.Env (`unwrapped`, `spec`, `reward_range`).
"""
import warnings

import numpy as np

from . import abi, robots, world
from .registry import HAVE_GYM, spaces

if HAVE_GYM:                                # the reference's base class (env.py:30), when the package is there
    import gym as _gym
    _EnvBase = _gym.Env
else:
    class _EnvBase(object):                 # what wrappers read of a gym.Env
        reward_range = (-float("inf"), float("inf"))
        spec = None

        @property
        def unwrapped(self):
            return self

        def seed(self, seed=None):
            return [seed]

        def __enter__(self):
            return self

        def __exit__(self, *args):
            self.close()
            return False

DEFAULT_KWARGS = {                       # nav_gym_env/__init__.py:6-38
    'robot_type': 'keti',
    'time_step': 0.2,
    'min_turning_radius': 0,
    'distance_threshold': 0.5,
    'num_scan_stack': 1,
    'linvel_range': [0, 0.5],
    'rotvel_range': [-0.64, 0.64],
    'human_v_pref_range': [0., 0.6],
    'human_has_legs_ratio': 0.5,
    'indoor_ratio': 0.5,
    'min_goal_dist': 10,
    'max_goal_dist': 20,
    'reward_scale': 15.,
    'reward_success_factor': 1,
    'reward_crash_factor': 1,
    'reward_progress_factor': 0.001,
    'reward_forward_factor': 0.0,
    'reward_rotation_factor': 0.005,
    'reward_discomfort_factor': 0.01,
    'env_param_range': dict(
        num_humans=([5, 15], 'int'),
        corridor_width=([3, 4], 'int'),
        iterations=([80, 150], 'int'),
        obstacle_number=([10, 10], 'int'),
        obstacle_width=([0.3, 1.0], 'float'),
        scan_noise_std=([0., 0.05], 'float'),
    ),
}


class _AgentView(object):
    """Read-only view of one agent of arena 0 with the attribute names RosEnv reads
    (ros_env.py:83-176): px, py, theta, vx, vy, footprint, ..."""

    def __init__(self, env, kind, index=0):
        self._env, self._kind, self._i = env, kind, index
        spec = robots.ROBOTS[env.robot_type] if kind == "robot" else robots.HUMAN
        for k, v in spec.items():
            setattr(self, k, v)

    def _pose(self):
        t = self._env.sim.t
        p = t["robot_pose"][0] if self._kind == "robot" else t["ped_pose"][0, self._i]
        return p.cpu().numpy()

    px = property(lambda s: float(s._pose()[0]))
    py = property(lambda s: float(s._pose()[1]))
    theta = property(lambda s: float(s._pose()[2]))

    @property
    def vx(self):
        return float(self._env.sim.t["ped_vel"][0, self._i, 0]) if self._kind != "robot" else 0.0

    @property
    def vy(self):
        return float(self._env.sim.t["ped_vel"][0, self._i, 1]) if self._kind != "robot" else 0.0

    @property
    def gx(self):
        return float(self._env.sim.t["robot_goal"][0, 0])

    @property
    def gy(self):
        return float(self._env.sim.t["robot_goal"][0, 1])


class NavGymEnv(_EnvBase):
    metadata = {"render.modes": ["human", "rgb_array"]}
    _warned = set()                         # fallbacks are announced once per process

    @classmethod
    def _warn_once(cls, key, text):
        if key not in cls._warned:
            cls._warned.add(key)
            warnings.warn(text, RuntimeWarning, stacklevel=3)

    def __init__(self, robot_type, time_step, min_turning_radius, distance_threshold, num_scan_stack,
                 linvel_range, rotvel_range, human_v_pref_range, human_has_legs_ratio, indoor_ratio,
                 min_goal_dist, max_goal_dist, reward_scale, reward_success_factor, reward_crash_factor,
                 reward_progress_factor, reward_forward_factor, reward_rotation_factor,
                 reward_discomfort_factor, env_param_range, *,
                 num_envs=1, n_beams=None, lidar=None, map_size=400, pedestrian_model="sfm", policy_weights=None,
                 num_humans=None, device="cuda:0", seed=0, env_index_base=0, auto_reset=None,
                 field_format=abi.FIELD_U16T, n_spawn=None, randomize_maps=False, plan_paths=True,
                 action_kind="twist", clip_actions=False, max_waypoints=64, march_rule=None, use_graphs=None,
                 regen_min_steps=0, pregen_pipeline=None, pregen_stage_cap=None, autoreset_mode="same_step",
                 final_observation=True, pregen_fallback_poll=None):
        from . import lib
        if robot_type not in robots.ROBOTS:
            raise NotImplementedError(robot_type)            # env.py:772-773
        # EzPickle (env.py:56-78): what the environment was made with is what a pickle of it carries
        self._ctor_kwargs = dict(
            robot_type=robot_type, time_step=time_step, min_turning_radius=min_turning_radius,
            distance_threshold=distance_threshold, num_scan_stack=num_scan_stack, linvel_range=linvel_range,
            rotvel_range=rotvel_range, human_v_pref_range=human_v_pref_range, human_has_legs_ratio=human_has_legs_ratio,
            indoor_ratio=indoor_ratio, min_goal_dist=min_goal_dist, max_goal_dist=max_goal_dist, reward_scale=reward_scale,
            reward_success_factor=reward_success_factor, reward_crash_factor=reward_crash_factor,
            reward_progress_factor=reward_progress_factor, reward_forward_factor=reward_forward_factor,
            reward_rotation_factor=reward_rotation_factor, reward_discomfort_factor=reward_discomfort_factor,
            env_param_range=env_param_range, num_envs=num_envs, n_beams=n_beams, lidar=lidar, map_size=map_size,
            pedestrian_model=pedestrian_model, policy_weights=policy_weights, num_humans=num_humans, device=device, seed=seed,
            env_index_base=env_index_base, auto_reset=auto_reset, field_format=field_format, n_spawn=n_spawn,
            randomize_maps=randomize_maps, plan_paths=plan_paths, action_kind=action_kind, clip_actions=clip_actions,
            max_waypoints=max_waypoints, march_rule=march_rule, use_graphs=use_graphs,
            regen_min_steps=regen_min_steps, pregen_pipeline=pregen_pipeline, pregen_stage_cap=pregen_stage_cap,
            autoreset_mode=autoreset_mode, final_observation=final_observation, pregen_fallback_poll=pregen_fallback_poll)
        self.robot_type = robot_type
        self.time_step = time_step
        self.min_turning_radius = min_turning_radius
        self.distance_threshold = distance_threshold
        self.num_scan_stack = num_scan_stack
        self.linvel_range = linvel_range
        self.rotvel_range = rotvel_range
        self.human_v_pref_range = human_v_pref_range
        self.human_has_legs_ratio = human_has_legs_ratio
        self.indoor_ratio = indoor_ratio
        self.min_goal_dist = min_goal_dist
        self.max_goal_dist = max_goal_dist
        self.env_param_range = env_param_range
        self.num_envs = int(num_envs)
        # map_size="reference": the reference's own two sizes (map_generator.py:108-142) -- every arena is allocated at
        # 1000 x 1000 cells, a corridor episode fills it, an outdoor episode draws its 400 x 400 map in the corner
        # [0, 400)^2 (the rest is occupied: nothing behind the border wall is ever seen).  An integer: one size for both.
        self.outdoor_map_size = 0
        if map_size == "reference":
            map_size, self.outdoor_map_size = 1000, 400
        self.map_size = int(map_size)
        self.device = device
        self.seed_value = int(seed)
        self.pedestrian_model = pedestrian_model
        self.auto_reset = (self.num_envs > 1) if auto_reset is None else bool(auto_reset)
        if autoreset_mode not in ("same_step", "next_step"):
            raise ValueError("autoreset_mode must be 'same_step' or 'next_step'")
        self.autoreset_mode = autoreset_mode
        self.final_observation = bool(final_observation) and self.auto_reset and autoreset_mode == "same_step"
        self._num_humans_fixed = num_humans
        self._episode_batch = 0
        self.randomize_maps = bool(randomize_maps)
        # start / goal pairs kept per arena: what an arena restarts from IN PLACE.  A world that draws a new map per episode
        # only does that for arenas beyond cfg.regen_cap in one step -- 4 pairs there (each is planned at every reset:
        # env.py:342-383), 16 where the map stays
        if n_spawn is None:
            n_spawn = 4 if self.randomize_maps else 16
        self.replan_cap = 1024                              # pedestrians re-planned per step, at most
        self._use_graphs_arg = use_graphs
        self.use_graphs = bool(randomize_maps) if use_graphs is None else bool(use_graphs)
        # pregen_pipeline = P > 0 (with randomize_maps): the next world of every arena is generated ahead of time by staging
        # passes on a side stream, one every P steps, and a finished arena takes it inside the step's own launch
        # (navsim_step_install) -- navsim_regen leaves the step's critical path.  It rests on regen_min_steps >= 4 P: an
        # episode that ended after fewer steps restarts on its OLD map (the reference draws a map at every reset: opt-in).
        # None (default): the pipelined reset path wherever it is available (packed field, i.e. map_size <= 1024; not with
        # pedestrian_model='policy') -- a pass every 4 steps (a c5-shaped world through this API: 5.4 -> 7.4 M env-steps/s, with planned
        # routes 1.3 -> 3.0 M, profiles/_diag/gym_c5_steps.py); for worlds whose reset is heavy (corridor maps with planned starts, the
        # reference's own kind) the passes alternate between two side streams, each of which starts one every 8 steps
        # (profiles/r05_refdef/pipeline_sweep.txt, profiles/r06_planner/ab_stage_lanes.txt).  With regen_min_steps = 0 it changes no result: the rollout is the one of
        # pregen_pipeline=0, bit for bit.
        self._pregen_auto = pregen_pipeline is None
        if pregen_pipeline is None:
            available = (pedestrian_model != "policy" and field_format == abi.FIELD_U16T and
                         (map_size == "reference" or int(map_size) <= 1024))
            heavy = bool(plan_paths) and float(indoor_ratio) > 0.0 and (map_size == "reference" or int(map_size) <= 1000)
            # (worlds of corridor maps with planned starts: a pass every 4 steps too since round 6 -- their passes alternate between
            #  two side streams, NavSim.enable_pregen stage_lanes, i.e. each stream still starts one every 8 steps)
            pregen_pipeline = 4 if available else 0
            if use_graphs:                         # (asked for: the graph replay of step + navsim_regen is the other form)
                pregen_pipeline = 0
        self.pregen_pipeline = int(pregen_pipeline) if (self.randomize_maps and self.auto_reset) else 0
        self.regen_min_steps = int(regen_min_steps)
        self.pregen_stage_cap = pregen_stage_cap        # arenas one staging pass serves at most (None: NavSim.enable_pregen's default)
        self.pregen_fallback_poll = pregen_fallback_poll   # None: NavSim.enable_pregen's default (on for corridor maps / planned starts)
        if self.pregen_pipeline:
            # regen_min_steps >= 4 P: the rule (fastest; short episodes keep their map).  Below that -- 0 is the reference's own
            # "a new map at every reset()" -- an arena that finishes before its world is staged is generated on the spot by
            # navsim_regen (NavSim.enable_pregen fallback): same rollout as without the pipeline, bit for bit
            if pedestrian_model == "policy":
                raise ValueError("pregen_pipeline is not available with pedestrian_model='policy'")
            self.use_graphs = False
        self._graphed = False
        self._overlap_replan = False
        self.plan_paths = bool(plan_paths) and int(map_size) <= 1000
        if self._use_graphs_arg is None and self.plan_paths and float(indoor_ratio) > 0.0:
            # corridor maps with planned starts: navsim_regen forks its distance transform beside the searches; as fork / join
            # nodes of a hipGraph that gains nothing (reference defaults, 1024 arenas: 0.92 M graphed, 0.96 M plain launches)
            self.use_graphs = False
        if plan_paths and not self.plan_paths:
            self._warn_once("plan_paths", "NavGymEnv: map_size %d > 1000: the planner's search lives in LDS (costmaps up to "
                            "200 x 200 cells), plan_paths falls back to False -- pedestrians head straight for their goals"
                            % int(map_size))
        if field_format == abi.FIELD_U16T and int(map_size) > 1024:
            field_format = abi.FIELD_F32     # beyond the rect records' and the packed regeneration's tested range
            self._warn_once("field_format", "NavGymEnv: map_size %d > 1024: the packed uint16 distance field and its rect "
                            "records are not used beyond 1024 cells per side, field_format falls back to FIELD_F32" % int(map_size))
        spec = robots.ROBOTS[robot_type]
        nh_hi = int(env_param_range["num_humans"][0][1]) if num_humans is None else int(num_humans)
        ped = {"none": abi.PED_NONE, "external": abi.PED_EXTERNAL, "sfm": abi.PED_SFM,
               "policy": abi.PED_EXTERNAL}[pedestrian_model]
        if pedestrian_model == "policy" and policy_weights is None:
            raise ValueError("pedestrian_model='policy' needs policy_weights (human_policy.pth is not shipped)")
        self._policy_weights = policy_weights
        if nh_hi == 0:
            ped = abi.PED_NONE
        cfg = lib.default_config(
            n_envs=self.num_envs, map_h=self.map_size, map_w=self.map_size, max_peds=max(nh_hi, 1),
            n_scan_stack=num_scan_stack, ped_model=ped, lidar_legs=1,
            auto_reset=(abi.AUTORESET_NONE if not self.auto_reset else
                        (abi.AUTORESET_NEXT_STEP if autoreset_mode == "next_step" else abi.AUTORESET_SAME_STEP)),
            n_spawn=n_spawn, add_scan_noise=1, env_index_base=env_index_base, field_format=field_format,
            time_step=time_step, axle_offset=spec["axle_offset"], min_turning_radius=float(min_turning_radius),
            distance_threshold=distance_threshold, range_max=spec["range_max"], seed=self.seed_value)
        # reset() -- of the whole batch and, with randomize_maps, of every finished arena -- runs on the device
        # (navsim_regen): planning on the costmap as env.py:342-383 when plan_paths, the map kind by indoor_ratio
        # (env.py:295), and the per-episode draws of env_param_range (env.py:281-292)
        if action_kind not in ("twist", "wheels"):
            raise ValueError("action_kind must be 'twist' or 'wheels'")
        self.action_kind = action_kind
        cfg.action_kind = abi.ACTION_WHEELS if action_kind == "wheels" else abi.ACTION_TWIST
        cfg.clamp_action = int(bool(clip_actions))
        cfg.linvel_lo, cfg.linvel_hi = float(linvel_range[0]), float(linvel_range[1])
        cfg.rotvel_lo, cfg.rotvel_hi = float(rotvel_range[0]), float(rotvel_range[1])
        cfg.wheel_radius = float(spec.get("wheel_radius", robots.HUSKY_WHEEL_RADIUS))
        cfg.wheel_track = float(spec.get("wheel_track", robots.HUSKY_TRACK))
        cfg.max_waypoints = int(max_waypoints)
        if march_rule is not None:                    # include/navsim.h NAVSIM_MARCH_*: the unpinned rounding of range_libc
            cfg.march_rule = int(march_rule)
        # with randomize_maps every step() is navsim_step + navsim_regen: the restarted arenas' first observations come from
        # regen's masked launch instead of a second scan inside the step (include/navsim.h defer_reset_scan), where a
        # launch is one generation of workgroups -- a few arenas per CU
        cfg.defer_reset_scan = int(self.randomize_maps and self.auto_reset and self.num_envs <= 1024)
        cfg.regen_min_steps = self.regen_min_steps if (self.randomize_maps and self.auto_reset) else 0
        if self.pregen_pipeline:
            cfg.defer_reset_scan = 0              # a staged world brings its first observation; restarts in place scan in the step
            cfg.regen_cap = self.num_envs         # every finished arena decides alone inside the step (navsim_step_install)
        cfg.regen_plan = int(self.plan_paths)
        cfg.regen_indoor_ratio = float(indoor_ratio)
        cfg.outdoor_map_size = int(self.outdoor_map_size)
        room = (self.outdoor_map_size or self.map_size) * cfg.resolution
        cfg.min_goal_dist = float(min(min_goal_dist, 0.4 * room))
        cfg.max_goal_dist = float(min(max_goal_dist, 0.8 * room))
        cfg.v_pref_lo, cfg.v_pref_hi = float(human_v_pref_range[0]), float(human_v_pref_range[1])
        cfg.has_legs_ratio = float(human_has_legs_ratio)
        epr = env_param_range
        cfg.obstacle_number, cfg.obstacle_number_hi = [int(x) for x in epr["obstacle_number"][0]]
        cfg.obstacle_width_lo, cfg.obstacle_width_hi = [float(x) for x in epr["obstacle_width"][0]]
        cfg.corridor_width_lo, cfg.corridor_width_hi = [int(x) for x in epr["corridor_width"][0]]
        cfg.iterations_lo, cfg.iterations_hi = [int(x) for x in epr["iterations"][0]]
        cfg.scan_noise_std_lo, cfg.scan_noise_std_hi = [float(x) for x in epr["scan_noise_std"][0]]
        if num_humans is None:
            cfg.num_humans_lo, cfg.num_humans_hi = [int(x) for x in epr["num_humans"][0]]
        else:
            cfg.num_humans_lo = cfg.num_humans_hi = int(num_humans)
        for i, v in enumerate(np.asarray(spec["threshold_footprint"], dtype=np.float64).reshape(-1)):
            cfg.robot_seen_footprint[i] = float(v)
        if lidar is not None:                         # (angle_min, angle_last, n_beams)
            cfg.angle_min, cfg.angle_last, cfg.n_beams = float(lidar[0]), float(lidar[1]), int(lidar[2])
        elif n_beams is not None and int(n_beams) == 1081:
            world.lidar_1081(cfg)
        elif n_beams is not None:
            world.lidar_full_circle(cfg, int(n_beams))
        else:                                         # keti_robot.py:44-48, env.py:388-390
            cfg.n_beams = spec["n_angles"]
            cfg.angle_min = spec["angle_min"]
            cfg.angle_last = spec["angle_max"] - spec["angle_increment"]
        self.cfg = cfg
        self._override_reward_factor(reward_scale, reward_success_factor, reward_crash_factor,
                                     reward_progress_factor, reward_forward_factor, reward_rotation_factor,
                                     reward_discomfort_factor)
        self.sim = None
        self.prev_obs = None
        self._bool = None
        self._views = {}                                    # per buffer parity: what step() hands out beside the observation
        self.robot = _AgentView(self, "robot")
        self.humans = []
        self._map_info = None
        self.scan_threshold = None
        self.scan_discomfort_threshold = None
        if action_kind == "wheels":                   # wheel speeds that reach every corner of the twist box
            corners = [robots.wheels_from_twist(v, w, cfg.wheel_radius, cfg.wheel_track)
                       for v in linvel_range for w in rotvel_range]
            lo, hi = float(np.min(corners)), float(np.max(corners))
            self.action_space = spaces.Box(low=np.array([lo, lo]), high=np.array([hi, hi]), dtype=np.float32)
        else:
            self.action_space = spaces.Box(low=np.array([linvel_range[0], rotvel_range[0]]),
                                           high=np.array([linvel_range[1], rotvel_range[1]]), dtype=np.float32)
        D = num_scan_stack * cfg.n_beams + 7
        self.observation_space = spaces.Dict({
            'observation': spaces.Box(-np.inf, np.inf, shape=(D,), dtype=np.float32),
            'achieved_goal': spaces.Box(-np.inf, np.inf, shape=(2,), dtype=np.float32),
            'desired_goal': spaces.Box(-np.inf, np.inf, shape=(2,), dtype=np.float32)})

    # ---- env.py:144-160 -------------------------------------------------------------------------
    def _override_reward_factor(self, reward_scale=15., reward_success_factor=1, reward_crash_factor=1,
                                reward_progress_factor=0.001, reward_forward_factor=0.0,
                                reward_rotation_factor=0.005, reward_discomfort_factor=0.01):
        self.reward_scale = reward_scale
        self.reward_success_factor = reward_success_factor
        self.reward_crash_factor = reward_crash_factor
        self.reward_progress_factor = reward_progress_factor
        self.reward_forward_factor = reward_forward_factor
        self.reward_rotation_factor = reward_rotation_factor
        self.reward_discomfort_factor = reward_discomfort_factor
        for k in ("scale", "success_factor", "crash_factor", "progress_factor", "forward_factor",
                  "rotation_factor", "discomfort_factor"):
            setattr(self.cfg, "reward_" + k, float(getattr(self, "reward_" + k)))
        if getattr(self, "sim", None) is not None:
            for k in ("scale", "success_factor", "crash_factor", "progress_factor", "forward_factor",
                      "rotation_factor", "discomfort_factor"):
                setattr(self.sim.cfg, "reward_" + k, float(getattr(self, "reward_" + k)))

    # ---- reset (env.py:730-831), all arenas ---------------------------------------------------------
    def reset(self, mask=None):
        """reset() of every arena (env.py:730-831).  Nothing is generated on the host: maps, distance fields,
        costmaps, start / goal pairs (joined by a planned path when plan_paths), pedestrians, the per-episode
        env_param draws and the first observations all come from navsim_regen with every arena marked
        finished (NavSim.regenerate_all).
        mask [num_envs] (bool / 0-1): reset only those arenas -- the reference's reset() is per environment -- the others
        keep their state and their observation rows (NavSim.reset_arenas: next start / goal pair and episode number, first
        observation; with randomize_maps a new world each)."""
        from . import lib
        lib.require_gpu()                                 # no CPU fallback: fail before any work
        import torch
        if mask is not None:
            if self.sim is None:
                raise RuntimeError("reset(mask) needs one reset() of every arena first")
            if self.pregen_pipeline:
                raise ValueError("reset(mask) is not available with the pipelined reset path (pregen_pipeline=0 for it)")
            m = torch.as_tensor(np.asarray(mask) if not hasattr(mask, "is_cuda") else mask).reshape(self.num_envs)
            if self._graphed:
                torch.cuda.synchronize(self.sim.device)
            self.sim.reset_arenas(m, new_world=self.randomize_maps)
            if "policy_prev_actions" in self.sim.t:       # env.py:739
                self.sim.t["policy_prev_actions"].mul_((m.to(self.sim.device) == 0).to(self.sim.t["policy_prev_actions"].dtype)[:, None, None])
            self._map_info = None
            self._humans_of_episode = None
            return self._obs_dict()
        self._bool = torch.bool
        from . import sim as simmod
        cfg = self.cfg
        first = self.sim is None
        if first:
            dev = torch.device(self.device)
            # rect records (the march's shortcut around most field reads): always for fixed maps; with a new map per episode
            # only in worlds of outdoor maps, whose records come out of the pass that writes the new field -- corridor maps
            # would need the verified builder for a handful of maps per step, which costs more than the records save
            with_rects = (cfg.field_format == abi.FIELD_U16T and self.map_size <= 1024 and
                          (not self.randomize_maps or not cfg.regen_indoor_ratio > 0.0))
            arrays = world.empty_world(cfg, device=self.device,
                                       plan_paths=self.plan_paths and cfg.ped_model != abi.PED_NONE,
                                       rect_table=with_rects)
            for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
                arrays[key] = simmod.scan_threshold(cfg, torch.from_numpy(robots.footprint_array(self.robot_type, name)).to(dev))
            # every map of this world comes from navsim_regen, whose generators close their maps with a border wall
            # (map_generator.py:11, 61-93): the LDS form of the march may be used (include/navsim.h closed_maps)
            cfg.closed_maps = int("rect_index" in arrays)
            self.sim = simmod.NavSim(cfg, arrays, device=self.device, final_obs=self.final_observation)
            self._views = {}
            self.scan_threshold = arrays["scan_threshold"]
            self.scan_discomfort_threshold = arrays["scan_discomfort"]
            if self.pedestrian_model == "policy":
                w = self._policy_weights
                if isinstance(w, str):
                    w = torch.load(w, map_location="cpu")
                self.sim.set_policy(w)
        elif "policy_prev_actions" in self.sim.t:
            self.sim.t["policy_prev_actions"].zero_()           # env.py:739
        self._episode_batch += 1
        self.sim.regenerate_all(new_episode=not first)
        # pedestrians on planned routes: navsim_replan overlaps the step (not with 'policy', whose control block runs in
        # front of every step and reads the routes)
        self._overlap_replan = ("costmap" in self.sim.t and self.sim.due is not None and self.pedestrian_model != "policy"
                                and not self.pregen_pipeline)
        if self.pregen_pipeline and first and self._pregen_auto:
            # the pipeline keeps a second, staged copy of every array navsim_regen writes (the maps above all) and builds it
            # through a few GB of scratch: where that does not fit, the env's own choice falls back to navsim_regen after every step
            need = 2 * sum(v.numel() * v.element_size() for k, v in self.sim.t.items() if k in self.sim.STAGED) + (6 << 30)
            free = torch.cuda.mem_get_info(torch.device(self.device))[0]
            if need > free:
                self._warn_once("pregen_memory", "NavGymEnv: %.0f GB free on the device, the pipelined reset path would need about "
                                "%.0f GB more than the world itself: pregen_pipeline falls back to 0 (navsim_regen after every step)"
                                % (free / 2 ** 30, need / 2 ** 30))
                self.pregen_pipeline = 0
                for c_ in (cfg, self.sim.cfg):             # (the simulator holds its own copy)
                    c_.regen_cap = min(self.num_envs, 64)
                    c_.defer_reset_scan = int(self.num_envs <= 1024)
                if self._use_graphs_arg is None:
                    self.use_graphs = not (self.plan_paths and float(self.indoor_ratio) > 0.0)
                self._overlap_replan = ("costmap" in self.sim.t and self.sim.due is not None and self.pedestrian_model != "policy")
        if self.pregen_pipeline:
            if cfg.field_format != abi.FIELD_U16T:
                raise ValueError("pregen_pipeline needs the packed distance field (map_size <= 1024)")
            if first:
                self.sim.enable_pregen(pipeline=self.pregen_pipeline, install=True, stage_cap=self.pregen_stage_cap,
                                       fallback_poll=self.pregen_fallback_poll)
            else:
                self.sim.restage_all()                    # the worlds behind the ones this reset() just drew
        if first and self.use_graphs and self.pedestrian_model != "policy":
            self.sim.enable_graphs(regen=self.randomize_maps and self.auto_reset,
                                   replan_cap=self.replan_cap if "costmap" in self.sim.t else 0)
            self._graphed = True
        self._map_info = None                             # read back from the device when somebody asks (map_info)
        self._humans_of_episode = None
        return self._obs_dict()

    @property
    def map_info(self):
        """Arena 0's map as RosEnv reads it (ros_env.py:69-81; env.py:297-310): built on first use after a reset() -- the
        occupancy grid is read back from the device's distance field, which step() and reset() themselves never need
        (round 4 copied it to the host in every reset())."""
        if self.sim is None:
            return None
        if self._map_info is None:
            cfg = self.cfg
            occ0 = self.sim.occupancy(0)
            live = self.map_size
            if self.outdoor_map_size and occ0[self.outdoor_map_size:, :].all() and occ0[:, self.outdoor_map_size:].all():
                live = self.outdoor_map_size              # arena 0 drew an outdoor map: the reference's 400 x 400 array
            self._map_info = {"data": (occ0[:live, :live].astype(np.int8) * 100), "origin": (cfg.origin_x, cfg.origin_y),
                              "resolution": cfg.resolution, "width": live, "height": live}
        return self._map_info

    @property
    def humans(self):
        """Views of arena 0's pedestrians (ros_env.py:120-176), as many as its current episode has."""
        if self.sim is None or "n_peds" not in self.sim.t:
            return []
        if self._humans_of_episode is None:
            self._humans_of_episode = [_AgentView(self, "human", i) for i in range(int(self.sim.t["n_peds"][0]))]
        return self._humans_of_episode

    @humans.setter
    def humans(self, value):
        self._humans_of_episode = list(value)

    def _obs_dict(self):
        # no kernel of its own: the observation rows and the two goal arrays are what the step (or navsim_regen's
        # first-observation launch) wrote (env.py:455-461: achieved_goal = the pose slots, desired_goal = the robot's goal)
        o = self.sim.obs
        d = {"observation": o, "achieved_goal": self.sim.out["achieved_goal"], "desired_goal": self.sim.out["desired_goal"]}
        if self.num_envs == 1:
            d = {k: v[0].double().cpu().numpy() for k, v in d.items()}
        self.prev_obs = d
        return d

    # ---- step (env.py:591-728) -------------------------------------------------------------------------
    def step(self, action, human_actions=None):
        if self.sim is None:
            raise RuntimeError("call reset() before step()")
        if human_actions is not None:
            self.sim.set_ped_cmd(human_actions)
        elif self.pedestrian_model == "policy":
            self.sim.ped_policy()                           # scans -> HumanPolicy actor -> (v, omega)
        a = np.asarray(action, dtype=np.float64).reshape(self.num_envs, 2) if not hasattr(action, "is_cuda") else action
        if self._graphed:                                   # step + regen + replan: one graph launch (NavSim.enable_graphs)
            _, out = self.sim.step_graphed(a)
        elif self.pregen_pipeline:
            # navsim_step_install: finished arenas take their staged worlds; with planned routes the re-plan of the PREVIOUS
            # step's arrivals inside the same launch where the search fits the arena's workgroup (else behind the step)
            in_step = "costmap" in self.sim.t and self.pedestrian_model != "policy"
            if in_step and self.sim.pg_replan_cap == 0:
                self.sim.pg_replan_cap = self.replan_cap
            _, out = self.sim.step(a)
            self.sim.regen()                               # ... and every P steps a staging pass goes to the side stream
            if "costmap" in self.sim.t and not (in_step and self.sim.pg_replan_in_step):
                self.sim.replan(self.replan_cap)
        elif self._overlap_replan:
            # planned routes: the re-plan of the previous step runs beside this step's launch (NavSim.launch_step_overlapped)
            _, out = self.sim.step_overlapped(a, self.replan_cap)
            if self.randomize_maps and self.auto_reset:
                self.sim.regen()
        else:
            _, out = self.sim.step(a)                      # (a float64 tensor on the device is read in place: no copy)
            if self.pedestrian_model == "policy" and self.auto_reset:
                # a new episode starts with prev_human_actions = 0 (env.py:739)
                started = self.sim.reset_flags if self.autoreset_mode == "next_step" else out["done"]
                self.sim.t["policy_prev_actions"].mul_((started == 0).to(self.sim.t["policy_prev_actions"].dtype)[:, None, None])
            if self.randomize_maps and self.auto_reset:
                self.sim.regen()
            if "costmap" in self.sim.t:
                self.sim.replan(self.replan_cap)           # ('policy': the control block in front of the next step reads the routes)
        obs = self._obs_dict()
        # the LAST observation of an episode that ended in this step (env.py:700-728 returns it with done = True): under
        # same-step auto-reset the row above already is the next episode's first one and the terminal one rides in info;
        # next-step auto-reset returns it as the observation and marks the arenas this step reset instead
        fin = self.sim.final
        if self.num_envs == 1:
            info = {"is_success": np.float32(out["is_success"][0].item()),
                    "is_crash": np.float32(out["is_crash"][0].item()),
                    "distance": float(out["distance"][0].item())}
            done = bool(out["done"][0].item())
            if fin is not None and done:
                g = fin["final_goals"][0].double().cpu().numpy()
                info["final_observation"] = {"observation": fin["final_obs"][0].double().cpu().numpy(),
                                             "achieved_goal": g[:2], "desired_goal": g[2:]}
            if self.auto_reset and self.autoreset_mode == "next_step":
                info["reset_mask"] = bool(self.sim.reset_flags[0].item())
            return obs, float(out["reward"][0].item()), done, info
        # (views of the buffers the kernel wrote, made ONCE per buffer parity: a step of 4096 arenas is 95 us on the device, and
        #  every tensor slice or .view() costs the host 3-5 us -- round 6 measured 42 -> 36 M env-steps/s through this method
        #  when the terminal rows' three views were built per call)
        v = self._views.get(self.sim.cur)
        if v is None:
            v = {"done": out["done"].view(self._bool)}
            if fin is not None:
                v["final"] = {"observation": fin["final_obs"], "achieved_goal": fin["final_goals"][:, :2],
                              "desired_goal": fin["final_goals"][:, 2:]}
            if self.auto_reset and self.autoreset_mode == "next_step":
                v["reset"] = self.sim.reset_flags.view(self._bool)
            self._views[self.sim.cur] = v
        info = {"is_success": out["is_success"], "is_crash": out["is_crash"], "distance": out["distance"]}
        done = v["done"]
        if fin is not None:
            info["final_observation"] = v["final"]
            info["final_mask"] = done
        if "reset" in v:
            info["reset_mask"] = v["reset"]
        return obs, out["reward"], done, info

    def counters(self, reset=True):
        """What the device-side reset path served and what its caps left waiting since the last call (abi.COUNTERS):
        regen_served / regen_unserved (finished arenas beyond cfg.regen_cap play their next episode on the old map),
        replan_served / replan_unserved (pedestrians beyond replan_cap wait for the next step), routes_cut (routes
        longer than max_waypoints) and routes_resumed (cut routes continued to their own goal)."""
        if self.sim is None:
            return {k: 0 for k in abi.COUNTERS}
        return self.sim.counters(reset=reset)

    def human_scans(self):
        """The pedestrians' own 512-beam half-plane scans (env.py:685-693), float32 [E, N, 512]: what
        the reference stacks and feeds to HumanPolicy.  With pedestrian_model='external' a caller can run
        that policy (or any other) on them and pass the resulting (v, w) to step(human_actions=...)."""
        return self.sim.ped_scans()

    # ---- HER batch API (env.py:464-589) ---------------------------------------------------------------
    def _rd(self, obs):
        import torch
        from . import sim as simmod
        o = torch.as_tensor(np.asarray(obs["observation"])) if not hasattr(obs["observation"], "is_cuda") else obs["observation"]
        g = torch.as_tensor(np.asarray(obs["desired_goal"])) if not hasattr(obs["desired_goal"], "is_cuda") else obs["desired_goal"]
        o = o.to(self.device)
        if o.dtype not in (torch.float32, torch.float64):
            o = o.double()
        return simmod.reward_done(self.sim.cfg, o.reshape(-1, o.shape[-1]), g.to(self.device).reshape(-1, 2),
                                  self.scan_threshold, self.scan_discomfort_threshold)

    def compute_rewards(self, actions, obs, make_render_reward_txt=False):
        return self._rd(obs)["reward"].cpu().numpy()

    def compute_terminals(self, obs):
        return self._rd(obs)["done"].cpu().numpy().astype(bool)

    def compute_reward(self, action, obs, make_render_reward_txt=False):
        return self.compute_rewards(np.asarray(action)[None], {k: np.asarray(v)[None] for k, v in obs.items()})[0]

    def compute_done(self, obs):
        return self.compute_terminals({k: np.asarray(v)[None] for k, v in obs.items()})[0]

    def compute_info(self, obs):
        r = self._rd({k: np.asarray(v)[None] for k, v in obs.items()})
        return {"is_success": np.float32(r["is_success"][0].item()), "is_crash": np.float32(r["is_crash"][0].item()),
                "distance": float(r["distance"][0].item())}

    def render(self, mode="human", arena=0, text=True):
        """The reference's picture of ONE arena (env.py:833-1050) as a float32 BGR array [800, 800, 3] in [0, 1]:
        map, goal, pedestrians with local goals, robot with its three rectangles, lidar returns.  Host-side NumPy
        (render.py); the debug text (env.py:1035-1046: observation tail and reward terms, `text=False` leaves it out) is drawn
        with a built-in font; the reference's OpenCV window is not opened -- both modes return the image.  `arena`
        selects which arena of the batch is drawn."""
        if self.sim is None:
            raise RuntimeError("call reset() before render()")
        from . import render as rd
        t, e, cfg = self.sim.t, int(arena), self.cfg
        occ = self.sim.occupancy(e)
        map_info = {"data": occ.astype(np.int8) * 100, "origin": (cfg.origin_x, cfg.origin_y),
                    "resolution": cfg.resolution, "width": self.map_size, "height": self.map_size}
        spec = robots.ROBOTS[self.robot_type]
        rp = t["robot_pose"][e].cpu().numpy()
        goal = t["robot_goal"][e].cpu().numpy()
        robot = dict(px=rp[0], py=rp[1], theta=rp[2], gx=goal[0], gy=goal[1], footprint=spec["footprint"],
                     threshold_footprint=spec["threshold_footprint"],
                     discomfort_threshold_footprint=spec["discomfort_threshold_footprint"])
        humans = []
        if "n_peds" in t:
            pp = t["ped_pose"][e].cpu().numpy()
            wp = t["ped_waypoints"][e, :, 0].cpu().numpy()
            for i in range(int(t["n_peds"][e])):
                humans.append(dict(px=pp[i, 0], py=pp[i, 1], theta=pp[i, 2], gx=wp[i, 0], gy=wp[i, 1],
                                   footprint=robots.HUMAN["footprint"]))
        o = self.sim.obs[e].cpu().numpy()
        B = cfg.n_beams
        scan = o[(cfg.n_scan_stack - 1) * B: cfg.n_scan_stack * B]
        img = rd.render_arena(map_info, robot, humans, scan, float(o[-1]),
                              dict(angle_min=cfg.angle_min, angle_last=cfg.angle_last, range_max=np.float32(cfg.range_max)))
        if text:
            self._make_render_txt(e)
            rd.overlay_text(img, self.render_obs_txt, self.render_reward_txt)
        return img

    def _make_render_txt(self, e=0):
        """render_obs_txt / render_reward_txt of arena e (env.py:182-217): the observation's tail and the six terms of
        compute_rewards for it, evaluated on the host from the device's latest observation (render.reward_terms)."""
        from . import render as rd
        cfg = self.cfg
        o = self.sim.obs[e].cpu().numpy().astype(np.float64)
        B, S = cfg.n_beams, cfg.n_scan_stack
        tail = o[S * B:]                                   # prev_pose(2) pose(2) vel(2) yaw (env.py:455)
        goal = self.sim.out["desired_goal"][e].cpu().numpy() if "desired_goal" in self.sim.out else self.sim.t["robot_goal"][e].cpu().numpy()
        steps = int(self.sim.t["steps"][e]) if "steps" in self.sim.t else 0
        self.render_obs_txt = rd.obs_text(steps, tail[0:2], tail[2:4], tail[4:6], tail[6], goal)
        factors = {k: getattr(self, k) for k in ("reward_scale", "reward_success_factor", "reward_crash_factor",
                                                 "reward_progress_factor", "reward_forward_factor", "reward_rotation_factor",
                                                 "reward_discomfort_factor")}
        terms = rd.reward_terms(self.sim.obs[e].cpu().numpy()[(S - 1) * B: S * B], tail[0:2], tail[2:4], tail[4:6], goal,
                                self.scan_threshold.cpu().numpy(), self.scan_discomfort_threshold.cpu().numpy(), factors,
                                self.distance_threshold)
        self.render_reward_txt = rd.reward_text(terms)
        return terms

    def close(self):
        if self.sim is not None:
            self.sim.close()                        # (staging passes in flight on a side stream: wait before the arrays go)
        self.sim = None
        self._views = {}

    # ---- EzPickle (env.py:30, 56-78): a pickle carries the constructor's arguments, the copy is built from them -----------
    def __getstate__(self):
        return {"_ezpickle_kwargs": dict(self._ctor_kwargs)}

    def __setstate__(self, d):
        self.__init__(**d["_ezpickle_kwargs"])

    def __reduce__(self):
        return (_rebuild_env, (type(self), dict(self._ctor_kwargs)))

    def state_dict(self):
        """The simulation itself (every device array of the state, the observation / output buffers, their parity) as
        host tensors: `load_state_dict` on an environment made with the same arguments and reset() once continues the
        rollout bit for bit.  (Not part of the reference's API: its pickles carry the constructor arguments only.)"""
        if self.sim is None:
            raise RuntimeError("call reset() before state_dict()")
        import torch
        torch.cuda.synchronize(self.sim.device)
        # (a world with slot tables -- the pipelined reset path -- holds 2 E map slots, E of them staged worlds that are not part
        #  of the snapshot: the arenas' maps are saved in arena order, the table itself is not)
        sd = {}
        for k in self.sim.t:
            if k.startswith(("replan_ws", "regen_ws", "policy_ws")) or k == "map_slot":
                continue
            sd["t." + k] = self.sim.by_arena(k).detach().cpu().clone()
        for i in (0, 1):
            sd["obs.%d" % i] = self.sim.obs_buf[i].cpu().clone()
            for k, v in self.sim.out_buf[i].items():
                sd["out.%d.%s" % (i, k)] = v.cpu().clone()
            if self.sim.due is not None:
                sd["due.%d" % i] = self.sim.due[i].cpu().clone()
            if self.sim.final_buf is not None:
                for k, v in self.sim.final_buf[i].items():
                    sd["final.%d.%s" % (i, k)] = v.cpu().clone()
        sd["cur"] = int(self.sim.cur)
        sd["steps_launched"] = int(self.sim._steps_launched)
        sd["episode_batch"] = int(self._episode_batch)
        return sd

    def load_state_dict(self, sd):
        if self.sim is None:
            raise RuntimeError("call reset() once before load_state_dict() (it allocates the device arrays)")
        import torch
        torch.cuda.synchronize(self.sim.device)
        slots = self.sim.t.get("map_slot")
        if slots is not None:                              # the snapshot's maps go to the slots 0 .. E - 1, in arena order
            slots.copy_(torch.arange(self.num_envs, dtype=slots.dtype, device=slots.device))
        for k, v in sd.items():
            if k.startswith("t."):
                dst = self.sim.t[k[2:]]
                if slots is not None and k[2:] in self.sim.MAPS:     # (2 E slots; a packed field is one flat blob)
                    dst.reshape(2 * self.num_envs, -1)[: self.num_envs].copy_(v.reshape(self.num_envs, -1))
                else:
                    dst.copy_(v)
        for i in (0, 1):
            self.sim.obs_buf[i].copy_(sd["obs.%d" % i])
            for k, v in self.sim.out_buf[i].items():
                v.copy_(sd["out.%d.%s" % (i, k)])
            if self.sim.due is not None:
                self.sim.due[i].copy_(sd["due.%d" % i])
            if self.sim.final_buf is not None:
                for k, v in self.sim.final_buf[i].items():
                    if "final.%d.%s" % (i, k) in sd:
                        v.copy_(sd["final.%d.%s" % (i, k)])
        self.sim.cur = int(sd["cur"])
        if getattr(self.sim, "pregen", False):             # the staged worlds are not part of the snapshot: stage them again
            self.sim.restage_all(slots_from_live=True)
        self.sim._steps_launched = int(sd["steps_launched"])
        self._episode_batch = int(sd["episode_batch"])
        self._map_info = None
        self._humans_of_episode = None
        torch.cuda.synchronize(self.sim.device)
        self._obs_dict()


def _rebuild_env(cls, kwargs):
    return cls(**kwargs)
