"""ctypes mirror of include/navsim.h (the C ABI of the batched NavGym step).

Pure declarations: importing this module touches neither the GPU nor the shared library.
Field order and types must match include/navsim.h exactly; tests/test_abi.py compares
``ctypes.sizeof`` with the ``sizeof`` the compiled library reports.
"""
import ctypes as C

ABI_VERSION = 6

OK = 0
E_ARG = -1
E_LAUNCH = -2
E_UNSUPPORTED = -3
E_NODEVICE = -4

PED_NONE = 0
PED_EXTERNAL = 1
PED_SFM = 2

MAX_PEDS = 64
MAX_WAYPOINTS = 256     # upper limit of cfg.max_waypoints (the stride P of ped_waypoints; navsim_default_config: 64)
OBS_TAIL = 7

FIELD_F32 = 0
FIELD_U16T = 1
FIELD_TILE = 8

MARCH_F64 = 0          # t += max(fl32(fl64(d) * 0.999), 1)
MARCH_F32 = 1          # t += max(d * 0.999f, 1): RangeLib.h's float member step_coeff (default since ABI 4)
MARCH_F32_FMA = 2      # MARCH_F32 with the sample position contracted into an FMA (GCC -O3 -march=native on FMA hardware)
MARCH_RULES = (MARCH_F64, MARCH_F32, MARCH_F32_FMA)

STEP_ALL, STEP_NOT_DUE, STEP_DUE = 0, 1, 2      # navsim_step_part

AUTORESET_NONE, AUTORESET_SAME_STEP, AUTORESET_NEXT_STEP = 0, 1, 2      # cfg.auto_reset (include/navsim.h NAVSIM_AUTORESET_*)

ACTION_TWIST = 0       # io.action = (v, omega)
ACTION_WHEELS = 1      # io.action = (omega_left, omega_right) of a skid-steer base, rad/s

N_COUNTERS = 8
COUNTERS = ("regen_served", "regen_unserved", "replan_served", "replan_unserved", "routes_cut", "routes_resumed",
            "regen_short", "regen_late")


class NavsimConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32),
        ("n_beams", C.c_int32),
        ("map_h", C.c_int32),
        ("map_w", C.c_int32),
        ("max_peds", C.c_int32),
        ("n_scan_stack", C.c_int32),
        ("ped_model", C.c_int32),
        ("lidar_legs", C.c_int32),
        ("auto_reset", C.c_int32),
        ("n_spawn", C.c_int32),
        ("add_scan_noise", C.c_int32),
        ("env_index_base", C.c_int32),
        ("field_format", C.c_int32),
        ("shared_field", C.c_int32),
        ("resolution", C.c_double),
        ("origin_x", C.c_double),
        ("origin_y", C.c_double),
        ("time_step", C.c_double),
        ("angle_min", C.c_double),
        ("angle_last", C.c_double),
        ("range_max", C.c_double),
        ("axle_offset", C.c_double),
        ("min_turning_radius", C.c_double),
        ("distance_threshold", C.c_double),
        ("reward_scale", C.c_double),
        ("reward_success_factor", C.c_double),
        ("reward_crash_factor", C.c_double),
        ("reward_progress_factor", C.c_double),
        ("reward_forward_factor", C.c_double),
        ("reward_rotation_factor", C.c_double),
        ("reward_discomfort_factor", C.c_double),
        ("sfm_tau", C.c_double),
        ("sfm_k_desired", C.c_double),
        ("sfm_k_social", C.c_double),
        ("sfm_k_obstacle", C.c_double),
        ("sfm_lambda", C.c_double),
        ("sfm_gamma", C.c_double),
        ("sfm_n", C.c_double),
        ("sfm_n_prime", C.c_double),
        ("sfm_sigma_obstacle", C.c_double),
        ("sfm_agent_radius", C.c_double),
        ("seed", C.c_uint64),
        ("ped_angle_min", C.c_double),
        ("ped_angle_last", C.c_double),
        ("ped_range_max", C.c_double),
        ("ped_n_beams", C.c_int32),
        ("regen_plan", C.c_int32),
        ("robot_seen_footprint", C.c_double * 8),
        ("regen_cap", C.c_int32),
        ("obstacle_number", C.c_int32),
        ("obstacle_width_lo", C.c_double),
        ("obstacle_width_hi", C.c_double),
        ("spawn_clearance", C.c_double),
        ("ped_clearance", C.c_double),
        ("min_goal_dist", C.c_double),
        ("max_goal_dist", C.c_double),
        ("ped_min_robot_dist", C.c_double),
        ("ped_min_goal_dist", C.c_double),
        ("v_pref_lo", C.c_double),
        ("v_pref_hi", C.c_double),
        ("has_legs_ratio", C.c_double),
        ("regen_indoor_ratio", C.c_double),
        ("obstacle_number_hi", C.c_int32),
        ("corridor_width_lo", C.c_int32),
        ("corridor_width_hi", C.c_int32),
        ("iterations_lo", C.c_int32),
        ("iterations_hi", C.c_int32),
        ("num_humans_lo", C.c_int32),
        ("num_humans_hi", C.c_int32),
        ("outdoor_map_size", C.c_int32),
        ("scan_noise_std_lo", C.c_double),
        ("scan_noise_std_hi", C.c_double),
        ("march_rule", C.c_int32),
        ("step_block", C.c_int32),
        ("ped_split", C.c_int32),
        ("regen_check_discomfort", C.c_int32),
        ("rect_lds", C.c_int32),
        ("max_waypoints", C.c_int32),
        ("action_kind", C.c_int32),
        ("clamp_action", C.c_int32),
        ("wheel_radius", C.c_double),
        ("wheel_track", C.c_double),
        ("linvel_lo", C.c_double),
        ("linvel_hi", C.c_double),
        ("rotvel_lo", C.c_double),
        ("rotvel_hi", C.c_double),
        ("closed_maps", C.c_int32),
        ("defer_reset_scan", C.c_int32),
        ("regen_min_steps", C.c_int32),
    ]

    def copy(self):
        other = NavsimConfig()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(self))
        return other


_P = C.c_void_p


class NavsimState(C.Structure):
    _fields_ = [(name, _P) for name in (
        "field", "field_overflow", "rect_table", "beam_table", "scan_threshold", "scan_discomfort", "scan_noise_std",
        "robot_pose", "robot_goal", "prev_action", "prev_pose", "n_hist", "episode", "steps",
        "n_peds", "ped_pose", "ped_vel", "ped_prev_yaw", "ped_dist", "ped_v_pref", "ped_has_legs",
        "ped_waypoints", "ped_n_waypoints", "ped_cmd",
        "spawn_pose", "spawn_goal", "costmap", "arena_cost", "launch_order", "regen_draws",
        "ped_goal", "counters", "rect_index",
        "ped_wp_head", "ped_due", "ped_due_prev", "done_steps", "map_slot",
    )]


POLICY_FIELDS = ("cv1_w", "cv1_b", "cv2_w", "cv2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "a1_w", "a1_b", "a2_w", "a2_b")
POLICY_SHAPES = {"cv1_w": (32, 3, 5), "cv1_b": (32,), "cv2_w": (32, 32, 3), "cv2_b": (32,), "fc1_w": (256, 4096),
                 "fc1_b": (256,), "fc2_w": (128, 260), "fc2_b": (128,), "a1_w": (128,), "a1_b": (1,), "a2_w": (128,),
                 "a2_b": (1,)}
# reference state_dict names (human_policy.py:24-30) -> fields of navsim_policy_weights
POLICY_STATE_DICT = {"act_fea_cv1.weight": "cv1_w", "act_fea_cv1.bias": "cv1_b", "act_fea_cv2.weight": "cv2_w",
                     "act_fea_cv2.bias": "cv2_b", "act_fc1.weight": "fc1_w", "act_fc1.bias": "fc1_b",
                     "act_fc2.weight": "fc2_w", "act_fc2.bias": "fc2_b", "actor1.weight": "a1_w", "actor1.bias": "a1_b",
                     "actor2.weight": "a2_w", "actor2.bias": "a2_b"}


class NavsimPolicyWeights(C.Structure):
    _fields_ = [(name, _P) for name in POLICY_FIELDS]


class NavsimCrowdParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "time_step", "discomfort_dist", "map_size_m", "map_resolution", "success_reward", "collision_penalty",
        "discomfort_penalty_factor", "rotation_penalty_factor", "timeout_penalty", "time_limit")]


class NavsimCrowdMapParams(C.Structure):
    _fields_ = [("angular_min", C.c_double), ("angular_max", C.c_double), ("angular_max_range", C.c_double),
                ("angular_dim", C.c_int32), ("normalize", C.c_int32), ("map_size_m", C.c_double),
                ("map_resolution", C.c_double), ("submap_size_m", C.c_double)]


class NavsimOrcaParams(C.Structure):
    _fields_ = [("time_step", C.c_float), ("neighbor_dist", C.c_float), ("time_horizon", C.c_float),
                ("time_horizon_obst", C.c_float), ("max_neighbors", C.c_int32)]


ORCA_MAX_AGENTS = 64
ORCA_MAX_EDGES = 128
CROWD_MAX_VERTS = 8
CROWD_INFO = ("Nothing", "Timeout", "ReachGoal", "Collision", "CollisionOtherAgent", "Danger")


class NavsimStepIO(C.Structure):
    _fields_ = [(name, _P) for name in (
        "action", "obs_prev", "obs", "achieved_goal", "desired_goal",
        "reward", "done", "is_success", "is_crash", "distance",
        "final_obs", "final_goals", "reset_mask",          # ABI 6
    )]


# dtype / trailing shape of every state array, in units of (E, N, B, S, K, P, H, W)
STATE_LAYOUT = {
    "field": ("float32", ("E", "H", "W")),          # FIELD_F32; a uint8 blob for packed formats
    "field_overflow": ("float32", ("E", "H", "W")),
    "rect_table": ("int32", ("E", "T", 4)),         # 16-byte two-rectangle records of the 8x8 tiles
    "beam_table": ("float64", ("B", 2)),
    "scan_threshold": ("float32", ("B",)),
    "scan_discomfort": ("float32", ("B",)),
    "scan_noise_std": ("float32", ("E",)),
    "robot_pose": ("float64", ("E", 3)),
    "robot_goal": ("float64", ("E", 2)),
    "prev_action": ("float64", ("E", 2)),
    "prev_pose": ("float64", ("E", 3)),
    "n_hist": ("int32", ("E",)),
    "episode": ("int64", ("E",)),
    "steps": ("int64", ("E",)),
    "n_peds": ("int32", ("E",)),
    "ped_pose": ("float64", ("E", "N", 3)),
    "ped_vel": ("float64", ("E", "N", 2)),
    "ped_prev_yaw": ("float64", ("E", "N")),
    "ped_dist": ("float64", ("E", "N", 3)),
    "ped_v_pref": ("float64", ("E", "N")),
    "ped_has_legs": ("uint8", ("E", "N")),
    "ped_waypoints": ("float64", ("E", "N", "P", 2)),
    "ped_n_waypoints": ("int32", ("E", "N")),
    "ped_cmd": ("float64", ("E", "N", 2)),
    "spawn_pose": ("float64", ("E", "K", 3)),
    "spawn_goal": ("float64", ("E", "K", 2)),
    "costmap": ("uint8", ("E", "Hc", "Wc")),
    "arena_cost": ("int32", ("E",)),
    "launch_order": ("int32", ("E",)),
    "regen_draws": ("float64", ("E", 464)),         # tests only: draws supplied (NAVSIM_DRAW_* layout)
    "ped_goal": ("float64", ("E", "N", 2)),         # goal of every pedestrian's current route
    "counters": ("int64", (N_COUNTERS,)),           # uint64 on the device; the totals stay far below 2^63
    "rect_index": ("uint8", ("E", "R")),            # index form of rect_table: 256 rectangles x 8 B + 2 B per tile, per arena
    "ped_wp_head": ("int32", ("E", "N")),           # ABI 5: index of every pedestrian's current waypoint (the pop advances it)
    "ped_due": ("int64", ("E",)),                   # uint64 on the device: bit i = pedestrian i waits for navsim_replan (written by the step)
    "ped_due_prev": ("int64", ("E",)),              # the flags of the previous step (navsim_step_part)
    "done_steps": ("int32", ("E",)),                # length of the episode that ended last (cfg.regen_min_steps)
    "map_slot": ("int32", ("E",)),                  # slot of the five per-map arrays that holds arena e's map (NULL: e)
}

IO_LAYOUT = {
    "action": ("float64", ("E", 2)),
    "obs_prev": ("float32", ("E", "D")),
    "obs": ("float32", ("E", "D")),
    "achieved_goal": ("float32", ("E", 2)),
    "desired_goal": ("float32", ("E", 2)),
    "reward": ("float64", ("E",)),
    "done": ("uint8", ("E",)),
    "is_success": ("float32", ("E",)),
    "is_crash": ("float32", ("E",)),
    "distance": ("float64", ("E",)),
}


# ABI 6: the terminal observation of arenas that restart inside the step that ends their episode (NAVSIM_AUTORESET_SAME_STEP)
FINAL_LAYOUT = {
    "final_obs": ("float32", ("E", "D")),
    "final_goals": ("float32", ("E", 4)),           # achieved_goal (2), desired_goal (2) of that observation
}


def rect_index_row_bytes(H, W):
    """navsim_rect_index_bytes(1, H, W): 256 list entries of 8 bytes, then 2 bytes per 8x8 tile (padded to 16)."""
    return 256 * 8 + ((((H + 7) // 8) * ((W + 7) // 8) * 2 + 15) // 16) * 16


def resolve_shape(shape, cfg):
    """Turns a symbolic shape of STATE_LAYOUT / IO_LAYOUT into integers for `cfg`."""
    sym = {
        "E": cfg.n_envs, "N": cfg.max_peds, "B": cfg.n_beams, "S": cfg.n_scan_stack,
        "K": max(cfg.n_spawn, 1), "P": cfg.max_waypoints, "H": cfg.map_h, "W": cfg.map_w,
        "D": cfg.n_scan_stack * cfg.n_beams + OBS_TAIL, "Hc": cfg.map_h // 5, "Wc": cfg.map_w // 5,
        "T": ((cfg.map_h + 7) // 8) * ((cfg.map_w + 7) // 8),
        "R": rect_index_row_bytes(cfg.map_h, cfg.map_w),
    }
    return tuple(sym[s] if isinstance(s, str) else s for s in shape)


def declare(lib, suffix=""):
    """Attaches argtypes/restypes for every entry point of include/navsim.h to `lib`.

    suffix="" is the HIP library (trailing stream argument); suffix="_cpu" is the oracle.
    """
    stream = [] if suffix else [_P]
    i32, f32, f64 = C.c_int32, C.c_float, C.c_double
    cfgp, stp, iop = C.POINTER(NavsimConfig), C.POINTER(NavsimState), C.POINTER(NavsimStepIO)

    def sig(name, args, res=C.c_int):
        fn = getattr(lib, name + suffix)
        fn.argtypes = args
        fn.restype = res
        return fn

    sig("navsim_default_config", [cfgp])
    if suffix:
        sig("navsim_build_dt", [_P, i32, i32, i32, _P])
    else:
        sig("navsim_build_dt", [_P, i32, i32, i32, _P, _P, C.c_size_t, _P])
    if not suffix:
        sig("navsim_field_bytes", [i32, i32, i32, i32], C.c_size_t)
        sig("navsim_build_field", [_P, i32, i32, i32, i32, _P, _P, _P, _P, C.c_size_t, _P])
        sig("navsim_rect_table_bytes", [i32, i32, i32], C.c_size_t)
        sig("navsim_build_rects_workspace_bytes", [i32, i32, i32], C.c_size_t)
        sig("navsim_build_rects", [_P, i32, i32, i32, _P, i32, _P, _P, _P, C.c_size_t, _P])
        sig("navsim_rect_index_bytes", [i32, i32, i32], C.c_size_t)
        sig("navsim_build_rect_index", [_P, i32, i32, i32, _P, _P, _P])
        sig("navsim_maps_closed", [_P, i32, i32, i32, _P, _P])
        sig("navsim_world_closed", [cfgp, stp, _P, _P])
    sig("navsim_cast_static", [_P, i32, i32, i32, _P, i32, f32, i32, _P] + stream)
    sig("navsim_render_polys", [_P, _P, i32, i32, _P, _P, i32, _P] + stream)
    sig("navsim_render_legs", [_P, _P, i32, i32, _P, _P, i32, _P] + stream)
    sig("navsim_integrate", [_P, _P, _P, i32, f64, f64] + stream)
    sig("navsim_reward_done", [cfgp, _P, _P, i32, i32, _P, _P, _P, _P, _P, _P, _P] + stream)
    sig("navsim_scan_threshold", [cfgp, _P, i32, _P] + stream)
    if not suffix:
        sig("navsim_beam_table", [cfgp, _P, _P])
        sig("navsim_debug_xy_to_ij", [cfgp, _P, i32, _P, i32, _P])
        sig("navsim_debug_kernarg_layout", [_P])
    sig("navsim_ped_scans", [cfgp, stp, _P] + stream)
    if not suffix:
        sig("navsim_ped_scans_part", [cfgp, stp, _P, i32, i32, _P])
        sig("navsim_ped_policy_part", [cfgp, stp, C.POINTER(NavsimPolicyWeights), _P, _P, _P, _P, C.c_size_t, i32, i32, _P])
    if suffix:
        sig("navsim_regen", [cfgp, stp, iop])
        sig("navsim_replan", [cfgp, stp, i32])
        sig("navsim_ped_policy", [cfgp, stp, C.POINTER(NavsimPolicyWeights), _P, _P, _P])
    else:
        sig("navsim_ped_policy_workspace_bytes", [cfgp], C.c_size_t)
        sig("navsim_ped_policy", [cfgp, stp, C.POINTER(NavsimPolicyWeights), _P, _P, _P, _P, C.c_size_t, _P])
        sig("navsim_ped_scan_policy", [cfgp, stp, C.POINTER(NavsimPolicyWeights), _P, _P, _P, _P, C.c_size_t, _P])
        sig("navsim_launch_order", [_P, _P, i32, _P])
        sig("navsim_replan_workspace_bytes", [cfgp, i32], C.c_size_t)
        sig("navsim_replan", [cfgp, stp, i32, _P, C.c_size_t, _P])
        sig("navsim_costmap", [_P, i32, i32, i32, _P, _P])
        sig("navsim_plan", [_P, _P, i32, i32, i32, f64, f64, f64, _P, _P, f64, i32, _P, _P, _P, _P, _P])
        sig("navsim_regen_workspace_bytes", [cfgp], C.c_size_t)
        sig("navsim_regen", [cfgp, stp, iop, _P, C.c_size_t, _P])
        sig("navsim_regen_swap", [cfgp, stp, stp, iop, _P, _P, _P, _P, _P])
        sig("navsim_regen_stage", [cfgp, stp, iop, _P, _P, _P, _P, C.c_size_t, _P])
        sig("navsim_regen_stage_part", [cfgp, stp, iop, _P, _P, _P, _P, C.c_size_t, i32, i32, _P])
        sig("navsim_step_install", [cfgp, stp, iop, stp, _P, _P, _P, _P, _P])
        sig("navsim_regen_helper", [_P])
        sig("navsim_step_install_replan", [cfgp, stp, iop, stp, _P, _P, _P, _P, C.c_int32, _P])
        sig("navsim_step_install_next", [cfgp, stp, iop, stp, _P, _P, _P, _P, _P, C.c_int32, _P])
    sig("navsim_crowd_check", [C.POINTER(NavsimCrowdParams), i32, i32, i32, _P, _P, _P, _P, _P, _P, _P, _P, _P] + stream)
    mpp = C.POINTER(NavsimCrowdMapParams)
    sig("navsim_crowd_angular_map", [mpp, i32, i32, i32, _P, _P, _P, _P] + stream)
    sig("navsim_crowd_local_map", [mpp, i32, i32, _P, _P, i32, _P] + stream)
    sig("navsim_crowd_orca", [C.POINTER(NavsimOrcaParams), i32, i32, _P, _P, _P, i32, i32, _P, _P, _P, _P, _P, _P] + stream)
    sig("navsim_crowd_agent_step", [_P, _P, _P, i32, f64] + stream)
    if not suffix:
        sig("navsim_prepare", [cfgp, stp, iop])
    sig("navsim_step", [cfgp, stp, iop] + stream)
    if not suffix:
        sig("navsim_step_part", [cfgp, stp, iop, i32, _P])
        sig("navsim_step_replan", [cfgp, stp, iop, i32, _P])
    sig("navsim_reset_obs", [cfgp, stp, iop, _P] + stream)
    sig("navsim_restart", [cfgp, stp, _P] + stream)
    return lib


# every symbol include/navsim.h declares (tests check the .so exports all of them)
EXPORTS = (
    "navsim_abi_version", "navsim_error_string", "navsim_last_hip_error", "navsim_default_config",
    "navsim_build_dt_workspace_bytes", "navsim_build_dt", "navsim_field_bytes", "navsim_build_field",
    "navsim_rect_table_bytes", "navsim_build_rects_workspace_bytes", "navsim_build_rects",
    "navsim_rect_index_bytes", "navsim_build_rect_index", "navsim_maps_closed", "navsim_world_closed",
    "navsim_cast_static",
    "navsim_render_polys", "navsim_render_legs", "navsim_integrate", "navsim_reward_done",
    "navsim_scan_threshold", "navsim_beam_table", "navsim_ped_scans", "navsim_ped_scans_part", "navsim_ped_policy_part", "navsim_regen_workspace_bytes", "navsim_regen", "navsim_regen_swap", "navsim_regen_stage", "navsim_regen_stage_part", "navsim_step_install", "navsim_regen_helper", "navsim_step_install_replan", "navsim_step_install_next",
    "navsim_costmap", "navsim_plan", "navsim_launch_order", "navsim_replan_workspace_bytes", "navsim_replan", "navsim_ped_policy_workspace_bytes", "navsim_ped_policy", "navsim_ped_scan_policy",
    "navsim_crowd_check", "navsim_crowd_angular_map", "navsim_crowd_local_map", "navsim_crowd_orca", "navsim_crowd_agent_step",
    "navsim_step", "navsim_step_part", "navsim_step_replan", "navsim_prepare", "navsim_reset_obs", "navsim_restart", "navsim_step_kernel_name",
    "navsim_sizeof_config", "navsim_sizeof_state", "navsim_sizeof_step_io", "navsim_debug_math", "navsim_debug_xy_to_ij",
    "navsim_debug_gather", "navsim_debug_kernarg_layout", "navsim_debug_set_stamps", "navsim_debug_spawn_decisions",
)
