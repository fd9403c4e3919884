/* navsim.h -- C ABI of the MI355X-native batched NavGym step().
 *
 * This is the drop-in boundary for the hot path of leekwoon/nav-gym
 * (NavGymEnv.step, nav_gym/src/nav_gym_env/env.py:591-728).  The reference has no
 * FFI of its own: its hot arithmetic lives behind four pip packages
 * (range_libc, CMap2D, pose2d, pyastar2d; env.py:12-15).  Every entry point below
 * names the reference call site it replaces.  All pointers are DEVICE pointers
 * owned by the caller (torch tensors in the Python host); the library never
 * allocates or frees caller-visible memory, never synchronises the device, and
 * never throws: every function returns 0 on success or a negative NAVSIM_E_* code.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *
 * The CPU oracle (oracle/navsim_ref.h) exports the same functions with a `_cpu`
 * suffix, host pointers and no stream argument.  The oracle is test
 * infrastructure; nothing in this library calls it.
 *
 * Conventions:
 *   occupancy[y][x] row-major, x = column = "i", y = row = "j"; cell (i,j) covers
 *   world [ox + i*res, ox + (i+1)*res) x [oy + j*res, ...).  Angles are CCW from +x.
 */
#ifndef NAVSIM_H
#define NAVSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAVSIM_ABI_VERSION 6

/* error codes */
#define NAVSIM_OK            0
#define NAVSIM_E_ARG        -1   /* null pointer / bad size */
#define NAVSIM_E_LAUNCH     -2   /* hipGetLastError() after a launch was not hipSuccess */
#define NAVSIM_E_UNSUPPORTED -3  /* configuration outside compiled limits */
#define NAVSIM_E_NODEVICE   -4   /* no HIP device visible */

/* pedestrian update model (navsim_config.ped_model) */
#define NAVSIM_PED_NONE      0   /* n_peds ignored, no pedestrians */
#define NAVSIM_PED_EXTERNAL  1   /* (v, omega) per pedestrian supplied by the caller: the slot the
                                    reference fills with HumanPolicy (env.py:650-662) */
#define NAVSIM_PED_SFM       2   /* build-defined social-force model */

/* distance-field storage (navsim_config.field_format, navsim_build_field) */
#define NAVSIM_FIELD_F32     0   /* float32 [E,H,W] row-major, distance in cells (what range_libc holds) */
#define NAVSIM_FIELD_U16T    1   /* uint16 squared distance in 8x8-cell tiles (128-B lines), 0xFFFF =
                                    "d2 >= 65535, read the float32 overflow plane"; d = sqrtf(d2) is
                                    bit-identical to the float32 field */
#define NAVSIM_FIELD_TILE    8

/* How range_libc's RayMarching::calc_range (env.py:425) rounds its sphere-tracing step `d * step_coeff` and its sample
 * position `x0 + ray_direction_x * t` (navsim_config.march_rule).  The package's source is not in the reference tree,
 * so which of these the upstream BINARY computes is UNPINNED; all are implemented on both sides
 * and the switch is this one field.  DESIGN.md records how many rays change their hit cell between them.
 *   The published header (RangeLib.h, class RayMarching) declares the coefficient as a float data member
 *   (`float step_coeff = 0.999;`, beside `float distThreshold = 0.0;`), so `d * step_coeff` is a float32 product:
 *   NAVSIM_MARCH_F32 is the reading of the SOURCE and the default since ABI 4 (rounds 1-3 defaulted to the double
 *   reading, NAVSIM_MARCH_F64, kept as a switch).  What the source cannot settle is the compiler: upstream's
 *   setup.py builds with -O3 -march=native -ffast-math, and GCC contracts a * b + c into one fused multiply-add
 *   wherever the target has FMA (-ffp-contract=fast is GCC's default outside ISO mode), i.e. the sample position
 *   becomes fmaf(dir, t, x0) on every x86 since Haswell: NAVSIM_MARCH_F32_FMA.  (The hit distance
 *   sqrtf(xd * xd + yd * yd) is unaffected: xd, yd are integers below 2^12, both products are exact.)
 *   A fourth family of last-bit differences has NO switch: upstream takes the ray direction from the C library's cosf / sinf
 *   of the float heading, this build from the correctly rounded fl32(cos64(fl64(heading))).  Measured (round 6,
 *   tests/test_oracle_crosscheck.py direction_rounding_sensitivity, 10^7 rays on twenty 500 x 500 maps, glibc 2.35): the two
 *   differ in dx or dy for 2.6 % of the headings, and 2 rays in 10^7 then change their hit cell (one ulp on either component
 *   changes about 100 in 10^7) -- an order of magnitude below the rules above (26 and 62 in 10^7). */
#define NAVSIM_MARCH_F64     0   /* t += max(fl32(fl64(d) * 0.999), 1): a double coefficient */
#define NAVSIM_MARCH_F32     1   /* t += max(d * 0.999f, 1): the float member `step_coeff` of the source (default) */
#define NAVSIM_MARCH_F32_FMA 2   /* NAVSIM_MARCH_F32 with the sample position contracted: px = (int)fmaf(dx, t, x0) */

/* compiled limits */
#define NAVSIM_MAX_PEDS      64
#define NAVSIM_MAX_WAYPOINTS 256 /* upper limit of navsim_config.max_waypoints (the stride P of ped_waypoints) */
#define NAVSIM_OBS_TAIL      7   /* prev_pose(2) pose(2) vel(2) yaw(1): env.py:455 */

/* ------------------------------------------------------------------------------------------
 * Simulator configuration: the registered kwargs of NavGym-v0 (__init__.py:6-38) plus the
 * batch geometry.  Plain data, passed by pointer, copied by the callee.
 * ---------------------------------------------------------------------------------------- */
typedef struct navsim_config {
    int32_t n_envs;          /* E: independent arenas in this shard */
    int32_t n_beams;         /* B: KetiRobot.n_angles (keti_robot.py:48); 512 native, 1081 bench */
    int32_t map_h;           /* H cells (rows, y) */
    int32_t map_w;           /* W cells (cols, x) */
    int32_t max_peds;        /* N: stride of the per-pedestrian arrays, <= NAVSIM_MAX_PEDS */
    int32_t n_scan_stack;    /* S: num_scan_stack (__init__.py:11) */
    int32_t ped_model;       /* NAVSIM_PED_* */
    int32_t lidar_legs;      /* robot scan renders legs of has_legs pedestrians (env.py:695-698) */
    int32_t auto_reset;      /* NAVSIM_AUTORESET_*: what happens to an arena whose episode ends (below) */
    int32_t n_spawn;         /* K: spawn table entries per env (auto_reset) */
    int32_t add_scan_noise;  /* 1: Gaussian noise on beams != range_max (env.py:437-440) */
    int32_t env_index_base;  /* global index of local env 0 (multi-GPU sharding; seeds RNG) */
    int32_t field_format;    /* NAVSIM_FIELD_* of navsim_state.field */
    int32_t shared_field;    /* 1: every arena reads arena 0's field (one map for the whole batch) */

    double resolution;       /* metres per cell (map_generator.py:116) */
    double origin_x, origin_y;
    double time_step;        /* __init__.py:8 */
    double angle_min;        /* first beam angle in the robot frame (keti_robot.py:45) */
    double angle_last;       /* last beam angle = angle_max - angle_increment (env.py:388-390) */
    double range_max;        /* keti_robot.py:47 */
    double axle_offset;      /* 0.14474 for KetiRobot (keti_robot.py:72-90); 0 = plain unicycle */
    double min_turning_radius;   /* env.py:595-600 */
    double distance_threshold;   /* __init__.py:10 */
    double reward_scale;
    double reward_success_factor;
    double reward_crash_factor;
    double reward_progress_factor;
    double reward_forward_factor;
    double reward_rotation_factor;
    double reward_discomfort_factor;

    /* social-force parameters (NAVSIM_PED_SFM only) */
    double sfm_tau;          /* relaxation time, s */
    double sfm_k_desired;
    double sfm_k_social;
    double sfm_k_obstacle;
    double sfm_lambda;       /* lambdaImportance */
    double sfm_gamma;
    double sfm_n;
    double sfm_n_prime;
    double sfm_sigma_obstacle;
    double sfm_agent_radius;

    uint64_t seed;           /* counter-based RNG key (scan noise, spawn choice) */

    /* pedestrian lidar (human.py:12-16) and the robot as pedestrians see it (env.py:404-405):
     * only used by navsim_ped_scans */
    double ped_angle_min, ped_angle_last, ped_range_max;
    int32_t ped_n_beams;
    int32_t regen_plan;               /* navsim_regen: 1 = sample on the costmap and keep only start/goal pairs
                                         joined by a path (env.py:342-383), pedestrians get path waypoints */
    double robot_seen_footprint[8];   /* threshold_footprint, 4 x (x, y) in the robot frame */

    /* device-side reset of finished arenas with a NEW map (navsim_regen):
     * the per-episode ranges of the reference (__init__.py:14-15, 17-18, 26-37; env.py:748-806) */
    int32_t regen_cap;                /* arenas regenerated per call at most (the rest keep their map) */
    int32_t obstacle_number;          /* env_param_range['obstacle_number'] */
    double obstacle_width_lo, obstacle_width_hi;   /* env_param_range['obstacle_width'] */
    double spawn_clearance;           /* robot start / goal cells keep this distance to obstacles, m */
    double ped_clearance;             /* same for pedestrians */
    double min_goal_dist, max_goal_dist;           /* __init__.py:17-18 */
    double ped_min_robot_dist;        /* env.py:372: 4 m */
    double ped_min_goal_dist;         /* env.py:788-791: 10 m */
    double v_pref_lo, v_pref_hi;      /* human_v_pref_range */
    double has_legs_ratio;            /* human_has_legs_ratio */
    double regen_indoor_ratio;        /* navsim_regen: probability that a new map is a corridor map
                                         (create_indoor_map, map_generator.py:97-123) instead of an outdoor one;
                                         env.py:742 indoor_ratio.  0 = outdoor only */

    /* The other per-episode ranges of env_param_range (__init__.py:28-37), drawn by navsim_regen for every new
     * episode like _sample_env_param does (env.py:281-292: 'int' = uniform over lo..hi inclusive, 'float' =
     * uniform(lo, hi)), hash-keyed by (seed, global arena, episode). */
    int32_t obstacle_number_hi;       /* 'obstacle_number' = obstacle_number .. obstacle_number_hi boxes (<= 64);
                                         below obstacle_number (the default 0): always obstacle_number */
    int32_t corridor_width_lo, corridor_width_hi;   /* 'corridor_width' (3, 4) */
    int32_t iterations_lo, iterations_hi;           /* 'iterations' (80, 150) */
    int32_t num_humans_lo, num_humans_hi;           /* 'num_humans': navsim_regen rewrites n_peds[e] (clipped to
                                                       max_peds); hi = 0: n_peds[e] is kept */
    int32_t outdoor_map_size;         /* navsim_regen: side of an OUTDOOR map in cells when it differs from the arena's
                                         allocation (the reference draws outdoor maps at 400 x 400 and corridor maps at
                                         1000 x 1000, map_generator.py:108-142): the map occupies cells [0, size)^2 of the
                                         map_h x map_w arena, the rest is occupied.  0 = map_w */
    double scan_noise_std_lo, scan_noise_std_hi;    /* 'scan_noise_std': navsim_regen rewrites scan_noise_std[e];
                                                       hi < 0: kept */

    int32_t march_rule;               /* NAVSIM_MARCH_* */
    /* Launch geometry of the fused step.  Validated plain data: the library reads no environment variable. */
    int32_t step_block;               /* threads per arena: 0 = chosen from the batch size,
                                         else 64, 256, 512 or 1024 */
    int32_t ped_split;                /* pedestrian update in its own kernel (a pack of arenas per workgroup) ahead of the
                                         step: 0 or 1 = no, inside the step on one wavefront beside the scan (faster at every
                                         batch size since round 3), 2 = yes (the pack shrinks to what fits 64 KB of LDS: one
                                         arena per workgroup at 64 pedestrians) */
    int32_t regen_check_discomfort;   /* navsim_regen: 1 (default) = a robot start whose FIRST scan (no pedestrians, no noise) has a
                                         beam inside the discomfort zone is dropped and the next start / goal pair of the
                                         spawn table takes its place, like reset() re-draws the robot (env.py:776-781) */
    int32_t rect_lds;                 /* fused step with rect records: 0 = the library stages the index form of an arena's record
                                         table (navsim_state.rect_index) in LDS whenever it is present and fits beside the step's
                                         other LDS at the launch's residency, 1 = never (records from global
                                         memory), 2 = whenever it fits one workgroup per CU */
    int32_t max_waypoints;            /* P: waypoints kept per pedestrian = stride of navsim_state.ped_waypoints, 1 ..
                                         NAVSIM_MAX_WAYPOINTS.  The reference keeps EVERY waypoint of a route
                                         (path_to_waypoints, env.py:1261-1277; env.py:788-804); navsim_default_config
                                         sets 64 = 128 m of route at the 2 m interval (a 1000 x 1000 map is 50 m wide).
                                         A longer route is stored cut after P waypoints, counted in
                                         counters[NAVSIM_COUNTER_ROUTES_CUT], and -- when navsim_state.ped_goal is
                                         present -- continued to the SAME goal by navsim_replan when the pedestrian
                                         reaches the cut */

    /* The action's form (round 4; BUILD-DEFINED like the Husky model itself: the reference drives KetiRobot with
     * (v, omega) only).  NAVSIM_ACTION_WHEELS: io->action holds the angular speeds (left, right) of a skid-steer
     * base's wheel pairs in rad/s, converted on the device as v = r (wl + wr) / 2, omega = r (wr - wl) / track
     * (third_party/husky_description/urdf/husky.urdf.xacro:61-67: track 0.5708, wheel radius 0.1651).
     * clamp_action = 1 clamps the twist to [linvel_lo, linvel_hi] x [rotvel_lo, rotvel_hi] before anything else
     * (the reference only prints a warning and never clips, env.py:606-613: 0 is the default).  The twist after
     * conversion and clamp is what the step integrates and what the observation's `vel` slots carry. */
    int32_t action_kind;              /* NAVSIM_ACTION_* */
    int32_t clamp_action;
    double wheel_radius, wheel_track;
    double linvel_lo, linvel_hi, rotvel_lo, rotvel_hi;   /* __init__.py:12-13 linvel_range, rotvel_range */
    int32_t closed_maps;              /* 1 = the caller asserts that EVERY map of the world has a ring of at least 3 occupied cells
                                         around it (all of the reference's maps have 5, map_generator.py:11, 61-93; every map
                                         navsim_regen draws is closed): set it from navsim_maps_closed's answers.
                                         The LDS form of the march (navsim_state.rect_index) needs it -- a ray of a closed map
                                         never leaves the map, so that form carries no bounds test -- and is not used without.
                                         A wrong assertion reads outside the tables. */
    int32_t defer_reset_scan;         /* 1 = navsim_step leaves the FIRST OBSERVATION of an arena it restarted (auto_reset) to the
                                         navsim_regen call that must follow it: the step does the restart's bookkeeping (pose, goal,
                                         episode, counters, tail of the row) and skips the second scan, navsim_regen produces the
                                         first observation of EVERY arena whose done flag is set -- those it gives a new world and
                                         those beyond regen_cap that restart in place -- in one masked launch over all arenas.
                                         Between the two calls the scan rows of a finished arena's observation are unspecified.
                                         Worth it where a launch is one generation of workgroups and lasts as long as its slowest
                                         arena (c5: the restarted arenas' second scans were 10 of the step's 49 us); the masked
                                         launch costs ~4 ns per arena, so not for thousands of arenas per GPU.  0 = the step scans
                                         itself (default).  Not with navsim_regen_swap. */
    int32_t regen_min_steps;          /* > 0: an arena whose episode ended after FEWER steps than this restarts in place (same map,
                                         next entry of its start / goal table) instead of receiving a new world from navsim_regen /
                                         navsim_regen_swap; counted in counters[NAVSIM_COUNTER_REGEN_SHORT].  A rule of the
                                         simulation's own history (navsim_state.done_steps), whatever the timing: it is what lets
                                         worlds be generated AHEAD of time several steps deep (nav_gym_amd/sim.py enable_pregen
                                         with a pipeline) and still be installed deterministically -- a world staged when an
                                         arena restarts is guaranteed complete regen_min_steps steps later.  0 (default): every
                                         finished arena is eligible.  Needs st->done_steps. */
} navsim_config;

/* navsim_config.auto_reset.  The reference's step() returns the LAST observation of an episode together with done = True
 * (env.py:700-728: after a crash the re-scan at the reverted pose) and leaves reset() to its caller (env.py:730).
 *   NONE       exactly that: nothing restarts; the caller resets (navsim_restart + navsim_reset_obs / navsim_regen).
 *   SAME_STEP  (build-defined vector-env reset, the default of a batch) the step that ends an episode also restarts the arena
 *              -- next start / goal pair of its spawn table, next episode number -- and io->obs holds the FIRST observation of
 *              the new episode; the terminal observation the reference would have returned goes to io->final_obs /
 *              io->final_goals when those are given (rows of the arenas whose done flag this call sets).
 *   NEXT_STEP  (gymnasium's next-step mode) the step that ends an episode returns the terminal observation in io->obs like
 *              the reference, and restarts the arena in the STATE only (pose, goal, episode number, step counter); the arena's
 *              next call must be a reset: the caller passes io->reset_mask = the done flags of the previous call, the arena's
 *              action is ignored, its workgroup produces the first observation (or installs the staged world,
 *              navsim_step_install; or leaves the row to the navsim_regen that follows, cfg.defer_reset_scan) and the call
 *              returns reward 0, done 0, info 0 for it.  A navsim_regen for such arenas takes io->done = that same mask. */
#define NAVSIM_AUTORESET_NONE      0
#define NAVSIM_AUTORESET_SAME_STEP 1
#define NAVSIM_AUTORESET_NEXT_STEP 2

#define NAVSIM_ACTION_TWIST  0   /* io->action = (v, omega): the reference's action (env.py:591) */
#define NAVSIM_ACTION_WHEELS 1   /* io->action = (omega_left, omega_right) in rad/s */

/* ------------------------------------------------------------------------------------------
 * Per-shard simulator state: structure of device arrays, env-major.  E = n_envs, N = max_peds,
 * B = n_beams, S = n_scan_stack, K = n_spawn, P = cfg.max_waypoints.
 * Poses are float64 like the reference's Python floats; scans are float32 like env.py:387.
 * ---------------------------------------------------------------------------------------- */
typedef struct navsim_state {
    /* world */
    const void*   field;            /* distance field of every arena in cfg.field_format
                                       (navsim_build_dt / navsim_build_field) */
    const float*  field_overflow;   /* [E,H,W] float32 distances, read only where the packed field
                                       holds 0xFFFF; may be NULL when navsim_build_field reported
                                       no saturated cell (NAVSIM_FIELD_U16T only) */
    const void*   rect_table;       /* [E, ceil(H/8)*ceil(W/8)] 16-byte two-rectangle records of the 8x8-cell tiles
                                       (navsim_build_rects); optional accelerator of the packed field
                                       (NAVSIM_FIELD_U16T only): a probe in a tile with a valid record gets its
                                       exact integer d2 from the record instead of the field.  NULL = every probe
                                       reads the field.  Results are unchanged.  navsim_regen keeps it current. */
    const double* beam_table;       /* [B,2] cos, sin of the robot-frame beam angles (navsim_beam_table);
                                       optional accelerator, NULL = evaluate every beam direction in full */
    const float*  scan_threshold;   /* [B] env.py:162-170 */
    const float*  scan_discomfort;  /* [B] env.py:172-180 */
    float*        scan_noise_std;   /* [E] env_param['scan_noise_std'] (env.py:439); navsim_regen redraws it per episode
                                       when cfg.scan_noise_std_hi >= 0 */

    /* robot */
    double*  robot_pose;            /* [E,3] px, py, theta in [0, 2pi) */
    double*  robot_goal;            /* [E,2] gx, gy */
    double*  prev_action;           /* [E,2] env.py:725 */
    double*  prev_pose;             /* [E,3] x, y, wrapped yaw of prev_obs (env.py:710-717) */
    int32_t* n_hist;                /* [E] len(prev_obs_queue), 0..S-1 (env.py:257-279) */
    int64_t* episode;               /* [E] episodes finished so far (spawn / RNG counter) */
    int64_t* steps;                 /* [E] steps_since_reset */

    /* pedestrians (ignored when ped_model == NAVSIM_PED_NONE) */
    int32_t* n_peds;                /* [E] live pedestrians, <= N; navsim_regen redraws it per episode when
                                       cfg.num_humans_hi > 0 (env_param['num_humans'], env.py:786) */
    double*  ped_pose;              /* [E,N,3] */
    double*  ped_vel;               /* [E,N,2] world vx, vy (human.py:35-36) */
    double*  ped_prev_yaw;          /* [E,N] wrapped yaw of the pedestrian's previous obs (env.py:245-247) */
    double*  ped_dist;              /* [E,N,3] leg odometry (env.py:255) */
    const double*  ped_v_pref;      /* [E,N] */
    const uint8_t* ped_has_legs;    /* [E,N] */
    double*  ped_waypoints;         /* [E,N,P,2] the waypoints of every pedestrian's current route as its planner stored them
                                       (path_to_waypoints, env.py:1261-1277); entry ped_wp_head is the current local goal.
                                       Written only by whoever plans a route (the caller, navsim_regen, navsim_replan, the
                                       step's table draw): the waypoint pop of env.py:633-642 advances ped_wp_head
                                       instead of shifting the list (ABI 5).  P = cfg.max_waypoints */
    int32_t* ped_n_waypoints;       /* [E,N] >= 1: waypoints STORED for the current route; entry n - 1 is the final one */
    const double*  ped_cmd;         /* [E,N,2] (v, omega) for NAVSIM_PED_EXTERNAL, else NULL */

    /* auto-reset tables */
    const double* spawn_pose;       /* [E,K,3] */
    const double* spawn_goal;       /* [E,K,2] */

    /* planning costmap of every arena, [E, H/5, W/5] uint8 (navsim_costmap), or NULL.  When present a
     * pedestrian that reaches its final waypoint waits for navsim_replan (env.py:667-680) instead of
     * drawing a straight-line goal from the spawn table, and navsim_regen refreshes it (regen_plan). */
    uint8_t* costmap;

    /* Scheduling hints for the fused step, both optional and without effect on any result.  arena_cost [E]:
     * the step writes how long each arena's workgroup lived (ticks of the chip-wide 100 MHz clock).
     * launch_order [E]: a permutation of the arenas; workgroup b then works on arena launch_order[b].
     * navsim_launch_order sorts arenas by descending cost, so that the longest ones start first and the
     * kernel does not end on a few stragglers. */
    uint32_t*      arena_cost;
    const int32_t* launch_order;

    /* TESTS ONLY ("draws supplied"): [E, NAVSIM_DRAWS_PER_ARENA] uniforms in [0, 1) that navsim_regen uses INSTEAD of
     * its hash-keyed ones for the per-episode parameters and the map generators, laid out as NAVSIM_DRAW_* below, so
     * that the reference's own generators can be replayed on the same draws (tests/golden/make_golden.py reset).
     * NULL (always, outside those tests) = hash-keyed draws. */
    const double* regen_draws;

    /* ---- ABI 4 ---- */
    /* [E,N,2] or NULL: the goal of every pedestrian's current route, written by whoever plans it (navsim_regen,
     * navsim_replan; the caller for routes it supplies).  A route was stored cut exactly when its last stored
     * waypoint differs from this goal (path_to_waypoints closes every list with the goal cell's centre, env.py:1273);
     * navsim_replan then plans from where the pedestrian stands to the SAME goal instead of drawing a new one.
     * NULL: a pedestrian that reaches the end of a cut list draws a new goal there. */
    double*  ped_goal;
    /* [NAVSIM_N_COUNTERS] uint64 or NULL: running totals, incremented on the device, zeroed by the caller when it
     * likes.  They make the library's caps observable: no call ever fails or blocks because of a cap. */
    unsigned long long* counters;
    /* [E, navsim_rect_index_bytes(1, H, W)] or NULL: the INDEX form of rect_table (navsim_build_rect_index) -- per arena the
     * distinct rectangles of its records (at most 255, 8 bytes each) and two list indices per 8x8 tile, 10 KB for a
     * 500 x 500 map.  The fused step copies the arena's row into LDS and its scans probe LDS instead of global memory
     * ("map tiles staged through LDS"; cfg.rect_lds).  Needs rect_table (other kernels keep reading the records);
     * navsim_regen keeps it current.  Results are unchanged. */
    void* rect_index;

    /* ---- ABI 5 ---- */
    /* [E,N]: index of the pedestrian's current waypoint in its row of ped_waypoints, 0 <= head < ped_n_waypoints.  The
     * reference pops a reached waypoint off the front of a Python list (env.py:633-642); here the pop is head += 1 and the
     * list stays where its planner put it.  Every planner resets it to 0.  Required with pedestrians. */
    int32_t* ped_wp_head;
    /* [E] or NULL: written by navsim_step for every arena it steps -- bit i set = pedestrian i stands within 0.5 m of its
     * final waypoint after this step's update, i.e. it is due for navsim_replan (env.py:667-680).  When present
     * navsim_replan takes its candidates from here instead of scanning the state itself (every candidate is re-checked
     * against the current state, so flags of pedestrians that were served or regenerated meanwhile are harmless), and
     * navsim_regen clears the word of every arena it gives a new world. */
    unsigned long long* ped_due;
    /* [E] or NULL: the flags the PREVIOUS step wrote (the caller alternates two buffers between ped_due and ped_due_prev).
     * Read by navsim_step_part only: it splits a step into the arenas with and without a pedestrian waiting for
     * navsim_replan, so that the re-plan of step t runs beside step t + 1 of all the other arenas. */
    const unsigned long long* ped_due_prev;
    /* [E] or NULL: written by navsim_step when an arena finishes (cfg.auto_reset): the number of steps its episode lasted.
     * Read by navsim_regen / navsim_regen_swap for cfg.regen_min_steps. */
    int32_t* done_steps;
    /* [E] or NULL: the slot of the five per-map arrays (field, field_overflow, rect_table, rect_index, costmap) that holds
     * arena e's map; NULL = slot e.  Those arrays then have max(map_slot) + 1 entries.  For navsim_step_install without map
     * copies: the live and the staged state point at the SAME five arrays of 2 E slots, each with its own table (live
     * 0 .. E-1, staged E .. 2E-1 to begin with), and an install exchanges the two states' entries for the arena.  Every
     * entry point that reads or writes a map goes through the table (the step through the instantiations navsim_step_install
     * uses, whatever the call: packed fields only, and navsim_step_part / navsim_step_replan return NAVSIM_E_UNSUPPORTED);
     * not with cfg.shared_field. */
    int32_t* map_slot;
} navsim_state;

#define NAVSIM_N_COUNTERS              8
#define NAVSIM_COUNTER_REGEN_SERVED    0   /* arenas given a new world by navsim_regen / installed by navsim_regen_swap */
#define NAVSIM_COUNTER_REGEN_UNSERVED  1   /* finished arenas a navsim_regen / navsim_regen_swap call left beyond
                                              cfg.regen_cap (or not staged yet): they play their next episode in place */
#define NAVSIM_COUNTER_REPLAN_SERVED   2   /* pedestrians navsim_replan planned for (whether or not a path was found) */
#define NAVSIM_COUNTER_REPLAN_UNSERVED 3   /* pedestrians due for a re-plan beyond max_queries: counted in every call that
                                              leaves them waiting */
#define NAVSIM_COUNTER_ROUTES_CUT      4   /* routes longer than cfg.max_waypoints, stored cut (navsim_regen with
                                              regen_plan, navsim_replan) */
#define NAVSIM_COUNTER_ROUTES_RESUMED  5   /* cut routes navsim_replan continued to their own goal (ped_goal) */
#define NAVSIM_COUNTER_REGEN_SHORT     6   /* finished arenas whose episode was shorter than cfg.regen_min_steps: restarted in place */
#define NAVSIM_COUNTER_REGEN_LATE      7   /* navsim_regen_swap (pipelined): eligible arenas whose staged world was not complete --
                                              the caller's pipeline is deeper than cfg.regen_min_steps covers; they restart in
                                              place and this stays 0 when the caller keeps the documented order */

#define NAVSIM_DRAWS_PER_ARENA      464
#define NAVSIM_DRAW_KIND            0   /* np.random.random() < indoor_ratio             (env.py:295) */
#define NAVSIM_DRAW_OBSTACLE_NUMBER 1   /* env_param 'obstacle_number', 'int'            (env.py:281-292) */
#define NAVSIM_DRAW_NUM_HUMANS      2   /* env_param 'num_humans', 'int' */
#define NAVSIM_DRAW_SCAN_NOISE_STD  3   /* env_param 'scan_noise_std', 'float' */
#define NAVSIM_DRAW_OBSTACLE_WIDTH  4   /* env_param 'obstacle_width', 'float' */
#define NAVSIM_DRAW_CORRIDOR_WIDTH  5   /* env_param 'corridor_width', 'int' */
#define NAVSIM_DRAW_ITERATIONS      6   /* env_param 'iterations', 'int' */
#define NAVSIM_DRAW_MAP             8   /* the generator's own draws in its own order: create_outdoor_map (x, y) per
                                           obstacle (map_generator.py:129-131); create_indoor_map (x, y, coin) per
                                           iteration (map_generator.py:101-106, 61) */

/* What step() returns (env.py:728) with a leading env axis. */
typedef struct navsim_step_io {
    const double* action;           /* [E,2] (linvel, rotvel) -- not clipped (env.py:611-613) */
    const float*  obs_prev;         /* [E,S*B+7] observation returned by the previous step/reset */
    float*   obs;                   /* [E,S*B+7] scan stack, prev_pose, pose, vel, yaw (env.py:455) */
    float*   achieved_goal;         /* [E,2] */
    float*   desired_goal;          /* [E,2] */
    double*  reward;                /* [E] */
    uint8_t* done;                  /* [E] */
    float*   is_success;            /* [E] info['is_success'] (env.py:475) */
    float*   is_crash;              /* [E] info['is_crash']   (env.py:476) */
    double*  distance;              /* [E] info['distance']   (env.py:474) */
    /* ---- ABI 6 ---- */
    /* [E, S*B+7] or NULL: the TERMINAL observation of every arena whose done flag this call sets under
     * NAVSIM_AUTORESET_SAME_STEP -- what the reference's step() returns with done = True (env.py:700-728): scan stack, tail
     * and, after a crash, the re-scan at the reverted pose -- written BEFORE the arena restarts; rows of other arenas are not
     * touched.  (NONE / NEXT_STEP: io->obs itself is that row; nothing is written here.) */
    float*   final_obs;
    /* [E,4] or NULL: achieved_goal (2) and desired_goal (2) of that terminal observation (env.py:455-461) */
    float*   final_goals;
    /* [E] uint8 or NULL: arenas to RESET instead of stepping (NAVSIM_AUTORESET_NEXT_STEP: the done flags of the previous
     * call; their state was restarted when they finished).  Must not alias io->done. */
    const uint8_t* reset_mask;
} navsim_step_io;

/* ---- library ---------------------------------------------------------------------------- */
int         navsim_abi_version(void);
const char* navsim_error_string(int code);
/* Fills a config with the registered NavGym-v0 defaults (__init__.py:6-38, keti_robot.py:44-48). */
int         navsim_default_config(navsim_config* cfg);
/* hipGetErrorString of the launch failure behind the last NAVSIM_E_LAUNCH on the calling thread */
const char* navsim_last_hip_error(void);

/* ---- a3: range_libc.PyOMap + PyRayMarching.__init__ (env.py:337-340) -------------------- */
/* Exact Euclidean distance transform of `occ` (nonzero = occupied), float32 cells.
 * workspace: navsim_build_dt_workspace_bytes() bytes of device scratch. */
size_t navsim_build_dt_workspace_bytes(int32_t n_maps, int32_t map_h, int32_t map_w);
int    navsim_build_dt(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w,
                       float* field, void* workspace, size_t workspace_bytes, void* stream);

/* Same transform, written in `format` (the layout the fused step streams).  `field` needs
 * navsim_field_bytes() bytes.  `overflow` (float32 [n,H,W], may be NULL) receives the exact float
 * distance of every cell; `n_saturated` (device int32, may be NULL) is incremented once per cell whose
 * squared distance does not fit the packed format -- when it stays 0 the overflow plane is never read
 * and may be dropped. */
size_t navsim_field_bytes(int32_t n_maps, int32_t map_h, int32_t map_w, int32_t format);
int    navsim_build_field(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, int32_t format,
                          void* field, float* overflow, int32_t* n_saturated,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Two-rectangle records of the distance field's 8x8-cell tiles (navsim_state.rect_table).  Obstacles of the
 * reference's maps are unions of axis-aligned rectangles of cells (map_generator.py:97-143), and the squared
 * distance of a cell to such a rectangle is max(x0-px, px-x1, 0)^2 + max(y0-py, py-y1, 0)^2.  A record holds two
 * rectangles of occupied cells (4 x int16 each: x0, y0, x1, y1) with d2(cell) = min over the two for EVERY in-map
 * cell of the tile -- the builder checks that cell by cell against `field` and marks the tile invalid (x0 of the
 * first rectangle = 0x7FFF) when no such pair is found.  Any map is accepted; maps that are not rectangle unions
 * just get fewer valid records.  `field` / `format` / `overflow` as produced by navsim_build_field for the same
 * occupancy grids (overflow may be NULL: tiles holding a saturated cell are then invalid).  H, W <= 1024. */
size_t navsim_rect_table_bytes(int32_t n_maps, int32_t map_h, int32_t map_w);
/* The index form of `table` (navsim_state.rect_index): per map the distinct rectangles of its valid records and two list
 * indices per tile.  n_rects [n_maps] int32 or NULL receives the number of distinct rectangles found (tiles naming one
 * beyond the 255th get no index and are read from the field). */
size_t navsim_rect_index_bytes(int32_t n_maps, int32_t map_h, int32_t map_w);
int    navsim_build_rect_index(const void* table, int32_t n_maps, int32_t map_h, int32_t map_w, void* index,
                               int32_t* n_rects, void* stream);
/* closed [n_maps] int32: 1 where every cell of the map's outer ring of 3 cells is occupied (navsim_config.closed_maps may
 * be set when all are 1). */
int    navsim_maps_closed(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, int32_t* closed, void* stream);
/* The same test on a WORLD: *n_open (device int32, zeroed by the caller) receives the number of arenas of st->field whose
 * outer ring of 3 cells has a free cell (a cell is occupied exactly when its distance is 0).  cfg->closed_maps is an
 * assertion the march of the LDS form relies on without testing it: a caller who assembles a world checks it with this
 * (nav_gym_amd/sim.py NavSim does, once, at construction); every map navsim_regen draws is closed by construction. */
int    navsim_world_closed(const navsim_config* cfg, const navsim_state* st, int32_t* n_open, void* stream);
size_t navsim_build_rects_workspace_bytes(int32_t n_maps, int32_t map_h, int32_t map_w);
int    navsim_build_rects(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, const void* field,
                          int32_t format, const float* overflow, void* table, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ---- a4: PyRayMarching.calc_range_many (env.py:425) ------------------------------------- */
/* queries [E, n_per_env, 3] float32 (x, y, theta) in cell units, out [E, n_per_env] in cells;
 * march_rule = NAVSIM_MARCH_*. */
int navsim_cast_static(const float* field, int32_t n_envs, int32_t map_h, int32_t map_w,
                       const float* queries, int32_t n_per_env, float max_range, int32_t march_rule,
                       float* out, void* stream);

/* ---- a5: CMap2D.flatten_contours + render_contours_in_lidar (env.py:430-431) ------------ */
/* ranges [E,B] in/out (metres); angles [E,B] float64; verts [E,V,3] float32 (contour id, x, y),
 * n_verts [E] live vertices per env (<= V); origin [E,2] float32. */
int navsim_render_polys(float* ranges, const double* angles, int32_t n_envs, int32_t n_beams,
                        const float* verts, const int32_t* n_verts, int32_t max_verts,
                        const float* origin, void* stream);

/* ---- a6: CSimAgent + CMap2D.render_agents_in_lidar (env.py:402, 432) -------------------- */
/* agents [E,A,8] float32: pos(3) dist(3) vel(2); n_agents [E] (<= A). */
int navsim_render_legs(float* ranges, const double* angles, int32_t n_envs, int32_t n_beams,
                       const float* agents, const int32_t* n_agents, int32_t max_agents,
                       const float* origin, void* stream);

/* ---- a8 / a9: Human.set_vel (human.py:32-41), KetiRobot.set_vel (keti_robot.py:64-93) --- */
/* pose [n,3] in/out, cmd [n,2] (v, omega), vel_out [n,2] world (vx, vy) or NULL.
 * axle_offset = 0 gives Human.set_vel, 0.14474 gives KetiRobot.set_vel. */
int navsim_integrate(double* pose, const double* cmd, double* vel_out, int32_t n,
                     double time_step, double axle_offset, void* stream);

/* ---- a12 / a13: compute_rewards, compute_terminals, compute_info (env.py:464-589) ------- */
/* HER batch API.  obs [n, S*B+7] float32 or float64 (obs_is_f64), goals [n,2] same dtype.
 * Any output pointer may be NULL. */
int navsim_reward_done(const navsim_config* cfg, const void* obs, const void* goals,
                       int32_t obs_is_f64, int32_t n,
                       const float* scan_threshold, const float* scan_discomfort,
                       double* reward, uint8_t* done, float* is_success, float* is_crash,
                       double* distance, void* stream);

/* ---- a14: _make_scan_threshold / _make_scan_discomfort_threshold (env.py:162-180) ------- */
/* footprint [n_vert,2] float32 (un-closed polygon in the robot frame), out [B]. */
int navsim_scan_threshold(const navsim_config* cfg, const float* footprint, int32_t n_vert,
                          float* out, void* stream);

/* cos / sin (float64, DESIGN.md section 4 functions) of np.linspace(angle_min, angle_last, B): lets the
 * step derive each beam direction by one angle addition and prove, per beam, that the float32
 * rounding equals the full evaluation's (falling back to it otherwise) -- results are unchanged. */
int navsim_beam_table(const navsim_config* cfg, double* table, void* stream);

/* ---- a1: NavGymEnv.step (env.py:591-728), fused ----------------------------------------- */
/* One launch: pedestrian update, robot integration, scan, reward/done/info, crash revert (or
 * respawn) with re-scan, observation packing.  State is updated in place. */
int navsim_step(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                void* stream);

/* navsim_step for a PART of the arenas (ABI 5), chosen by the flags the previous step left in st->ped_due_prev:
 * NAVSIM_STEP_NOT_DUE steps the arenas none of whose pedestrians waits for navsim_replan, NAVSIM_STEP_DUE the others;
 * the two calls together are exactly one navsim_step (arenas are independent; every arena is stepped by one of them),
 * NAVSIM_STEP_ALL is navsim_step.  What it is for: navsim_replan is a chain of dependent breadth-first levels that serves
 * ~2 % of the arenas per step and used to sit serially behind every step (env.py:667-680 plans inside step()); with
 *     stream A:  navsim_step_part(NOT_DUE)                                    } step t + 1
 *     stream B:  navsim_replan (of step t) -> navsim_step_part(DUE)            }
 * the re-plan runs beside the step of the arenas that do not need it (nav_gym_amd/sim.py NavSim.step_overlapped).
 * Both calls use the same io (rows of arenas a call does not step are not touched) and the same two flag buffers. */
#define NAVSIM_STEP_ALL     0
#define NAVSIM_STEP_NOT_DUE 1
#define NAVSIM_STEP_DUE     2
int navsim_step_part(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io, int32_t part,
                     void* stream);

/* ---- env.py:685-693: the scan of every pedestrian (input of the reference's HumanPolicy) --------- */
/* out [E, N, ped_n_beams] float32 metres: static map from the pedestrian's integer cell, the robot as its
 * threshold footprint and the other pedestrians as footprint rectangles (lidar_legs=False), clipped to
 * ped_range_max, no noise.  Uses the CURRENT state (call it after navsim_step / navsim_reset_obs).
 * Rows of pedestrians >= n_peds[e] are left untouched. */
int navsim_ped_scans(const navsim_config* cfg, const navsim_state* st, float* out, void* stream);
/* The same for the arenas [e0, e0 + n_e) only (rows of the other arenas are not touched).  With navsim_ped_policy_part it
 * lets a caller pipeline the two: the scans of the next slice of arenas -- a latency-bound march -- run on one stream beside
 * the network of the current slice -- FMA / MFMA-bound -- on another (nav_gym_amd/sim.py NavSim.ped_policy, round 5). */
int navsim_ped_scans_part(const navsim_config* cfg, const navsim_state* st, float* out, int32_t e0, int32_t n_e, void* stream);

/* ---- reset() of finished arenas on the device, with a new random map ------------ */
/* For every arena with done[e] != 0 (at most cfg.regen_cap, lowest indices first): a fresh outdoor map
 * (create_outdoor_map, map_generator.py:126-143, counter-based RNG keyed by seed / global arena /
 * episode), its distance field, a new table of start / goal pairs (clearance and distance rules of
 * env.py:748-783 without the A* path test), the robot placed on one of them, every pedestrian re-drawn
 * (start >= 4 m from the robot, goal >= 10 m away, v_pref, has_legs: env.py:786-806) and the first
 * observation of the new episode (env.py:808-831) written to io->obs.  Call it right after navsim_step on
 * the same stream with the same io.  Mutates field, spawn tables and pedestrian parameters in place.
 * Square maps; FIELD_F32, or FIELD_U16T.  A packed world of at most 520 cells per side has no overflow plane (no
 * cell of such a map can be 256 cells from every obstacle) and must not be given one; a larger packed world MUST
 * carry st->field_overflow, which is regenerated together with the field (NAVSIM_E_UNSUPPORTED otherwise).
 * cfg->regen_plan = 1: starts and goals are centres of free COSTMAP cells and a pair is kept only when the
 * planner joins it (robot: path no longer than twice the straight line, env.py:756-762; pedestrians get the
 * path's waypoints every 2 m, env.py:804); four rounds of candidates, the last one stays if none passes.
 * Finished arenas beyond cfg.regen_cap keep their map and play their next episode in place (the step has already
 * respawned them from the spawn table); counters[NAVSIM_COUNTER_REGEN_UNSERVED] counts them. */
size_t navsim_regen_workspace_bytes(const navsim_config* cfg);
/* navsim_regen off the step's critical path (round 3).  The world an arena receives at the end of an episode depends
 * on (cfg.seed, global arena index, episode number) only, so it can be generated AHEAD of time: keep a second, STAGED
 * navsim_state (own copies of every array navsim_regen writes, its episode[e] = live episode[e] + 1) and a staged
 * observation buffer; run the ordinary navsim_regen on the staged state for the arenas flagged in want[] (as its
 * io->done), on a side stream, while steps run.  After a step, navsim_regen_swap installs the staged world of every
 * finished arena (io->done[e] != 0; at most cfg.regen_cap, lowest indices first -- navsim_regen's own rule): it copies
 * the arena's rows of those arrays and of the observation from the staged state into the live one, marks the arena for
 * the next staging pass and sets stage->episode[e] = live episode + 1.  The live state is then exactly what navsim_regen would have left.
 * The caller orders the streams: a swap after the staging pass that served its arenas, the next staging pass after the
 * swap (nav_gym_amd/sim.py NavSim.enable_pregen).  Needs cfg.auto_reset = 1 (the step advances episode[e] at `done`).
 * A finished arena that is NOT installed by a call -- beyond regen_cap, or its staged world not ready (want[e] != 0) --
 * plays on in place like one beyond navsim_regen's cap and is staged again for its new episode number; with no such
 * surplus the live state after every call equals navsim_regen's bit for bit.
 * want [E], mark [E]: uint8 flags, zero-initialised by the caller with every arena staged (or all ones in want and one
 * navsim_regen_stage per regen_cap arenas first).  navsim_regen_swap only reads want and writes mark;
 * navsim_regen_stage (io->done must be `want`, io->obs the staged observation buffer) merges mark into want, runs
 * navsim_regen on the staged state and clears want for the arenas it served. */
int    navsim_regen_swap(const navsim_config* cfg, const navsim_state* live, const navsim_state* stage,
                         const navsim_step_io* io, const float* stage_obs, const uint8_t* want, uint8_t* mark,
                         const long long* ready, void* stream);
int    navsim_regen_stage(const navsim_config* cfg, const navsim_state* stage, const navsim_step_io* io, uint8_t* want,
                          uint8_t* mark, long long* ready, void* workspace, size_t workspace_bytes, void* stream);
/* navsim_regen_stage for a SHARE of the arenas (round 6): part p of n takes the marks of the 32-bit words w of mark[] with
 * w % n == p -- arenas in groups of four -- into ITS OWN want[] (one array per part; io->done = that array) and stages them.  Parts
 * own disjoint arenas: passes of different parts may run at the same time on different streams, each with its own workspace
 * (the launches of a staging pass for corridor maps with planned starts are latency-bound chains that leave the chip idle:
 * NavSim alternates two parts).  navsim_regen_stage(...) = part 0 of 1. */
int    navsim_regen_stage_part(const navsim_config* cfg, const navsim_state* stage, const navsim_step_io* io, uint8_t* want,
                               uint8_t* mark, long long* ready, void* workspace, size_t workspace_bytes, int32_t part,
                               int32_t n_parts, void* stream);
/* Round 5, the PIPELINED form (ready != NULL, cfg.regen_min_steps > 0): staging passes need not finish before the next swap.
 * ready [2 E] int64 (the second half is the passes' scratch; initialise [0, E) with the staged episode numbers once every arena
 * is staged): navsim_regen_stage records for every arena it served the episode number it staged; navsim_regen_swap installs a finished arena's staged world iff its episode lasted at least cfg.regen_min_steps
 * steps (live->done_steps) AND ready[e] equals the episode the arena now starts -- the first is a rule of the simulation (the
 * oracle's navsim_regen_cpu applies it too), the second a safety net that only fails when the caller queued its staging
 * passes too late (counters[NAVSIM_COUNTER_REGEN_LATE]); want[] is then not consulted.  mark [E rounded up to 4] uint8,
 * 4-byte aligned: set and consumed with 32-bit atomics, so swaps may run while a pass merges it.  The caller's order
 * (NavSim.enable_pregen(pipeline=P)): a staging pass every P steps on a side stream behind an event of that step; before the
 * swap of step k P, wait for the pass queued at step (k - 2) P; cfg.regen_min_steps >= 4 P. */
/* navsim_step + the pipelined navsim_regen_swap in ONE launch: an arena that finishes in this step, whose episode lasted
 * cfg.regen_min_steps steps and whose staged world is ready (ready[e] == the episode that starts), takes that world -- its own
 * workgroup copies the staged rows in place of the restart's second scan -- and asks for the next one (mark, stage->episode);
 * every other finished arena restarts in place exactly as navsim_step leaves it and asks for a world with its new episode
 * number.  Result = navsim_step followed by navsim_regen with regen_cap >= n_envs (every finished arena decides alone: the
 * call needs cfg.regen_cap >= cfg.n_envs; the staging passes' cap is the one of the config THEY are given).  Counters as
 * navsim_regen_swap.  Packed fields, pedestrians inside the step (ped_split 0 / 1); NAVSIM_E_UNSUPPORTED otherwise.
 * The caller queues the staging passes as for the pipelined swap (replace "swap of step" by "step"). */
/* NAVSIM_AUTORESET_NEXT_STEP with late = NULL and cfg.regen_min_steps = 0 (ABI 6): NO fallback call is needed -- an arena that is
 * reset by this call (io->reset_mask) and finds no world staged regenerates its own world inside the launch, its workgroup
 * running navsim_regen's device functions for that one arena in place of the step it does not take (rare, and slow for that
 * one workgroup; the result is navsim_regen's bit for bit).  Worlds of outdoor maps (cfg.regen_indoor_ratio = 0) without
 * cfg.regen_plan and without a costmap, at least 256 threads per arena, cfg.march_rule = NAVSIM_MARCH_F32 (the default);
 * NAVSIM_E_UNSUPPORTED otherwise (pass `late` then). */
/* late [E] uint8 or NULL -- the FALLBACK that needs no rule: the launch writes late[e] = 1 for every arena that finished,
 * is due a new world (by cfg.regen_min_steps, if set) and whose staged world was not ready, 0 for every other arena; the caller
 * follows the launch with navsim_regen(cfg', st, io' with io'->done = late) on the same stream (cfg' = cfg with a regen_cap that
 * sizes the workspace; with no arena late the call's launches find nothing to do).  The result is then navsim_step +
 * navsim_regen whatever the staging passes' timing -- also with cfg.regen_min_steps = 0, i.e. the reference's "a new map at
 * every reset()" unchanged; the passes only decide how many arenas take the fast path.  late = NULL needs
 * cfg.regen_min_steps >= 1 and the caller's order of passes and waits (above). */
int    navsim_step_install(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                           const navsim_state* stage, const float* stage_obs, uint8_t* mark, const long long* ready,
                           uint8_t* late, void* stream);
/* ... with navsim_replan(cfg, st, max_queries) of the PREVIOUS step's flags inside the same launch (navsim_step_replan's form and
 * conditions: costmaps of up to one word per thread of the arena's workgroup, else NAVSIM_E_UNSUPPORTED -- the caller then
 * runs navsim_replan behind navsim_step_install).  max_queries < 0: navsim_step_install. */
int    navsim_step_install_replan(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                                  const navsim_state* stage, const float* stage_obs, uint8_t* mark, const long long* ready,
                                  uint8_t* late, int32_t max_queries, void* stream);
/* NAVSIM_AUTORESET_NEXT_STEP (ABI 6): navsim_step_install[_replan] where an arena that finds no world staged is regenerated
 * BESIDE the next launch instead of behind this one.  An arena that finishes in this call learns at once whether the world of
 * the episode it will start is staged (ready[e]); if not, late_next[e] = 1 (0 for every other arena).  The next call receives
 * those flags as late_prev: it only zeroes the outputs of the arenas flagged there, while the caller's
 *     navsim_regen(cfg' = cfg with defer_reset_scan = 1 and a regen_cap of its own, st, io' with io'->done = late_prev)
 * runs on ANOTHER stream at the same time (both behind the call that wrote the flags; the caller joins the two streams
 * before its next call).  The arenas not flagged install their staged worlds at the front of the launch.  Same rollout as
 * navsim_step + navsim_regen keyed on io->reset_mask, whatever the staging passes' timing -- with no rule (cfg.regen_min_steps
 * = 0) and with nothing of the reset path on the steps' critical path.  late_next, late_prev: [E] uint8, two buffers the
 * caller alternates; max_queries < 0: no re-plan inside the launch. */
int    navsim_step_install_next(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                                const navsim_state* stage, const float* stage_obs, uint8_t* mark, const long long* ready,
                                uint8_t* late_next, const uint8_t* late_prev, int32_t max_queries, void* stream);
/* navsim_regen's helper stream.  Worlds of corridor maps with planned starts (cfg.regen_indoor_ratio > 0, cfg.regen_plan): the
 * distance transform of the new maps runs on a second stream between two events on `stream` (inside a hipGraph capture of
 * `stream` it joins and leaves the capture through them); environment NAVSIM_REGEN_FORK=0 keeps the call on `stream` alone.
 * navsim_regen_helper(s): the calling host thread's helper stream from now on (NULL / never called: one of the library's own,
 * created by navsim_prepare or the first such call).  Why a caller would choose: HIP spreads a process's streams over a few
 * hardware queues in creation order, and two streams on the same queue take turns -- NavSim times candidates against the
 * streams the helper is to run beside (nav_gym_amd/sim.py concurrent_stream).  A helper equal to the stream of a navsim_regen
 * call means "this call does not fork" (the per-step fallback of the pipelined reset path: its helper belongs to the passes). */
int    navsim_regen_helper(void* stream);
int    navsim_regen(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- reset path: costmap (env.py:312-332), pyastar2d.astar_path (env.py:343-354),
 *      path_to_waypoints (env.py:1261-1277) ---------------------------------------------------- */
/* cost [n, H/5, W/5] uint8 (1 = blocked): 5x nearest subsampling + 9x9 dilation (reflect-101 border). */
int navsim_costmap(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, uint8_t* cost, void* stream);
/* n queries; query q plans on costmap map_index[q] (q when NULL).  Shortest 4-connected path between the
 * cells of start[q] and goal[q] (build-defined tie-break), cut into waypoints every
 * `interval` metres.  wp [n,max_wp,2], n_wp [n] (0 = no path), path_cells [n], path_len [n] (env.py:757-759);
 * the last three may be NULL.  The search runs in LDS (4 bytes per costmap cell: up to 200 x 200 cells,
 * i.e. 1000 x 1000 maps; larger costmaps return NAVSIM_E_UNSUPPORTED). */
int    navsim_plan(const uint8_t* cost, const int32_t* map_index, int32_t n_queries, int32_t cost_h, int32_t cost_w,
                   double cost_resolution, double origin_x, double origin_y, const double* start, const double* goal,
                   double interval, int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len,
                   void* stream);

/* env.py:667-680: every pedestrian within 0.5 m of its final waypoint gets a new goal -- a free costmap cell
 * more than cfg->ped_min_goal_dist away (up to 4 rounds of 16 draws) -- and the waypoints of the shortest
 * path to it every 2 m; it keeps its old waypoint when no round finds a path ("only if the human is not
 * adjacent to a wall").  At most max_queries pedestrians per call, in (arena, pedestrian) order; the rest
 * are served by a later call and counted in counters[NAVSIM_COUNTER_REPLAN_UNSERVED].  With st->ped_due present the
 * candidates are the pedestrians flagged there by the last navsim_step (the same set, found without a pass over the
 * state).  A pedestrian standing at
 * the end of a CUT route (st->ped_goal present and different from its last stored waypoint) is first planned to
 * that same goal; only if no path joins them does it draw a new goal like the others.
 * Needs st->costmap.  Call after navsim_step / navsim_regen on the same stream. */
size_t navsim_replan_workspace_bytes(const navsim_config* cfg, int32_t max_queries);
int    navsim_replan(const navsim_config* cfg, const navsim_state* st, int32_t max_queries, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- pedestrian control block (env.py:617-662) with the reference's HumanPolicy actor --------------
 * Weights of human_policy.py:24-30 in torch's own layouts (state_dict order). */
typedef struct navsim_policy_weights {
    const float* cv1_w;   /* [32,3,5]    act_fea_cv1.weight (Conv1d 3->32, k5, s2, p1) */
    const float* cv1_b;   /* [32] */
    const float* cv2_w;   /* [32,32,3]   act_fea_cv2.weight (Conv1d 32->32, k3, s2, p1) */
    const float* cv2_b;   /* [32] */
    const float* fc1_w;   /* [256,4096]  act_fc1.weight */
    const float* fc1_b;   /* [256] */
    const float* fc2_w;   /* [128,260]   act_fc2.weight */
    const float* fc2_b;   /* [128] */
    const float* a1_w;    /* [128]       actor1.weight */
    const float* a1_b;    /* [1] */
    const float* a2_w;    /* [128]       actor2.weight */
    const float* a2_b;    /* [1] */
} navsim_policy_weights;

/* For every live pedestrian: pop waypoints closer than 1 m (env.py:633-640), local goal in the body frame
 * (env.py:644-645), network input = its latest scan clipped to [0, 6], / 6 - 0.5, the same in all three
 * channels (env.py:629-630, 647), speed input = prev_actions; HumanPolicy actor forward
 * (human_policy.py:43-52); prev_actions <- clip(mean, [0,-1], [1,1]) (env.py:655-658);
 * ped_cmd <- prev_actions * v_pref (env.py:659-662), ready for navsim_step with NAVSIM_PED_EXTERNAL.
 * ped_scans [E,N,512] float32 is the output of navsim_ped_scans on the current state (cfg.ped_n_beams must
 * be 512); prev_actions [E,N,2] float32 in/out (zeros at reset, env.py:739); ped_cmd [E,N,2] float64 out.
 * Arithmetic: every dot product is a float32 fused-multiply-add chain in index order
 * starting from 0, bias added last -- what v_mfma_f32_32x32x2_f32 computes -- so device and oracle agree bit
 * for bit; against torch's kernels the mean differs by ~1e-6. */
size_t navsim_ped_policy_workspace_bytes(const navsim_config* cfg);
int    navsim_ped_policy(const navsim_config* cfg, const navsim_state* st, const navsim_policy_weights* w,
                         const float* ped_scans, float* prev_actions, double* ped_cmd, void* workspace,
                         size_t workspace_bytes, void* stream);
/* navsim_ped_policy for the pedestrians of the arenas [e0, e0 + n_e) only; same arrays, same workspace (calls on one workspace
 * must be ordered on one stream). */
int    navsim_ped_policy_part(const navsim_config* cfg, const navsim_state* st, const navsim_policy_weights* w,
                              const float* ped_scans, float* prev_actions, double* ped_cmd, void* workspace,
                              size_t workspace_bytes, int32_t e0, int32_t n_e, void* stream);
/* navsim_ped_scans + navsim_ped_policy in one pass over the pedestrians (round 4): every pedestrian's scan is taken from the
 * CURRENT state by the workgroup that convolves it (env.py:685-693 -> 629-630, 647 -> human_policy.py:38-42), stays in LDS and
 * never travels through HBM; the march of one pedestrian runs beside the convolutions of others.  scans_out [E,N,512] float32
 * or NULL: the clipped scans, exactly what navsim_ped_scans writes (rows of dead slots untouched).  prev_actions, ped_cmd,
 * workspace (navsim_ped_policy_workspace_bytes) as for navsim_ped_policy; results bit-identical to the two calls. */
int    navsim_ped_scan_policy(const navsim_config* cfg, const navsim_state* st, const navsim_policy_weights* w,
                              float* scans_out, float* prev_actions, double* ped_cmd, void* workspace,
                              size_t workspace_bytes, void* stream);

/* order[0..n) = the arenas sorted by descending cost (ties in unspecified order): longest-first launch order
 * for navsim_step (navsim_state.launch_order).  cost is what the step wrote to navsim_state.arena_cost. */
int navsim_launch_order(const uint32_t* cost, int32_t* order, int32_t n, void* stream);

/* ---- CrowdSim-v0: collision tests, goal test and reward / info selection of CrowdSim.step
 *      (nav_gym/src/crowd_sim/envs/crowd_sim.py:808-945, phase != 'test', border = None) -------------- */
typedef struct navsim_crowd_params {
    double time_step, discomfort_dist, map_size_m, map_resolution;
    double success_reward, collision_penalty, discomfort_penalty_factor, rotation_penalty_factor, timeout_penalty;
    double time_limit;
} navsim_crowd_params;
#define NAVSIM_CROWD_NOTHING 0
#define NAVSIM_CROWD_TIMEOUT 1
#define NAVSIM_CROWD_REACH_GOAL 2
#define NAVSIM_CROWD_COLLISION 3
#define NAVSIM_CROWD_COLLISION_OTHER 4
#define NAVSIM_CROWD_DANGER 5
/* Per env: closest approach of every agent to the robot over the step (point_to_segment_dist,
 * crowd_sim/envs/utils/utils.py:4-26, minus both radii: crowd_sim.py:808-826), the occupancy-grid windows
 * around the robot's next position (collision: half-width ceil(r / sqrt 2 / res) cells, crowd_sim.py:831-861;
 * discomfort: ceil((r + discomfort_dist) / res), crowd_sim.py:872-896), the goal test (crowd_sim.py:909-915)
 * and the reward / done / info cascade (crowd_sim.py:917-949).
 * free_map [E,G,G] uint8, 1 = free, indexed [x][y] like CrowdSim.map; robot [E,10] = px, py, next_px, next_py,
 * next_vx, next_vy, gx, gy, radius, action.r; agents [E,A,5] = px, py, vx, vy, radius; n_agents [E] or NULL
 * (= A); global_time [E].  Outputs reward [E], done [E], info [E] (NAVSIM_CROWD_*), min_dist [E] (Danger). */
int navsim_crowd_check(const navsim_crowd_params* p, int32_t n_envs, int32_t max_agents, int32_t grid,
                       const uint8_t* free_map, const double* robot, const double* agents, const int32_t* n_agents,
                       const double* global_time, double* reward, uint8_t* done, int32_t* info, double* min_dist,
                       void* stream);

/* ---- CrowdSim-v0 local maps (crowd_sim.py:999-1186) ------------------------------------------------ */
#define NAVSIM_CROWD_MAX_VERTS 8
typedef struct navsim_crowd_map_params {
    double angular_min, angular_max;    /* angular_map_min_angle / max_angle (config angle_min / angle_max x pi) */
    double angular_max_range;           /* angular_map_max_range */
    int32_t angular_dim;                /* angular_map_dim */
    int32_t normalize;                  /* get_local_map_angular(normalize=True) */
    double map_size_m, map_resolution, submap_size_m;     /* get_local_map */
} navsim_crowd_map_params;
/* get_local_map_angular (crowd_sim.py:1055-1102) with calculate_angular_map_distances (crowd_sim.py:999-1053): per
 * env the distance to the closest obstacle outline in each of angular_dim sectors of the robot frame, float64.
 * robot [E,4] = px, py, theta, radius; verts [E, max_obst, n_vert, 2] (n_vert <= NAVSIM_CROWD_MAX_VERTS, the
 * reference's obstacles have 4: crowd_sim.py:250-258); n_obst [E] or NULL (= max_obst); out [E, angular_dim]. */
int navsim_crowd_angular_map(const navsim_crowd_map_params* p, int32_t n_envs, int32_t max_obst, int32_t n_vert,
                             const double* robot, const double* verts, const int32_t* n_obst, double* out, void* stream);
/* get_local_map (crowd_sim.py:1104-1166): the binary submap of submap_size_m around the robot, rotated into its
 * heading (rotate_grid_around_center, crowd_sim.py:1168-1186; rotate = 0 skips the rotation) and thresholded at
 * 0.9.  free_map [E,G,G] uint8 (1 = free) indexed [x][y] like CrowdSim.map; robot as above; out [E,S,S] uint8 with
 * S = round(submap_size_m / map_resolution). */
int navsim_crowd_local_map(const navsim_crowd_map_params* p, int32_t n_envs, int32_t grid, const uint8_t* free_map,
                           const double* robot, int32_t rotate, uint8_t* out, void* stream);

/* ---- CrowdSim-v0 pedestrians: ORCA through rvo2 (crowd_sim/envs/policy/orca.py:85-135) and Agent.step
 *      (crowd_sim/envs/utils/agent.py:108-141) -------------------------------------------------------------- */
#define NAVSIM_ORCA_MAX_AGENTS 64      /* agents per query incl. agent 0 */
#define NAVSIM_ORCA_MAX_EDGES  128     /* obstacle edges per polygon set (max_obst * n_vert) */
typedef struct navsim_orca_params {
    float time_step;            /* CrowdSim.time_step (rvo2.PyRVOSimulator(time_step, ...)) */
    float neighbor_dist;        /* orca.py:62 10 */
    float time_horizon;         /* orca.py:64 5 */
    float time_horizon_obst;    /* orca.py:65 5 */
    int32_t max_neighbors;      /* orca.py:63 10 */
} navsim_orca_params;
/* What ORCA.predict does per pedestrian, for Q pedestrians at once: an RVO2 simulator holding agent 0 (the
 * pedestrian itself: position, velocity, radius + 0.01 + safety_space, max speed v_pref, preferred velocity toward
 * its goal) and the other agents it sees (preferred velocity 0), plus the static obstacle polygons; one doStep();
 * the new velocity of agent 0 and the ActionRot(v, r = atan2(vy, vx) - theta) made of it.  Only agent 0's
 * velocity is read back, so only it is computed.  rvo2's source is not in the reference tree: the RVO2 Library 2.0
 * algorithm is restated (oracle/navsim_ref.c says where it knowingly differs: no kd-trees) -- UNPINNED.
 * agents [Q, A, 6] float64 = px, py, vx, vy, radius, max_speed (A <= NAVSIM_ORCA_MAX_AGENTS); n_agents [Q] or NULL;
 * pref_vel [Q,2]; verts [S, O, V, 2] counter-clockwise polygons (O * V <= NAVSIM_ORCA_MAX_EDGES), n_obst [S] or
 * NULL, obst_set [Q] polygon set of each query or NULL (= set 0); theta [Q] or NULL; out_vel [Q,2]; out_action
 * [Q,2] or NULL.  float32 arithmetic like the library's. */
int navsim_crowd_orca(const navsim_orca_params* p, int32_t n_queries, int32_t max_agents, const double* agents,
                      const int32_t* n_agents, const double* pref_vel, int32_t max_obst, int32_t n_vert,
                      const double* verts, const int32_t* n_obst, const int32_t* obst_set, const double* theta,
                      double* out_vel, double* out_action, void* stream);
/* Agent.step with an ActionRot (agent.py:108-141): theta' = theta + r; p += (cos, sin)(theta') * v * dt;
 * vel = v * (cos, sin)(theta'); theta = theta' mod 2 pi.  pose [n,3] in/out, action [n,2], vel [n,2] or NULL. */
int navsim_crowd_agent_step(double* pose, const double* action, double* vel, int32_t n, double time_step, void* stream);

/* navsim_replan + navsim_step in ONE launch (ABI 5): the pedestrians the PREVIOUS step flagged in st->ped_due_prev are
 * re-planned (exactly as navsim_replan(cfg, st, max_queries) would: same candidates, same order, same cap and counters)
 * and then every arena is stepped (exactly as navsim_step would).  The arena with a waiting pedestrian is re-planned by its
 * own workgroup right before that workgroup steps it, and those workgroups open the launch, so the searches -- a chain of
 * dependent breadth-first levels for ~2 % of the arenas -- run beside the step of all the others instead of behind it
 * (round 4) or on a second stream (navsim_step_part).  Needs st->costmap, st->ped_due / ped_due_prev (two buffers the
 * caller alternates, as for navsim_step_part).  A rollout  step_replan, step_replan, ...  equals  step, replan, step,
 * replan, ...  shifted by one re-plan: the first call finds no flags, a final navsim_replan leaves the same state. */
int navsim_step_replan(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io, int32_t max_queries,
                       void* stream);

/* Everything a later navsim_step / navsim_reset_obs / navsim_regen / navsim_replan with this configuration would set ONCE per
 * kernel (dynamic LDS above 64 KB needs hipFuncSetAttribute) is set now; nothing is launched.  Call it before capturing those
 * calls in a hipGraph (nav_gym_amd/sim.py NavSim.enable_graphs): attribute calls do not belong inside a capture.  io: the
 * buffers a reset would use (checked like navsim_reset_obs checks them). */
int navsim_prepare(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io);

/* First observation after reset() (env.py:822-831): scan at the current robot pose, stack filled
 * with copies, prev_pose = pose, vel = 0; sets prev_pose/prev_action/n_hist.  `mask` [E] uint8 or
 * NULL selects which envs are (re)initialised; others keep obs_prev -> obs copied through. */
int navsim_reset_obs(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                     const uint8_t* mask, void* stream);

/* reset() of SOME arenas, part one (env.py:730-746; ABI 6): every arena with mask[e] != 0 takes the next start / goal pair of
 * its spawn table and the next episode number, its step counter returns to 0 (and done_steps[e], when present, records the
 * steps of the episode that is abandoned) -- what the step does at `done` under auto-reset.  Follow it with
 * navsim_reset_obs(mask) (same map: first observations) or navsim_regen with io->done = mask (a new world per arena).
 * Needs the spawn tables. */
int navsim_restart(const navsim_config* cfg, const navsim_state* st, const uint8_t* mask, void* stream);

/* Name of the fused step kernel as rocprofv3 reports it (bench.py / profiles). */
const char* navsim_step_kernel_name(void);

/* ---- test hooks (used by tests/ only) ---------------------------------------------------- */
size_t navsim_sizeof_config(void);
size_t navsim_sizeof_state(void);
size_t navsim_sizeof_step_io(void);
/* deterministic device math of DESIGN.md section 4: fn 0 sin, 1 cos, 2 atan2(x, x2), 3 exp(x<=0),
 * 4 angle_correction (utils.py:5-9), 5 python-float % 2pi, 6 the packed field's sqrtf on integers, 11 / 12 the
 * march step of sqrtf(x) under NAVSIM_MARCH_F64 in its float64 form / in its float32-only form (7-10: diagnostics), 13 the
 * beam-table direction against the full evaluation, 14 the social-force pair term's antisymmetry (0 = f(i,j) == -f(j,i) bitwise).
 * x, x2, out are device float64 [n]. */
int navsim_debug_math(int32_t fn, const double* x, const double* x2, double* out, int32_t n, void* stream);
/* batch_xy_to_ij (env.py:1228-1253) exactly as the scan evaluates it: xy [n,2] float64 in, ij [n,2] int32 out;
 * as_f32 = 1 rounds the inputs to float32 first and divides in float32 (the lidar origin, env.py:386, 419). */
int navsim_debug_xy_to_ij(const navsim_config* cfg, const double* xy, int32_t as_f32, int32_t* ij, int32_t n, void* stream);

/* The spawn loops' acceptance rules (env.py:366-383, 748-762, 786-793) on SUPPLIED candidates -- the very device
 * functions navsim_regen's samplers call -- so that tests can compare them with the decisions the reference's own
 * _sample_start_goal_path / reset() took on the same candidates (tests/golden/golden_reset.npz).
 * cost [Hc,Wc] uint8 costmap (nonzero = blocked); kind [n]: 0 = the robot's pair, 1 = a pedestrian's; start, goal
 * [n,2] metres; robot [n,2] the robot's xy for kind 1 (may be NULL); wp_scratch [n, cfg.max_waypoints, 2].
 * code [n]: 0 kept, 1 start closer than cfg.ped_min_robot_dist to the robot, 2 goal distance outside its interval
 * (robot: cfg.min_goal_dist < d < cfg.max_goal_dist; pedestrian: d > cfg.ped_min_goal_dist), 3 no path, 4 (robot)
 * path_distance > 2 |goal - start|. */
int navsim_debug_spawn_decisions(const navsim_config* cfg, const uint8_t* cost, int32_t Hc, int32_t Wc, int32_t n,
                                 const int32_t* kind, const double* start, const double* goal, const double* robot,
                                 double* wp_scratch, int32_t* code, void* stream);

/* ---- measurement hooks (used by profiles/ only; nothing of the hot path calls them) ---------- */
/* Scattered-read microbenchmark behind DESIGN.md section 6's "a scattered 4-byte read costs a 128-byte fill"
 * (profiles/gather_granularity.py): n_threads lanes each read `iters` pseudo-random words of x[n_words];
 * mode selects the access width / stride pattern; out[n_threads] keeps the loads alive. */
int navsim_debug_gather(const float* x, uint64_t n_words, int32_t mode, int32_t iters, int32_t n_threads,
                        float* out, void* stream);
/* Phase time stamps of the fused step (profiles/_r01_tools/stamp_phases.py, profiles/_diag/tail_profile.py):
 * device buffer of 8 x n_envs uint64 receiving s_memtime / s_memrealtime at the phase boundaries of every arena's
 * workgroup; NULL disables.  Only a library built with -DNAVSIM_STAMPS records anything; the shipped build
 * compiles no stamp into the kernel and returns NAVSIM_E_UNSUPPORTED. */
int navsim_debug_set_stamps(unsigned long long* buf);
/* Self-test of the way the step kernels read their arguments (round 6: kernels_step.hpp NAVSIM_KERNARGS -- references into the
 * kernarg segment laid out as a struct of the kernel's leading parameters, instead of the by-value copies): one launch whose
 * parameters carry byte patterns compares every byte of the views with the copies.  NAVSIM_OK when they agree;
 * NAVSIM_E_UNSUPPORTED when a toolchain lays kernel arguments out differently (build with -DNAVSIM_KARG_VIEW=0 then). */
int navsim_debug_kernarg_layout(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAVSIM_H */
