/* navsim_ref.h -- TEST INFRASTRUCTURE: CPU oracle for the NavGym step() hot path.
 *
 * A plain-C, one-env-at-a-time restatement of the reference algorithm
 * (leekwoon/nav-gym, nav_gym/src/nav_gym_env/env.py:591-728 and the L1 packages it calls).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (nav-gym_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   pinned   : rows a7-a9, a11-a13, a15, a16 of SURVEY.md section 8a and the orchestration of a1
 *              are checked against golden vectors captured from the importable reference modules
 *              (tests/golden/make_golden.py -> tests/golden/ npz files).
 *   UNPINNED : rows a3-a6 (range_libc distance transform + ray marching, CMap2D polygon and leg
 *              rendering).  Their source is not in /root/reference (pip packages
 *              pyrangelibc-danieldugas, pymap2d; no version pin: nav_gym/setup.py:23-26) and the
 *              reference holds no test or golden vector for them -- "parity unpinned".  This file
 *              restates their published algorithms; every assumption is stated at the function.
 *              They are cross-checked by independent means only (SciPy EDT, brute-force DDA,
 *              closed-form ray/rectangle and ray/circle cases).
 *   BUILD-DEFINED : the social-force pedestrian model and the respawn tables (not in the reference).
 *
 * Same entry points as include/navsim.h with a `_cpu` suffix, host pointers, no stream.
 */
#ifndef NAVSIM_REF_H
#define NAVSIM_REF_H

#include "../include/navsim.h"

#ifdef __cplusplus
extern "C" {
#endif

int navsim_default_config_cpu(navsim_config* cfg);

int navsim_build_dt_cpu(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, float* field);

int navsim_cast_static_cpu(const float* field, int32_t n_envs, int32_t map_h, int32_t map_w,
                           const float* queries, int32_t n_per_env, float max_range, int32_t march_rule,
                           float* out);

/* brute-force reference of the same ray rule, used only to cross-check the sphere trace:
 * walks t = 0, 1, 2, ... (unit steps) and reports the first occupied cell. */
int navsim_cast_unit_steps_cpu(const uint8_t* occ, int32_t map_h, int32_t map_w,
                               const float* queries, int32_t n, float max_range, float* out);

int navsim_render_polys_cpu(float* ranges, const double* angles, int32_t n_envs, int32_t n_beams,
                            const float* verts, const int32_t* n_verts, int32_t max_verts,
                            const float* origin);

int navsim_render_legs_cpu(float* ranges, const double* angles, int32_t n_envs, int32_t n_beams,
                           const float* agents, const int32_t* n_agents, int32_t max_agents,
                           const float* origin);

/* leg circle centres of one CSimAgent: out[4] = right x,y, left x,y (float32) */
int navsim_leg_centres_cpu(const float* agent8, float* out4);

int navsim_integrate_cpu(double* pose, const double* cmd, double* vel_out, int32_t n,
                         double time_step, double axle_offset);

int navsim_reward_done_cpu(const navsim_config* cfg, const void* obs, const void* goals,
                           int32_t obs_is_f64, int32_t n,
                           const float* scan_threshold, const float* scan_discomfort,
                           double* reward, uint8_t* done, float* is_success, float* is_crash,
                           double* distance);

int navsim_scan_threshold_cpu(const navsim_config* cfg, const float* footprint, int32_t n_vert,
                              float* out);

int navsim_ped_scans_cpu(const navsim_config* cfg, const navsim_state* st, float* out);

int navsim_regen_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io);
int navsim_ped_policy_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_policy_weights* w,
                          const float* ped_scans, float* prev_actions, double* ped_cmd);
int navsim_crowd_check_cpu(const navsim_crowd_params* p, int32_t n_envs, int32_t max_agents, int32_t grid,
                           const uint8_t* free_map, const double* robot, const double* agents, const int32_t* n_agents,
                           const double* global_time, double* reward, uint8_t* done, int32_t* info, double* min_dist);
int navsim_replan_cpu(const navsim_config* cfg, const navsim_state* st, int32_t max_queries);

/* costmap (env.py:312-332), shortest 4-connected path (env.py:343-354) and path_to_waypoints
 * (env.py:1261-1277); see navsim_ref.c for the stated tie-break */
int navsim_costmap_cpu(const uint8_t* occ, int32_t n_maps, int32_t map_h, int32_t map_w, uint8_t* cost);
int navsim_path_to_waypoints_cpu(const double* path, int32_t n, double interval, double* wp, int32_t max_wp);
int navsim_plan_cpu(const uint8_t* cost, const int32_t* map_index, int32_t n_queries, int32_t Hc, int32_t Wc,
                    double res_c, double ox, double oy, const double* start, const double* goal, double interval,
                    int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len);

int navsim_step_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io);
/* same, envs [e0, e1) only: lets the CPU baseline split envs over threads */
int navsim_step_range_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                          int32_t e0, int32_t e1);

/* CPU baseline: n_steps steps of every arena on n_threads POSIX threads, arenas split statically, no barrier between steps
 * (arenas are independent).  actions [n_steps, E, 2]; obs_a = current observations, obs_b = the second buffer (step s reads
 * one and writes the other); io = the output arrays.  See navsim_ref.c. */
int navsim_step_threads_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io, const double* actions,
                            float* obs_a, float* obs_b, int32_t n_threads, int32_t n_steps);

/* a copy of the fields [E, H, W] first touched by the threads of navsim_step_threads_cpu(n_threads) (node-local on a
 * multi-socket host); NULL on failure; release with navsim_free_cpu */
float* navsim_field_local_copy_cpu(const float* field, int32_t n_envs, int32_t map_h, int32_t map_w, int32_t n_threads);
void navsim_free_cpu(void* p);

/* reset() of some arenas, part one (include/navsim.h navsim_restart) */
int navsim_restart_cpu(const navsim_config* cfg, const navsim_state* st, const uint8_t* mask);
int navsim_reset_obs_cpu(const navsim_config* cfg, const navsim_state* st, const navsim_step_io* io,
                         const uint8_t* mask);

/* a15: batch_xy_to_ij (env.py:1228-1253).  xy [n,2] float64, out [n,2] int64. */
int navsim_xy_to_ij_cpu(const double* xy, int32_t n, double origin_x, double origin_y,
                        double resolution, int32_t height, int32_t width, int64_t* out);

/* same for float32 inputs (the scan origin, env.py:419): float32 arithmetic, see navsim_ref.c */
int navsim_xy_to_ij_f32_cpu(const float* xy, int32_t n, double origin_x, double origin_y,
                            double resolution, int32_t height, int32_t width, int64_t* out);

/* a7: one _update_dist_travelled step for n pedestrians (env.py:237-255). */
int navsim_leg_odometry_cpu(const double* pose, const double* vel, const double* prev_yaw,
                            double time_step, int32_t n, double* dist);

/* deterministic math (navmath_ref.h) exposed for tests: fn 0 sin, 1 cos, 2 atan2(x=y_in,y=x2),
 * 3 exp_neg, 4 wrap_pi, 5 mod_2pi */
int navsim_math_cpu(int32_t fn, const double* x, const double* x2, double* out, int32_t n);

/* CrowdSim-v0 local maps (crowd_sim.py:999-1186); see include/navsim.h navsim_crowd_angular_map / _local_map */
int navsim_crowd_angular_map_cpu(const navsim_crowd_map_params* p, int32_t n_envs, int32_t max_obst, int32_t n_vert,
                                 const double* robot, const double* verts, const int32_t* n_obst, double* out);
int navsim_crowd_local_map_cpu(const navsim_crowd_map_params* p, int32_t n_envs, int32_t grid, const uint8_t* free_map,
                               const double* robot, int32_t rotate, uint8_t* out);

/* CrowdSim-v0 pedestrians: ORCA (RVO2 restated, unpinned) and Agent.step; see include/navsim.h */
int navsim_crowd_orca_cpu(const navsim_orca_params* p, int32_t n_queries, int32_t max_agents, const double* agents,
                          const int32_t* n_agents, const double* pref_vel, int32_t max_obst, int32_t n_vert,
                          const double* verts, const int32_t* n_obst, const int32_t* obst_set, const double* theta,
                          double* out_vel, double* out_action);
int navsim_crowd_agent_step_cpu(double* pose, const double* action, double* vel, int32_t n, double time_step);

/* tests only: the spawn loops' acceptance rules on supplied candidates; see include/navsim.h navsim_debug_spawn_decisions */
int navsim_spawn_decisions_cpu(const navsim_config* c, const uint8_t* cost, int32_t Hc, int32_t Wc, int32_t n,
                               const int32_t* kind, const double* start, const double* goal, const double* robot,
                               int32_t* code);

/* statistics for DESIGN.md: distance-field probes of the last cast/step on this thread */
int64_t navsim_probe_count_cpu(int32_t reset);
/* rays traced on this thread since the last reset by number of probes: hist256[n], n = 255 collects >= 255 */
int navsim_probe_hist_cpu(int64_t* hist256, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
