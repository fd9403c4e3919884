#define _GNU_SOURCE            /* sched_getaffinity / pthread_setaffinity_np: the CPU baseline pins its threads */
/* navsim_ref.c -- TEST INFRASTRUCTURE: CPU oracle for the NavGym step() hot path.
 * See navsim_ref.h for the parity status of every row.  Compile with -ffp-contract=off
 * (oracle/Makefile): the HIP kernels must reproduce these float32/float64 operation sequences
 * bit for bit, so no fused multiply-add may be introduced behind our back.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/nav_gym/src/nav_gym_env/ unless stated).
 */
#include "navsim_ref.h"
#include "navmath_ref.h"

#include <float.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdlib.h>
#include <string.h>


static __thread int64_t g_probe_count = 0;
static __thread int64_t g_probe_hist[256];      /* rays by number of probes (last bin: >= 255), SURVEY.md 8d work counters */
int navsim_costmap_cpu(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, uint8_t* cost);
int navsim_plan_cpu(const uint8_t* cost, const int32_t* map_index, int32_t n_maps, int32_t Hc, int32_t Wc,
                    double res_c, double ox, double oy, const double* start, const double* goal, double interval,
                    int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len);
static int plan_cpu(const uint8_t* cost, const int32_t* map_index, int32_t n_maps, int32_t Hc, int32_t Wc,
                    double res_c, double ox, double oy, const double* start, const double* goal, double interval,
                    int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len, int32_t* n_total);

int64_t navsim_probe_count_cpu(int32_t reset) {
    int64_t v = g_probe_count;
    if (reset) g_probe_count = 0;
    return v;
}

/* rays traced on this thread since the last reset, binned by the number of distance-field probes each one made */
int navsim_probe_hist_cpu(int64_t* hist256, int32_t reset) {
    if (hist256) memcpy(hist256, g_probe_hist, sizeof(g_probe_hist));
    if (reset) memset(g_probe_hist, 0, sizeof(g_probe_hist));
    return NAVSIM_OK;
}

/* =========================================================================================
 * config defaults: __init__.py:6-38, keti_robot.py:44-48
 * ======================================================================================= */
int navsim_default_config_cpu(navsim_config* c) {
    if (!c) return NAVSIM_E_ARG;
    memset(c, 0, sizeof(*c));
    c->n_envs = 1;
    c->n_beams = 512;                       /* keti_robot.py:48 */
    c->map_h = 400; c->map_w = 400;         /* map_generator.py:133 */
    c->max_peds = 16;
    c->n_scan_stack = 1;                    /* __init__.py:11 */
    c->ped_model = NAVSIM_PED_NONE;
    c->lidar_legs = 1;                      /* env.py:697 */
    c->resolution = 0.05;                   /* map_generator.py:139 */
    c->time_step = 0.2;                     /* __init__.py:8 */
    c->angle_min = -3.141592;               /* keti_robot.py:45 */
    c->angle_last = 3.141592 - 0.0122718463;/* keti_robot.py:44,46; env.py:389 */
    c->range_max = 25.0;                    /* keti_robot.py:47 */
    c->axle_offset = 0.14474;               /* keti_robot.py:73 */
    c->min_turning_radius = 0.0;            /* __init__.py:9 */
    c->distance_threshold = 0.5;            /* __init__.py:10 */
    c->reward_scale = 15.0;                 /* __init__.py:19-25 */
    c->reward_success_factor = 1.0;
    c->reward_crash_factor = 1.0;
    c->reward_progress_factor = 0.001;
    c->reward_forward_factor = 0.0;
    c->reward_rotation_factor = 0.005;
    c->reward_discomfort_factor = 0.01;
    c->sfm_tau = 0.5;                       /* build-defined: DESIGN.md section 5 */
    c->sfm_k_desired = 1.0;
    c->sfm_k_social = 2.1;
    c->sfm_k_obstacle = 10.0;
    c->sfm_lambda = 2.0;
    c->sfm_gamma = 0.35;
    c->sfm_n = 2.0;
    c->sfm_n_prime = 3.0;
    c->sfm_sigma_obstacle = 0.8;
    c->sfm_agent_radius = 0.35;
    c->ped_angle_min = -1.57079632679;      /* human.py:13 */
    c->ped_angle_last = 1.57079632679 - 0.00613592315;   /* human.py:12,14; env.py:389 */
    c->ped_range_max = 6.0;                 /* human.py:15 */
    c->ped_n_beams = 512;                   /* human.py:16 */
    {   /* keti_robot.py:18-23 threshold_footprint */
        const double fp[8] = {0.6, 0.6, -0.7, 0.6, -0.7, -0.6, 0.6, -0.6};
        for (int i = 0; i < 8; ++i) c->robot_seen_footprint[i] = fp[i];
    }
    c->regen_cap = 64;
    c->obstacle_number = 10;                /* __init__.py:34 */
    c->obstacle_width_lo = 0.3;             /* __init__.py:35 */
    c->obstacle_width_hi = 1.0;
    c->spawn_clearance = 1.2;
    c->ped_clearance = 0.5;
    c->min_goal_dist = 10.0;                /* __init__.py:17-18 */
    c->max_goal_dist = 20.0;
    c->ped_min_robot_dist = 4.0;            /* env.py:372 */
    c->ped_min_goal_dist = 10.0;            /* env.py:788-791 */
    c->v_pref_lo = 0.0;                     /* __init__.py:14 */
    c->v_pref_hi = 0.6;
    c->has_legs_ratio = 0.5;                /* __init__.py:15 */
    c->regen_indoor_ratio = 0.0;
    c->obstacle_number_hi = 0;
    c->corridor_width_lo = 3; c->corridor_width_hi = 4;
    c->iterations_lo = 80; c->iterations_hi = 150;
    c->num_humans_lo = 0; c->num_humans_hi = 0;
    c->scan_noise_std_lo = 0.0; c->scan_noise_std_hi = -1.0;
    c->regen_check_discomfort = 1;          /* env.py:776-781 */
    c->defer_reset_scan = 0;
    c->march_rule = NAVSIM_MARCH_F32;       /* RangeLib.h: `float step_coeff = 0.999;` (include/navsim.h NAVSIM_MARCH_*) */
    c->max_waypoints = 64;                  /* include/navsim.h */
    c->action_kind = NAVSIM_ACTION_TWIST;   /* env.py:591 */
    c->clamp_action = 0;                    /* env.py:611-613: never clipped */
    c->wheel_radius = 0.1651;               /* third_party/husky_description/urdf/husky.urdf.xacro:67 */
    c->wheel_track = 0.5708;                /* husky.urdf.xacro:62 */
    c->linvel_lo = 0.0; c->linvel_hi = 0.5;         /* __init__.py:12 */
    c->rotvel_lo = -0.64; c->rotvel_hi = 0.64;      /* __init__.py:13 */
    c->seed = 1234;
    return NAVSIM_OK;
}

/* =========================================================================================
 * a3  range_libc.PyOMap + PyRayMarching.__init__  (called at env.py:337-340)
 *
 * [UPSTREAM-RECALL, unpinned] PyOMap(bool[H,W]) stores grid[x][y] = arr[y][x]; the
 * DistanceTransform of RangeLib.h sets f = 0 on occupied cells, +inf elsewhere, runs the
 * Felzenszwalb-Huttenlocher exact squared Euclidean transform and takes sqrt.  The exact
 * transform has one answer, so we state the result, not the float parabola arithmetic:
 *     field[y][x] = sqrtf((float)d2),  d2 = min over occupied (x',y') of (x-x')^2 + (y-y')^2
 * computed here with Meijster's integer two-pass algorithm.  A column without obstacles uses
 * g = 2^15, so an empty map yields d = 32768 everywhere (any value >= max_range behaves alike).
 * ======================================================================================= */
#define DT_INF 32768LL

static inline int64_t mj_f(int64_t x, int64_t i, const int64_t* g) {
    return (x - i) * (x - i) + g[i] * g[i];
}
static inline int64_t mj_sep(int64_t i, int64_t u, const int64_t* g) {
    return (u * u - i * i + g[u] * g[u] - g[i] * g[i]) / (2 * (u - i));
}

int navsim_build_dt_cpu(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, float* field) {
    if (!occ || !field || n_maps < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    int64_t* g = (int64_t*)malloc(sizeof(int64_t) * (size_t)H * W);
    int64_t* s = (int64_t*)malloc(sizeof(int64_t) * (size_t)W);
    int64_t* t = (int64_t*)malloc(sizeof(int64_t) * (size_t)W);
    int64_t* row = (int64_t*)malloc(sizeof(int64_t) * (size_t)W);
    if (!g || !s || !t || !row) { free(g); free(s); free(t); free(row); return NAVSIM_E_ARG; }
    for (int32_t m = 0; m < n_maps; ++m) {
        const uint8_t* o = occ + (size_t)m * H * W;
        float* f = field + (size_t)m * H * W;
        /* phase 1: per column, distance to the nearest occupied cell in that column */
        for (int x = 0; x < W; ++x) {
            g[x] = o[x] ? 0 : DT_INF;
            for (int y = 1; y < H; ++y)
                g[(size_t)y * W + x] = o[(size_t)y * W + x] ? 0 :
                    (g[(size_t)(y - 1) * W + x] >= DT_INF ? DT_INF : g[(size_t)(y - 1) * W + x] + 1);
            for (int y = H - 2; y >= 0; --y) {
                int64_t below = g[(size_t)(y + 1) * W + x];
                if (below < DT_INF && below + 1 < g[(size_t)y * W + x]) g[(size_t)y * W + x] = below + 1;
            }
        }
        /* phase 2: per row, lower envelope of the parabolas (x-i)^2 + g(i)^2 */
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) row[x] = g[(size_t)y * W + x];
            int64_t q = 0; s[0] = 0; t[0] = 0;
            for (int64_t u = 1; u < W; ++u) {
                while (q >= 0 && mj_f(t[q], s[q], row) > mj_f(t[q], u, row)) --q;
                if (q < 0) { q = 0; s[0] = u; }
                else {
                    int64_t w = 1 + mj_sep(s[q], u, row);
                    if (w < W) { ++q; s[q] = u; t[q] = w; }
                }
            }
            for (int64_t u = W - 1; u >= 0; --u) {
                int64_t d2 = mj_f(u, s[q], row);
                f[(size_t)y * W + u] = sqrtf((float)d2);
                if (u == t[q]) --q;
            }
        }
    }
    free(g); free(s); free(t); free(row);
    return NAVSIM_OK;
}

/* =========================================================================================
 * a4  range_libc.PyRayMarching.calc_range_many  (env.py:425)
 *
 * [UPSTREAM-RECALL, unpinned] RangeLib.h RayMarching::calc_range, float32 throughout:
 *     dx = cosf(theta); dy = sinf(theta); t = 0
 *     while t < max_range:
 *         px = (int)(x + dx*t); py = (int)(y + dy*t)            (C truncation toward zero)
 *         if px or py outside the grid: return max_range
 *         d = distance_field[px][py]
 *         if d <= 0: return sqrtf((px-x)^2 + (py-y)^2)
 *         t += max(d * step_coeff, 1.0)
 *     return max_range
 * Two roundings recall cannot settle, hence a switch (navsim_config.march_rule, include/navsim.h NAVSIM_MARCH_*; the
 * `march_rule` argument of the mirror primitive):
 *   `d * step_coeff` -- the published class keeps `float step_coeff = 0.999;` as a data member, so the product is the
 *   float32 d * 0.999f (NAVSIM_MARCH_F32, the default since round 4); were it a double literal the product would be
 *   fl32(fl64(d) * 0.999) (NAVSIM_MARCH_F64, the default of rounds 1-3);
 *   `x + dx*t` -- two float32 roundings as written, ONE when the compiler contracts it into an FMA, which GCC does for
 *   upstream's build flags (-O3 -march=native -ffast-math) on every x86 with FMA: NAVSIM_MARCH_F32_FMA = the float
 *   coefficient with px = (int)fmaf(dx, t, x).
 * tests/test_oracle_crosscheck.py counts how many rays change their hit cell between the rules and DESIGN.md records
 * the numbers.
 * Assumptions stated: no ROS world<->grid conversion (numpy-constructed map), grid[x][y] is
 * occupancy[y][x]; cosf/sinf are replaced by the specified nvr_cos/nvr_sin evaluated in double on
 * the float32 heading and rounded once to float32.
 * ======================================================================================= */
static float trace_ray_counted(const float* f, int H, int W, float x0, float y0, float dx, float dy,
                               float max_range, int march_rule, int* n_probes) {
    float t = 0.0f;
    const int fma_pos = march_rule == NAVSIM_MARCH_F32_FMA;
    while (t < max_range) {
        float fx = fma_pos ? fmaf(dx, t, x0) : x0 + dx * t;
        float fy = fma_pos ? fmaf(dy, t, y0) : y0 + dy * t;
        int px = (int)fx;
        int py = (int)fy;
        if (px >= W || px < 0 || py < 0 || py >= H) return max_range;
        float d = f[(size_t)py * W + px];
        ++g_probe_count;
        ++*n_probes;
        if (d <= 0.0f) {
            float xd = (float)px - x0;
            float yd = (float)py - y0;
            return sqrtf(xd * xd + yd * yd);
        }
        float step = (march_rule != NAVSIM_MARCH_F64) ? d * 0.999f : (float)((double)d * 0.999);
        t += (step > 1.0f) ? step : 1.0f;
    }
    return max_range;
}
static float trace_ray(const float* f, int H, int W, float x0, float y0, float dx, float dy,
                       float max_range, int march_rule) {
    int n = 0;
    float r = trace_ray_counted(f, H, W, x0, y0, dx, dy, max_range, march_rule, &n);
    ++g_probe_hist[n < 255 ? n : 255];
    return r;
}

static inline void beam_dir(float heading, float* dx, float* dy) {
    double s, c;
    nvr_sincos((double)heading, &s, &c);
    *dx = (float)c;
    *dy = (float)s;
}

int navsim_cast_static_cpu(const float* field, int32_t E, int32_t H, int32_t W,
                           const float* q, int32_t n, float max_range, int32_t march_rule, float* out) {
    if (!field || !q || !out || E < 0 || n < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    if (march_rule < NAVSIM_MARCH_F64 || march_rule > NAVSIM_MARCH_F32_FMA) return NAVSIM_E_ARG;
    for (int e = 0; e < E; ++e) {
        const float* f = field + (size_t)e * H * W;
        for (int k = 0; k < n; ++k) {
            const float* qq = q + ((size_t)e * n + k) * 3;
            float dx, dy;
            beam_dir(qq[2], &dx, &dy);
            out[(size_t)e * n + k] = trace_ray(f, H, W, qq[0], qq[1], dx, dy, max_range, march_rule);
        }
    }
    return NAVSIM_OK;
}

/* cross-check only (tests/test_oracle_crosscheck.py, the "fourth rounding"): the march with the ray directions SUPPLIED --
 * q [n,4] = x, y, dx, dy -- and the directions themselves, as the specification evaluates them (which = 0: the correctly
 * rounded fl32(cos64(fl64(heading))) of beam_dir) or as the C library of THIS machine does (which = 1: cosf / sinf, what
 * upstream's RayMarching::calc_range calls; glibc's are not correctly rounded for every argument) */
int navsim_cast_dirs_cpu(const float* field, int32_t H, int32_t W, const float* q, int32_t n, float max_range,
                         int32_t march_rule, float* out) {
    if (!field || !q || !out || n < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    if (march_rule < NAVSIM_MARCH_F64 || march_rule > NAVSIM_MARCH_F32_FMA) return NAVSIM_E_ARG;
    for (int k = 0; k < n; ++k) {
        const float* qq = q + (size_t)k * 4;
        int cnt = 0;
        out[k] = trace_ray_counted(field, H, W, qq[0], qq[1], qq[2], qq[3], max_range, march_rule, &cnt);
    }
    return NAVSIM_OK;
}
int navsim_dirs_cpu(const float* heading, int32_t n, int32_t which, float* out2) {
    if (!heading || !out2 || n < 0) return NAVSIM_E_ARG;
    for (int k = 0; k < n; ++k) {
        if (which == 0) beam_dir(heading[k], &out2[2 * k], &out2[2 * k + 1]);
        else { out2[2 * k] = cosf(heading[k]); out2[2 * k + 1] = sinf(heading[k]); }
    }
    return NAVSIM_OK;
}

/* cross-check only: same sampling rule with unit steps everywhere (no distance field) */
int navsim_cast_unit_steps_cpu(const uint8_t* occ, int32_t H, int32_t W, const float* q, int32_t n,
                               float max_range, float* out) {
    if (!occ || !q || !out) return NAVSIM_E_ARG;
    for (int k = 0; k < n; ++k) {
        float x0 = q[3 * k], y0 = q[3 * k + 1], dx, dy;
        beam_dir(q[3 * k + 2], &dx, &dy);
        float t = 0.0f, r = max_range;
        while (t < max_range) {
            int px = (int)(x0 + dx * t), py = (int)(y0 + dy * t);
            if (px >= W || px < 0 || py < 0 || py >= H) break;
            if (occ[(size_t)py * W + px]) {
                float xd = (float)px - x0, yd = (float)py - y0;
                r = sqrtf(xd * xd + yd * yd);
                break;
            }
            t += 1.0f;
        }
        out[k] = r;
    }
    return NAVSIM_OK;
}

/* =========================================================================================
 * a5  CMap2D.flatten_contours + render_contours_in_lidar  (env.py:430-431)
 *
 * [UPSTREAM-RECALL, unpinned] for every beam and every polygon edge: ray/segment intersection
 * distance from the lidar origin along the beam; ranges[k] = min(ranges[k], dist).  Polygons are
 * closed automatically (edge from the last vertex of a contour id back to its first): env.py:168
 * passes an un-closed 4-vertex footprint and needs all four sides.  Stated arithmetic (float32):
 *     e = q - p;  w = p - o;  denom = c*e.y - s*e.x;  if denom == 0: skip (parallel)
 *     t = (w.x*e.y - w.y*e.x)/denom;  u = (w.x*s - w.y*c)/denom;  hit iff t >= 0 and 0 <= u <= 1
 * with (c, s) the beam direction of the float32-rounded angle (same direction as the map ray).
 * ======================================================================================= */
static inline void seg_merge(float* r, float ox, float oy, float c, float s,
                             float px, float py, float qx, float qy) {
    float ex = qx - px, ey = qy - py;
    float wx = px - ox, wy = py - oy;
    float denom = c * ey - s * ex;
    if (denom == 0.0f) return;
    float t = (wx * ey - wy * ex) / denom;
    float u = (wx * s - wy * c) / denom;
    if (t >= 0.0f && u >= 0.0f && u <= 1.0f && t < *r) *r = t;
}

static void render_polys_env(float* ranges, const double* angles, int B, const float* verts, int nv,
                             float ox, float oy) {
    for (int k = 0; k < B; ++k) {
        float c, s;
        beam_dir((float)angles[k], &c, &s);
        float r = ranges[k];
        int start = 0;
        while (start < nv) {
            int end = start;
            while (end + 1 < nv && verts[3 * (end + 1)] == verts[3 * start]) ++end;
            for (int v = start; v <= end; ++v) {
                int w = (v == end) ? start : v + 1;
                seg_merge(&r, ox, oy, c, s, verts[3 * v + 1], verts[3 * v + 2],
                          verts[3 * w + 1], verts[3 * w + 2]);
            }
            start = end + 1;
        }
        ranges[k] = r;
    }
}

int navsim_render_polys_cpu(float* ranges, const double* angles, int32_t E, int32_t B,
                            const float* verts, const int32_t* n_verts, int32_t V, const float* origin) {
    if (!ranges || !angles || !verts || !n_verts || !origin) return NAVSIM_E_ARG;
    for (int e = 0; e < E; ++e)
        render_polys_env(ranges + (size_t)e * B, angles + (size_t)e * B, B,
                         verts + (size_t)e * V * 3, n_verts[e], origin[2 * e], origin[2 * e + 1]);
    return NAVSIM_OK;
}

/* =========================================================================================
 * a6  CSimAgent + CMap2D.render_agents_in_lidar  (env.py:402, 432)
 *
 * [UPSTREAM-RECALL, unpinned] two leg discs per agent, radius 0.03 m; body-frame offsets
 *     front = 0.3*cos(2*dist_x/0.3 + dist_theta),  side = 0.1*cos(2*dist_y/0.1 + dist_theta)
 *     right leg (front, side + 0.1),  left leg (-front, -side - 0.1)
 * rotated by the agent heading and translated to the agent position (converter map has
 * resolution 1, origin 0: env.py:102-103, so ij == xy).  Stated arithmetic: offsets and the
 * frame change in double on the float32 inputs, centres rounded to float32; ray/circle in float32:
 *     w = centre - o;  b = w.c + w.s (projection);  x = w.x*s - w.y*c;  disc = r^2 - x^2
 *     miss if disc < 0;  t = b - sqrtf(disc);  miss if t < 0;  ranges = min(ranges, t)
 * ======================================================================================= */
#define LEG_RADIUS 0.03f

static void leg_centres(const float* a, float* out4) {
    double px = (double)a[0], py = (double)a[1], th = (double)a[2];
    double dxx = (double)a[3], dyy = (double)a[4], dth = (double)a[5];
    double front = 0.3 * nvr_cos(dxx * 2.0 / 0.3 + dth);
    double side = 0.1 * nvr_cos(dyy * 2.0 / 0.1 + dth);
    double s, c;
    nvr_sincos(th, &s, &c);
    double rx = front, ry = side + 0.1;
    double lx = -front, ly = -side - 0.1;
    out4[0] = (float)((c * rx - s * ry) + px);
    out4[1] = (float)((s * rx + c * ry) + py);
    out4[2] = (float)((c * lx - s * ly) + px);
    out4[3] = (float)((s * lx + c * ly) + py);
}

int navsim_leg_centres_cpu(const float* agent8, float* out4) {
    if (!agent8 || !out4) return NAVSIM_E_ARG;
    leg_centres(agent8, out4);
    return NAVSIM_OK;
}

static inline void circle_merge(float* r, float ox, float oy, float c, float s, float cx, float cy,
                                float rad) {
    float wx = cx - ox, wy = cy - oy;
    float b = wx * c + wy * s;
    float x = wx * s - wy * c;
    float disc = rad * rad - x * x;
    if (disc < 0.0f) return;
    float t = b - sqrtf(disc);
    if (t >= 0.0f && t < *r) *r = t;
}

int navsim_render_legs_cpu(float* ranges, const double* angles, int32_t E, int32_t B,
                           const float* agents, const int32_t* n_agents, int32_t A, const float* origin) {
    if (!ranges || !angles || !agents || !n_agents || !origin) return NAVSIM_E_ARG;
    for (int e = 0; e < E; ++e) {
        float ox = origin[2 * e], oy = origin[2 * e + 1];
        for (int i = 0; i < n_agents[e]; ++i) {
            float cc[4];
            leg_centres(agents + ((size_t)e * A + i) * 8, cc);
            for (int k = 0; k < B; ++k) {
                float c, s;
                beam_dir((float)angles[(size_t)e * B + k], &c, &s);
                float r = ranges[(size_t)e * B + k];
                circle_merge(&r, ox, oy, c, s, cc[0], cc[1], LEG_RADIUS);
                circle_merge(&r, ox, oy, c, s, cc[2], cc[3], LEG_RADIUS);
                ranges[(size_t)e * B + k] = r;
            }
        }
    }
    return NAVSIM_OK;
}

/* =========================================================================================
 * a8 / a9  Human.set_vel (human.py:32-41), KetiRobot.set_vel (keti_robot.py:64-93)
 * Heading-first unicycle; the Keti variant moves the point 0.14474 m ahead of the pose
 * (keti_robot.py:71-90; the 4x4 products there reduce to px + tx exactly).
 * ======================================================================================= */
static void set_vel(double* p, double v, double w, double dt, double off, double* vel) {
    double s0, c0, s1, c1;
    nvr_sincos(p[2], &s0, &c0);
    if (vel) { vel[0] = v * c0; vel[1] = v * s0; }     /* human.py:35-36: OLD heading */
    double rx = p[0] + off * c0;
    double ry = p[1] + off * s0;
    double th = p[2] + w * dt;
    nvr_sincos(th, &s1, &c1);
    rx = rx + c1 * v * dt;
    ry = ry + s1 * v * dt;
    p[0] = rx + (-off) * c1;
    p[1] = ry + (-off) * s1;
    p[2] = nvr_mod_2pi(p[2] + w * dt);
}

int navsim_integrate_cpu(double* pose, const double* cmd, double* vel_out, int32_t n, double dt,
                         double off) {
    if (!pose || !cmd) return NAVSIM_E_ARG;
    for (int i = 0; i < n; ++i)
        set_vel(pose + 3 * i, cmd[2 * i], cmd[2 * i + 1], dt, off, vel_out ? vel_out + 2 * i : NULL);
    return NAVSIM_OK;
}

/* =========================================================================================
 * a15  batch_xy_to_ij (env.py:1228-1253): (xy - origin)/resolution stored to float32, clipped
 * (i against height, j against width -- env.py:1244-1247), truncated.  This is the float64-input
 * form (callers env.py:348-349); the float32-input form used by the scan is xy_to_ij_f32 below.
 * ======================================================================================= */
static inline void xy_to_ij(double x, double y, const navsim_config* c, int* i, int* j) {
    float fi = (float)((x - c->origin_x) / c->resolution);
    float fj = (float)((y - c->origin_y) / c->resolution);
    if (fi >= (float)c->map_h) fi = (float)(c->map_h - 1);
    if (fj >= (float)c->map_w) fj = (float)(c->map_w - 1);
    if (fi < 0.0f) fi = 0.0f;
    if (fj < 0.0f) fj = 0.0f;
    *i = (int)fi;
    *j = (int)fj;
}

/* The scan origin (env.py:419) feeds FLOAT32 coordinates (lidar_pos, env.py:386) into the same
 * function.  Under NumPy >= 2 (NEP 50; the only NumPy the reference can be executed with here, and
 * what the golden traces record) the Python-scalar origin and resolution are "weak", so the
 * subtraction and the division happen in float32.  NumPy 1.x promoted the scalars to float64; the
 * two differ by one cell on grid-aligned poses (e.g. x = 1.05 -> 20 vs 21).  The oracle follows the
 * executable evidence; DESIGN.md section 3 records the alternative. */
static inline void xy_to_ij_f32(float x, float y, const navsim_config* c, int* i, int* j) {
    float fi = (x - (float)c->origin_x) / (float)c->resolution;
    float fj = (y - (float)c->origin_y) / (float)c->resolution;
    if (fi >= (float)c->map_h) fi = (float)(c->map_h - 1);
    if (fj >= (float)c->map_w) fj = (float)(c->map_w - 1);
    if (fi < 0.0f) fi = 0.0f;
    if (fj < 0.0f) fj = 0.0f;
    *i = (int)fi;
    *j = (int)fj;
}

int navsim_xy_to_ij_cpu(const double* xy, int32_t n, double ox, double oy, double res,
                        int32_t height, int32_t width, int64_t* out) {
    if (!xy || !out) return NAVSIM_E_ARG;
    navsim_config c;
    memset(&c, 0, sizeof(c));
    c.origin_x = ox; c.origin_y = oy; c.resolution = res; c.map_h = height; c.map_w = width;
    for (int k = 0; k < n; ++k) {
        int i, j;
        xy_to_ij(xy[2 * k], xy[2 * k + 1], &c, &i, &j);
        out[2 * k] = i; out[2 * k + 1] = j;
    }
    return NAVSIM_OK;
}

int navsim_xy_to_ij_f32_cpu(const float* xy, int32_t n, double ox, double oy, double res,
                            int32_t height, int32_t width, int64_t* out) {
    if (!xy || !out) return NAVSIM_E_ARG;
    navsim_config c;
    memset(&c, 0, sizeof(c));
    c.origin_x = ox; c.origin_y = oy; c.resolution = res; c.map_h = height; c.map_w = width;
    for (int k = 0; k < n; ++k) {
        int i, j;
        xy_to_ij_f32(xy[2 * k], xy[2 * k + 1], &c, &i, &j);
        out[2 * k] = i; out[2 * k + 1] = j;
    }
    return NAVSIM_OK;
}

/* =========================================================================================
 * a7  _update_dist_travelled (env.py:237-255) + pose2d.inverse_pose2d / apply_tf_to_vel
 * vrot = (theta - prev_yaw)/dt with prev_yaw the WRAPPED yaw of the previous obs and theta in
 * [0, 2pi) (the 2*pi jumps are reference behaviour); world (vx, vy) rotated
 * by -theta ([UPSTREAM-RECALL] pose2d.rotate: x' = cos*x - sin*y, y' = sin*x + cos*y with the
 * angle -theta); dist += vel_body * dt.
 * ======================================================================================= */
static void leg_odometry(const double* pose, const double* vel, double prev_yaw, double dt, double* dist) {
    double vrot = (pose[2] - prev_yaw) / dt;
    double s, c;
    nvr_sincos(-pose[2], &s, &c);
    double bx = c * vel[0] - s * vel[1];
    double by = s * vel[0] + c * vel[1];
    dist[0] += bx * dt;
    dist[1] += by * dt;
    dist[2] += vrot * dt;
}

int navsim_leg_odometry_cpu(const double* pose, const double* vel, const double* prev_yaw, double dt,
                            int32_t n, double* dist) {
    if (!pose || !vel || !prev_yaw || !dist) return NAVSIM_E_ARG;
    for (int i = 0; i < n; ++i) leg_odometry(pose + 3 * i, vel + 2 * i, prev_yaw[i], dt, dist + 3 * i);
    return NAVSIM_OK;
}

/* =========================================================================================
 * a2  _compute_scan (env.py:385-441) for the robot: static map + pedestrians.
 * ======================================================================================= */
static const double HUMAN_FOOTPRINT[4][2] = {   /* human.py:5-10 */
    {0.22, 0.19}, {-0.22, 0.19}, {-0.22, -0.19}, {0.22, -0.19}};

typedef struct { float px, py, qx, qy; } seg_t;
typedef struct { float cx, cy; } disc_t;

/* builds the dynamic-obstacle primitives seen by the robot (env.py:392-414) */
static void gather_prims(const navsim_config* c, const navsim_state* st, int e, int n,
                         seg_t* segs, int* nseg, disc_t* discs, int* ndisc) {
    const int N = c->max_peds;
    *nseg = 0; *ndisc = 0;
    for (int i = 0; i < n; ++i) {
        const double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
        if (st->ped_has_legs[(size_t)e * N + i] && c->lidar_legs) {
            float a[8];
            const double* dd = st->ped_dist + ((size_t)e * N + i) * 3;
            a[0] = (float)pp[0]; a[1] = (float)pp[1]; a[2] = (float)pp[2];     /* env.py:399 */
            a[3] = (float)dd[0]; a[4] = (float)dd[1]; a[5] = (float)dd[2];     /* env.py:400 */
            a[6] = 0.0f; a[7] = 0.0f;
            float cc[4];
            leg_centres(a, cc);
            discs[*ndisc].cx = cc[0]; discs[*ndisc].cy = cc[1]; ++*ndisc;
            discs[*ndisc].cx = cc[2]; discs[*ndisc].cy = cc[3]; ++*ndisc;
        } else {
            /* env.py:408-414: footprint + closing vertex through T(px,py)*R(theta), then float32 */
            double s, cs;
            nvr_sincos(pp[2], &s, &cs);
            float vx[4], vy[4];
            for (int v = 0; v < 4; ++v) {
                double x = HUMAN_FOOTPRINT[v][0], y = HUMAN_FOOTPRINT[v][1];
                vx[v] = (float)((cs * x - s * y) + pp[0]);
                vy[v] = (float)((s * x + cs * y) + pp[1]);
            }
            for (int v = 0; v < 4; ++v) {
                int w = (v + 1) & 3;
                segs[*nseg].px = vx[v]; segs[*nseg].py = vy[v];
                segs[*nseg].qx = vx[w]; segs[*nseg].qy = vy[w];
                ++*nseg;
            }
        }
    }
}

static inline double linspace_k(const navsim_config* c, int k) {
    /* np.linspace(angle_min, angle_max - angle_increment, n): arange*step + start, last = stop */
    int B = c->n_beams;
    if (B == 1) return c->angle_min;
    if (k == B - 1) return c->angle_last;
    double step = (c->angle_last - c->angle_min) / (double)(B - 1);
    return (double)k * step + c->angle_min;
}

static void robot_scan(const navsim_config* c, const navsim_state* st, int e, int n_peds,
                       const double* pose, float* ranges) {
    const int B = c->n_beams, H = c->map_h, W = c->map_w;
    const float* f = (const float*)st->field + (size_t)(c->shared_field ? 0 : e) * H * W;
    float lx = (float)pose[0], ly = (float)pose[1], lth = (float)pose[2];     /* env.py:386 */
    int i0, j0;
    xy_to_ij_f32(lx, ly, c, &i0, &j0);                                         /* env.py:419 */
    seg_t segs[4 * NAVSIM_MAX_PEDS];
    disc_t discs[2 * NAVSIM_MAX_PEDS];
    int nseg = 0, ndisc = 0;
    if (n_peds > 0) gather_prims(c, st, e, n_peds, segs, &nseg, discs, &ndisc);
    float max_range = (float)((int64_t)H * W);                                 /* env.py:337 */
    float res = (float)c->resolution;
    float rmax = (float)c->range_max;
    for (int k = 0; k < B; ++k) {
        double ang = linspace_k(c, k) + (double)lth;                           /* env.py:388-390 */
        float heading = (float)ang;                                            /* env.py:424 */
        float dx, dy;
        beam_dir(heading, &dx, &dy);
        float r = trace_ray(f, H, W, (float)i0, (float)j0, dx, dy, max_range, c->march_rule); /* env.py:425 */
        r = r * res;                                                           /* env.py:426 */
        for (int q = 0; q < nseg; ++q)
            seg_merge(&r, lx, ly, dx, dy, segs[q].px, segs[q].py, segs[q].qx, segs[q].qy);
        for (int q = 0; q < ndisc; ++q)
            circle_merge(&r, lx, ly, dx, dy, discs[q].cx, discs[q].cy, LEG_RADIUS);
        if (r < 0.0f) r = 0.0f;                                                /* env.py:435 */
        if (r > rmax) r = rmax;
        ranges[k] = r;
    }
}

/* env.py:776-781: reset() re-draws the robot when its first scan -- taken before any pedestrian exists -- has a beam
 * inside the discomfort zone.  (The reference's check scan carries the episode's scan noise; the oracle has no noise
 * generator and the device adds none here: BUILD-DEFINED, like every random number.) */
static int spawn_in_discomfort(const navsim_config* c, const navsim_state* st, int e, const double* pose) {
    float* ranges = (float*)malloc(sizeof(float) * (size_t)c->n_beams);
    robot_scan(c, st, e, 0, pose, ranges);
    int bad = 0;
    for (int k = 0; k < c->n_beams && !bad; ++k) bad = ranges[k] < st->scan_discomfort[k];
    free(ranges);
    return bad;
}

/* env.py:685-693: pedestrian scans (human lidar, robot + other pedestrians as polygons, no legs) */
int navsim_ped_scans_cpu(const navsim_config* c, const navsim_state* st, float* out) {
    if (!c || !st || !out || c->ped_model == NAVSIM_PED_NONE) return NAVSIM_E_ARG;
    if (c->field_format != NAVSIM_FIELD_F32) return NAVSIM_E_UNSUPPORTED;
    const int N = c->max_peds, PB = c->ped_n_beams, H = c->map_h, W = c->map_w;
    const float max_range = (float)((int64_t)H * W);
    const float res = (float)c->resolution, rmax = (float)c->ped_range_max;
    const double step = (PB > 1) ? (c->ped_angle_last - c->ped_angle_min) / (double)(PB - 1) : 0.0;
    for (int e = 0; e < c->n_envs; ++e) {
        const float* f = (const float*)st->field + (size_t)(c->shared_field ? 0 : e) * H * W;
        int n = st->n_peds[e] > N ? N : st->n_peds[e];
        /* rectangles of every agent: pedestrians 0..n-1, then the robot (env.py:404-414) */
        float vx[NAVSIM_MAX_PEDS + 1][4], vy[NAVSIM_MAX_PEDS + 1][4];
        for (int a = 0; a <= n; ++a) {
            const double* pose = (a < n) ? st->ped_pose + ((size_t)e * N + a) * 3 : st->robot_pose + 3 * (size_t)e;
            double s, cs;
            nvr_sincos(pose[2], &s, &cs);
            for (int v = 0; v < 4; ++v) {
                double x = (a < n) ? HUMAN_FOOTPRINT[v][0] : c->robot_seen_footprint[2 * v];
                double y = (a < n) ? HUMAN_FOOTPRINT[v][1] : c->robot_seen_footprint[2 * v + 1];
                vx[a][v] = (float)((cs * x - s * y) + pose[0]);
                vy[a][v] = (float)((s * x + cs * y) + pose[1]);
            }
        }
        for (int i = 0; i < n; ++i) {
            const double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
            float lx = (float)pp[0], ly = (float)pp[1], lth = (float)pp[2];
            int i0, j0;
            xy_to_ij_f32(lx, ly, c, &i0, &j0);
            float* row = out + ((size_t)e * N + i) * PB;
            for (int k = 0; k < PB; ++k) {
                double lin = (PB == 1) ? c->ped_angle_min : ((k == PB - 1) ? c->ped_angle_last : (double)k * step + c->ped_angle_min);
                float heading = (float)(lin + (double)lth);
                float dx, dy;
                beam_dir(heading, &dx, &dy);
                float r = trace_ray(f, H, W, (float)i0, (float)j0, dx, dy, max_range, c->march_rule) * res;
                for (int a = 0; a <= n; ++a) {
                    if (a == i) continue;
                    for (int v = 0; v < 4; ++v) {
                        int w = (v + 1) & 3;
                        seg_merge(&r, lx, ly, dx, dy, vx[a][v], vy[a][v], vx[a][w], vy[a][w]);
                    }
                }
                if (r < 0.0f) r = 0.0f;
                if (r > rmax) r = rmax;
                row[k] = r;
            }
        }
    }
    return NAVSIM_OK;
}

/* a14  _make_scan_threshold (env.py:162-180): contour-only scan of a footprint at pose 0 */
int navsim_scan_threshold_cpu(const navsim_config* c, const float* fp, int32_t nv, float* out) {
    if (!c || !fp || !out || nv < 2 || nv > 16) return NAVSIM_E_ARG;
    float rmax = (float)c->range_max;
    for (int k = 0; k < c->n_beams; ++k) {
        double ang = linspace_k(c, k) + (double)0.0f;
        float dx, dy;
        beam_dir((float)ang, &dx, &dy);
        float r = rmax;
        for (int v = 0; v < nv; ++v) {
            int w = (v + 1 == nv) ? 0 : v + 1;
            seg_merge(&r, 0.0f, 0.0f, dx, dy, fp[2 * v], fp[2 * v + 1], fp[2 * w], fp[2 * w + 1]);
        }
        if (r < 0.0f) r = 0.0f;
        if (r > rmax) r = rmax;
        out[k] = r;
    }
    return NAVSIM_OK;
}

/* =========================================================================================
 * a12 / a13  compute_rewards (env.py:521-589), compute_terminals (491-512), compute_info (464-482)
 * on one row: scan = latest scan of the stack, tail = prev_pose, pose, vel, yaw.
 * ======================================================================================= */
typedef struct { double reward; int done; float success, crash; double distance; } rd_t;

static rd_t reward_done_row(const navsim_config* c, const double* scan_d, const float* scan_f,
                            const double* prev_pose, const double* pose, const double* vel,
                            const double* goal, const float* thr, const float* dthr) {
    rd_t o;
    const int B = c->n_beams;
    double dx = goal[0] - pose[0], dy = goal[1] - pose[1];
    double distance = sqrt(dx * dx + dy * dy);                                 /* env.py:539 */
    double px = goal[0] - prev_pose[0], py = goal[1] - prev_pose[1];
    double prev_distance = sqrt(px * px + py * py);                            /* env.py:540 */
    int success = distance < c->distance_threshold;                            /* env.py:542 */
    int crash = 0, discomfort = 0;
    double ratio_min = 0.0;
    for (int k = 0; k < B; ++k) {
        double s = scan_d ? scan_d[k] : (double)scan_f[k];
        if (s - (double)thr[k] < 0.0) crash = 1;                               /* env.py:543-544 */
        if (s - (double)dthr[k] < 0.0) discomfort = 1;                         /* env.py:545-546 */
        float den = (dthr[k] - thr[k]) + 1e-6f;                                /* env.py:566 (float32) */
        double ratio = (s - (double)thr[k]) / (double)den;                     /* env.py:564-567 */
        if (k == 0 || ratio < ratio_min) ratio_min = ratio;
    }
    if (crash) discomfort = 0;                                                 /* env.py:547 */
    double r_success = success ? 1.0 * c->reward_success_factor * c->reward_scale : 0.0;
    double r_crash = crash ? -1.0 * c->reward_crash_factor * c->reward_scale : 0.0;
    double r_progress = (prev_distance - distance) * c->reward_progress_factor * c->reward_scale;
    double r_forward = vel[0] * c->reward_forward_factor * c->reward_scale;
    double r_rotation = -1.0 * (vel[1] * vel[1]) * c->reward_rotation_factor * c->reward_scale;
    double r_discomfort = discomfort
        ? -(1.0 - ratio_min) * c->reward_discomfort_factor * c->reward_scale : 0.0;
    o.reward = r_success + r_crash + r_progress + r_forward + r_rotation + r_discomfort;
    o.done = success || crash;                                                 /* env.py:511 */
    o.success = (float)success;
    o.crash = (float)crash;
    o.distance = distance;
    return o;
}

int navsim_reward_done_cpu(const navsim_config* c, const void* obs, const void* goals, int32_t is64,
                           int32_t n, const float* thr, const float* dthr, double* reward,
                           uint8_t* done, float* is_success, float* is_crash, double* distance) {
    if (!c || !obs || !goals || !thr || !dthr) return NAVSIM_E_ARG;
    const int B = c->n_beams, S = c->n_scan_stack, D = S * B + 7;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)B);
    for (int r = 0; r < n; ++r) {
        double tail[7], goal[2];
        if (is64) {
            const double* o = (const double*)obs + (size_t)r * D;
            memcpy(tmp, o + (size_t)(S - 1) * B, sizeof(double) * B);
            memcpy(tail, o + (size_t)S * B, sizeof(tail));
            goal[0] = ((const double*)goals)[2 * r]; goal[1] = ((const double*)goals)[2 * r + 1];
        } else {
            const float* o = (const float*)obs + (size_t)r * D;
            for (int k = 0; k < B; ++k) tmp[k] = (double)o[(size_t)(S - 1) * B + k];
            for (int k = 0; k < 7; ++k) tail[k] = (double)o[(size_t)S * B + k];
            goal[0] = (double)((const float*)goals)[2 * r]; goal[1] = (double)((const float*)goals)[2 * r + 1];
        }
        rd_t o = reward_done_row(c, tmp, NULL, tail, tail + 2, tail + 4, goal, thr, dthr);
        if (reward) reward[r] = o.reward;
        if (done) done[r] = (uint8_t)o.done;
        if (is_success) is_success[r] = o.success;
        if (is_crash) is_crash[r] = o.crash;
        if (distance) distance[r] = o.distance;
    }
    free(tmp);
    return NAVSIM_OK;
}

/* =========================================================================================
 * BUILD-DEFINED social-force pedestrian update (NAVSIM_PED_SFM).
 * Not in the reference (its pedestrians are driven by HumanPolicy, env.py:650-662, whose weights
 * are missing).  Follows the public pedsim model named by BASELINE.json's north_star; force names
 * as in third_party/pedsim_msgs/msg/AgentForce.msg:3-6.
 * ======================================================================================= */
/* waypoint pop (env.py:633-642): the reference pops reached waypoints off the front of a Python list; the list here stays
 * as its planner stored it and the index of the current waypoint advances (navsim_state.ped_wp_head, ABI 5).
 * wp = the pedestrian's row, nw = waypoints stored. */
static void pop_waypoints(const double* wp, int32_t* head, int nw, const double* pp) {
    while (*head + 1 < nw) {
        double ddx = pp[0] - wp[2 * *head], ddy = pp[1] - wp[2 * *head + 1];
        if (sqrt(ddx * ddx + ddy * ddy) < 1.0) *head += 1;
        else break;
    }
}

static void sfm_update(const navsim_config* c, const navsim_state* st, int e, int n,
                       const double* robot_pose, const double* robot_prev_action) {
    const int N = c->max_peds, H = c->map_h, W = c->map_w, P = c->max_waypoints;
    const float* f = (const float*)st->field + (size_t)(c->shared_field ? 0 : e) * H * W;
    double ax[NAVSIM_MAX_PEDS + 1], ay[NAVSIM_MAX_PEDS + 1];      /* agent positions (peds + robot) */
    double avx[NAVSIM_MAX_PEDS + 1], avy[NAVSIM_MAX_PEDS + 1];
    for (int i = 0; i < n; ++i) {
        const double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
        const double* vv = st->ped_vel + ((size_t)e * N + i) * 2;
        ax[i] = pp[0]; ay[i] = pp[1]; avx[i] = vv[0]; avy[i] = vv[1];
    }
    {
        double s, cs;
        nvr_sincos(robot_pose[2], &s, &cs);
        ax[n] = robot_pose[0]; ay[n] = robot_pose[1];
        avx[n] = robot_prev_action[0] * cs; avy[n] = robot_prev_action[0] * s;
    }
    double nvx[NAVSIM_MAX_PEDS], nvy[NAVSIM_MAX_PEDS];
    for (int i = 0; i < n; ++i) {
        const double* wp = st->ped_waypoints + (((size_t)e * N + i) * P + (size_t)st->ped_wp_head[(size_t)e * N + i]) * 2;   /* the current waypoint */
        double vpref = st->ped_v_pref[(size_t)e * N + i];
        /* desired force */
        double ex = wp[0] - ax[i], ey = wp[1] - ay[i];
        double L = sqrt(ex * ex + ey * ey);
        if (L > 1e-9) { ex = ex / L; ey = ey / L; } else { ex = 0.0; ey = 0.0; }
        double fdx = (vpref * ex - avx[i]) / c->sfm_tau;
        double fdy = (vpref * ey - avy[i]) / c->sfm_tau;
        /* social force */
        double fsx = 0.0, fsy = 0.0;
        for (int j = 0; j <= n; ++j) {
            if (j == i) continue;
            double dxx = ax[j] - ax[i], dyy = ay[j] - ay[i];
            double dist = sqrt(dxx * dxx + dyy * dyy);
            if (dist < 1e-9) continue;
            double ddx = dxx / dist, ddy = dyy / dist;
            double ivx = c->sfm_lambda * (avx[i] - avx[j]) + ddx;
            double ivy = c->sfm_lambda * (avy[i] - avy[j]) + ddy;
            double il = sqrt(ivx * ivx + ivy * ivy);
            if (il < 1e-9) continue;
            double idx = ivx / il, idy = ivy / il;
            double theta = nvr_atan2(idx * ddy - idy * ddx, idx * ddx + idy * ddy);
            double Bq = c->sfm_gamma * il;
            double a1 = c->sfm_n_prime * Bq * theta;
            double a2 = c->sfm_n * Bq * theta;
            double fv = -nvr_exp_neg(-dist / Bq - a1 * a1);
            double sgn = (theta > 0.0) ? 1.0 : ((theta < 0.0) ? -1.0 : 0.0);
            double fa = -sgn * nvr_exp_neg(-dist / Bq - a2 * a2);
            fsx += fv * idx + fa * (-idy);
            fsy += fv * idy + fa * idx;
        }
        /* obstacle force from the distance field (central differences) */
        double fox = 0.0, foy = 0.0;
        {
            int ci, cj;
            xy_to_ij(ax[i], ay[i], c, &ci, &cj);
            if (ci > W - 1) ci = W - 1;
            if (cj > H - 1) cj = H - 1;
            int il_ = ci > 0 ? ci - 1 : 0, ir = ci < W - 1 ? ci + 1 : W - 1;
            int jl = cj > 0 ? cj - 1 : 0, jr = cj < H - 1 ? cj + 1 : H - 1;
            double d = (double)f[(size_t)cj * W + ci] * c->resolution;
            double gx = (double)f[(size_t)cj * W + ir] - (double)f[(size_t)cj * W + il_];
            double gy = (double)f[(size_t)jr * W + ci] - (double)f[(size_t)jl * W + ci];
            double gl = sqrt(gx * gx + gy * gy);
            if (gl > 0.0) {
                double mag = nvr_exp_neg(-(d - c->sfm_agent_radius) / c->sfm_sigma_obstacle);
                fox = mag * (gx / gl);
                foy = mag * (gy / gl);
            }
        }
        double accx = c->sfm_k_desired * fdx + c->sfm_k_social * fsx + c->sfm_k_obstacle * fox;
        double accy = c->sfm_k_desired * fdy + c->sfm_k_social * fsy + c->sfm_k_obstacle * foy;
        double vx = avx[i] + accx * c->time_step;
        double vy = avy[i] + accy * c->time_step;
        double sp = sqrt(vx * vx + vy * vy);
        if (sp > vpref) {
            double k = (sp > 0.0) ? vpref / sp : 0.0;
            vx = vx * k; vy = vy * k;
        }
        nvx[i] = vx; nvy[i] = vy;
    }
    for (int i = 0; i < n; ++i) {
        double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
        double* vv = st->ped_vel + ((size_t)e * N + i) * 2;
        pp[0] = pp[0] + nvx[i] * c->time_step;
        pp[1] = pp[1] + nvy[i] * c->time_step;
        double sp = sqrt(nvx[i] * nvx[i] + nvy[i] * nvy[i]);
        if (sp > 1e-6) pp[2] = nvr_mod_2pi(nvr_atan2(nvy[i], nvx[i]));
        vv[0] = nvx[i]; vv[1] = nvy[i];
    }
}

/* =========================================================================================
 * observation packing: _convert_obs (env.py:443-462) + _stack_scan (env.py:257-279), float32 out
 * ======================================================================================= */
static void pack_obs(const navsim_config* c, const float* scan, const float* obs_prev, int n_hist,
                     const double* prev_xy, const double* pose, const double* vel, const double* goal,
                     float* obs, float* ag, float* dg) {
    const int B = c->n_beams, S = c->n_scan_stack;
    for (int j = 0; j < S - 1; ++j) {
        int age = S - 1 - j;                 /* this slot holds scan_{t-age} when available */
        const float* src = (age <= n_hist && obs_prev) ? obs_prev + (size_t)(j + 1) * B : scan;
        memcpy(obs + (size_t)j * B, src, sizeof(float) * B);
    }
    memcpy(obs + (size_t)(S - 1) * B, scan, sizeof(float) * B);
    float* tail = obs + (size_t)S * B;
    tail[0] = (float)prev_xy[0]; tail[1] = (float)prev_xy[1];
    tail[2] = (float)pose[0];    tail[3] = (float)pose[1];
    tail[4] = (float)vel[0];     tail[5] = (float)vel[1];
    tail[6] = (float)nvr_wrap_pi(pose[2]);                                     /* env.py:454 */
    if (ag) { ag[0] = (float)pose[0]; ag[1] = (float)pose[1]; }
    if (dg) { dg[0] = (float)goal[0]; dg[1] = (float)goal[1]; }
}

/* =========================================================================================
 * a1  NavGymEnv.step (env.py:591-728) for one env
 * ======================================================================================= */
static void reset_env(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int e, float* scan);

/* the next start / goal pair of the arena's table and the next episode number: what ends an episode under auto-reset
 * (BUILD-DEFINED vector-env reset; include/navsim.h NAVSIM_AUTORESET_*, navsim_restart) */
static void restart_state(const navsim_config* c, const navsim_state* st, int e) {
    const uint64_t genv = (uint64_t)(c->env_index_base + e);
    uint64_t h = nvr_hash4(c->seed, genv, (uint64_t)st->episode[e], 0x5eedULL);
    int idx = (int)(h % (uint64_t)c->n_spawn);
    const double* sp = st->spawn_pose + ((size_t)e * c->n_spawn + idx) * 3;
    const double* sg = st->spawn_goal + ((size_t)e * c->n_spawn + idx) * 2;
    double* rp = st->robot_pose + 3 * (size_t)e;
    rp[0] = sp[0]; rp[1] = sp[1]; rp[2] = sp[2];
    st->robot_goal[2 * e] = sg[0]; st->robot_goal[2 * e + 1] = sg[1];
    if (st->done_steps) st->done_steps[e] = (int32_t)st->steps[e];   /* how long the episode lasted (cfg.regen_min_steps) */
    st->episode[e] += 1;
    st->steps[e] = 0;
}

int navsim_restart_cpu(const navsim_config* c, const navsim_state* st, const uint8_t* mask) {
    if (!c || !st || !mask || c->n_spawn < 1 || !st->spawn_pose || !st->spawn_goal) return NAVSIM_E_ARG;
    for (int e = 0; e < c->n_envs; ++e) if (mask[e]) restart_state(c, st, e);
    return NAVSIM_OK;
}

static void step_env(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int e,
                     float* scan) {
    const int B = c->n_beams, S = c->n_scan_stack, N = c->max_peds, D = S * B + 7;
    const int P = c->max_waypoints;
    const double dt = c->time_step;
    const uint64_t genv = (uint64_t)(c->env_index_base + e);
    if (io->reset_mask && io->reset_mask[e]) {
        /* NAVSIM_AUTORESET_NEXT_STEP: the arena finished in the previous call (its state restarted there); this call resets
         * it -- first observation as reset() gives it (env.py:808-831), pedestrians not advanced, reward / done / info zero.
         * cfg.defer_reset_scan: the row is left to the navsim_regen_cpu that follows (io->done = the same mask) */
        io->reward[e] = 0.0; io->done[e] = 0; io->is_success[e] = 0.0f; io->is_crash[e] = 0.0f; io->distance[e] = 0.0;
        if (st->ped_due && c->ped_model != NAVSIM_PED_NONE) st->ped_due[e] = 0;
        if (!c->defer_reset_scan) reset_env(c, st, io, e, scan);
        return;
    }
    double a0 = io->action[2 * e], a1 = io->action[2 * e + 1];
    st->steps[e] += 1;                                                         /* env.py:592 */
    if (c->action_kind == NAVSIM_ACTION_WHEELS) {      /* BUILD-DEFINED: skid-steer wheel speeds (left, right) -> twist */
        const double wl = a0, wr = a1;
        a0 = c->wheel_radius * (wl + wr) * 0.5;
        a1 = c->wheel_radius * (wr - wl) / c->wheel_track;
    }
    if (c->clamp_action) {                             /* build option; the reference never clips (env.py:611-613) */
        a0 = a0 < c->linvel_lo ? c->linvel_lo : (a0 > c->linvel_hi ? c->linvel_hi : a0);
        a1 = a1 < c->rotvel_lo ? c->rotvel_lo : (a1 > c->rotvel_hi ? c->rotvel_hi : a1);
    }
    if (c->min_turning_radius > 0.0) {                                         /* env.py:595-600 */
        double lim = fabs(a1) * c->min_turning_radius;
        if (a0 >= 0.0) a0 = (a0 > lim) ? a0 : lim;
        else           a0 = (a0 < -lim) ? a0 : -lim;
    }
    double* rp = st->robot_pose + 3 * (size_t)e;
    double* goal = st->robot_goal + 2 * (size_t)e;
    double* pa = st->prev_action + 2 * (size_t)e;
    double* pv = st->prev_pose + 3 * (size_t)e;
    int n = (c->ped_model == NAVSIM_PED_NONE) ? 0 : st->n_peds[e];
    if (n > N) n = N;

    /* ---- pedestrians: waypoint pop (env.py:633-642), control + integration (env.py:650-662) */
    for (int i = 0; i < n; ++i) {
        double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
        const double* wp = st->ped_waypoints + (((size_t)e * N + i) * P) * 2;
        pop_waypoints(wp, st->ped_wp_head + (size_t)e * N + i, st->ped_n_waypoints[(size_t)e * N + i], pp);
    }
    if (n > 0 && c->ped_model == NAVSIM_PED_EXTERNAL) {
        for (int i = 0; i < n; ++i) {
            const double* cmd = st->ped_cmd + ((size_t)e * N + i) * 2;
            set_vel(st->ped_pose + ((size_t)e * N + i) * 3, cmd[0], cmd[1], dt, 0.0,
                    st->ped_vel + ((size_t)e * N + i) * 2);                     /* env.py:662 */
        }
    } else if (n > 0 && c->ped_model == NAVSIM_PED_SFM) {
        sfm_update(c, st, e, n, rp, pa);
    }

    /* ---- robot (env.py:664) */
    set_vel(rp, a0, a1, dt, c->axle_offset, NULL);

    /* ---- pedestrians at their final waypoint take a new goal (env.py:667-680).  With a resident
     * costmap the pedestrian waits for navsim_replan_cpu, which plans like the reference; without one
     * it draws a goal >= 10 m away from the env's spawn table and heads straight for it. */
    unsigned long long due = 0;                   /* navsim_state.ped_due: who stands on its final waypoint after this update */
    for (int i = 0; i < n; ++i) {
        double* pp = st->ped_pose + ((size_t)e * N + i) * 3;
        double* wp = st->ped_waypoints + (((size_t)e * N + i) * P) * 2;
        int* nw = st->ped_n_waypoints + (size_t)e * N + i;
        double ddx = pp[0] - wp[2 * (*nw - 1)], ddy = pp[1] - wp[2 * (*nw - 1) + 1];
        const int at_final = sqrt(ddx * ddx + ddy * ddy) < 0.5;
        if (at_final) due |= 1ull << i;
        if (at_final && c->n_spawn > 0 && st->spawn_pose && !st->costmap) {
            uint64_t h = nvr_hash4(c->seed, genv, (uint64_t)i + 1000, (uint64_t)st->steps[e]);
            for (int tries = 0; tries < c->n_spawn; ++tries) {
                int idx = (int)((h + (uint64_t)tries) % (uint64_t)c->n_spawn);
                const double* cand = st->spawn_pose + ((size_t)e * c->n_spawn + idx) * 3;
                double gx = cand[0] - pp[0], gy = cand[1] - pp[1];
                if (sqrt(gx * gx + gy * gy) > 10.0) {
                    wp[0] = cand[0]; wp[1] = cand[1]; *nw = 1;
                    st->ped_wp_head[(size_t)e * N + i] = 0;
                    if (st->ped_goal) { st->ped_goal[((size_t)e * N + i) * 2] = cand[0]; st->ped_goal[((size_t)e * N + i) * 2 + 1] = cand[1]; }
                    break;
                }
            }
        }
    }
    if (st->ped_due && c->ped_model != NAVSIM_PED_NONE) st->ped_due[e] = due;

    /* ---- leg odometry (env.py:683), then the pedestrian's obs yaw is refreshed (env.py:685-693;
     * the per-pedestrian 512-beam scans only feed HumanPolicy and are not produced: DESIGN.md) */
    for (int i = 0; i < n; ++i) {
        size_t q = (size_t)e * N + i;
        leg_odometry(st->ped_pose + q * 3, st->ped_vel + q * 2, st->ped_prev_yaw[q], dt, st->ped_dist + q * 3);
        st->ped_prev_yaw[q] = nvr_wrap_pi(st->ped_pose[q * 3 + 2]);
    }

    /* ---- robot scan A, reward / done / info on it (env.py:695-703) */
    robot_scan(c, st, e, n, rp, scan);
    rd_t o = reward_done_row(c, NULL, scan, pv, rp, pa, goal, st->scan_threshold, st->scan_discomfort);
    io->reward[e] = o.reward;
    io->done[e] = (uint8_t)o.done;
    io->is_success[e] = o.success;
    io->is_crash[e] = o.crash;
    io->distance[e] = o.distance;

    const float* obs_prev = io->obs_prev ? io->obs_prev + (size_t)e * D : NULL;
    float* obs = io->obs + (size_t)e * D;
    float* ag = io->achieved_goal ? io->achieved_goal + 2 * (size_t)e : NULL;
    float* dg = io->desired_goal ? io->desired_goal + 2 * (size_t)e : NULL;
    int n_hist = st->n_hist[e];

    const int restart = o.done && c->auto_reset != NAVSIM_AUTORESET_NONE && c->n_spawn > 0;
    if (restart && c->auto_reset == NAVSIM_AUTORESET_SAME_STEP) {
        if (io->final_obs) {
            /* what the reference's step() returns with done = True (env.py:700-728), before the restart takes the row */
            double tp[3] = {rp[0], rp[1], rp[2]};
            double vel[2] = {pa[0], pa[1]};
            float* tscan = scan;
            float* tmp = NULL;
            if (o.crash != 0.0f) {                                             /* env.py:707-723 */
                tp[0] = pv[0]; tp[1] = pv[1]; tp[2] = pv[2];
                tmp = (float*)malloc(sizeof(float) * (size_t)B);
                robot_scan(c, st, e, n, tp, tmp);
                tscan = tmp;
            }
            float fag[2], fdg[2];
            pack_obs(c, tscan, obs_prev, n_hist, pv, tp, vel, goal, io->final_obs + (size_t)e * D, fag, fdg);
            if (io->final_goals) {
                float* fg = io->final_goals + 4 * (size_t)e;
                fg[0] = fag[0]; fg[1] = fag[1]; fg[2] = fdg[0]; fg[3] = fdg[1];
            }
            free(tmp);
        }
        /* BUILD-DEFINED vector-env reset: respawn from the table, first obs as in reset()
         * (env.py:736-738, 822-831): prev_action = 0, prev_pose = pose, stack filled */
        restart_state(c, st, e);
        double zero[2] = {0.0, 0.0};
        /* cfg.defer_reset_scan: the first observation comes from the navsim_regen call that follows (its masked
         * navsim_reset_obs over every finished arena); until then the scan rows of this row are unspecified (here: scan A) */
        if (!c->defer_reset_scan) robot_scan(c, st, e, n, rp, scan);
        pack_obs(c, scan, NULL, 0, rp, rp, zero, goal, obs, ag, dg);
        pa[0] = 0.0; pa[1] = 0.0;
        st->n_hist[e] = (S - 1 < 1) ? S - 1 : 1;
    } else {
        double vel[2] = {pa[0], pa[1]};                                        /* env.py:453 */
        if (o.crash != 0.0f) {                                                 /* env.py:707-723 */
            rp[0] = pv[0]; rp[1] = pv[1]; rp[2] = pv[2];
            robot_scan(c, st, e, n, rp, scan);
        }
        pack_obs(c, scan, obs_prev, n_hist, pv, rp, vel, goal, obs, ag, dg);
        pa[0] = a0; pa[1] = a1;                                                /* env.py:725 */
        st->n_hist[e] = (n_hist + 1 < S - 1) ? n_hist + 1 : S - 1;             /* env.py:727 */
        pv[0] = rp[0]; pv[1] = rp[1]; pv[2] = nvr_wrap_pi(rp[2]);              /* env.py:726 */
        /* NAVSIM_AUTORESET_NEXT_STEP: the outputs above are the ended episode's (what the reference returns); the state
         * already belongs to the next one, whose first observation the next call's reset of this arena produces */
        if (restart) restart_state(c, st, e);
        return;
    }
    pv[0] = rp[0]; pv[1] = rp[1]; pv[2] = nvr_wrap_pi(rp[2]);                  /* env.py:726 */
}

/* first observation of ONE arena from its current state (env.py:808-831); navsim_reset_obs_cpu's body */
static void reset_env(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int e, float* scan) {
    const int B = c->n_beams, S = c->n_scan_stack, N = c->max_peds, D = S * B + 7;
    float* obs = io->obs + (size_t)e * D;
    double* rp = st->robot_pose + 3 * (size_t)e;
    double* goal = st->robot_goal + 2 * (size_t)e;
    int n = (c->ped_model == NAVSIM_PED_NONE) ? 0 : st->n_peds[e];
    if (n > N) n = N;
    for (int i = 0; i < n; ++i) {
        size_t q = (size_t)e * N + i;
        st->ped_dist[q * 3] = 0.0; st->ped_dist[q * 3 + 1] = 0.0; st->ped_dist[q * 3 + 2] = 0.0; /* env.py:809 */
        st->ped_prev_yaw[q] = nvr_wrap_pi(st->ped_pose[q * 3 + 2]);                              /* env.py:812-820 */
    }
    double zero[2] = {0.0, 0.0};
    robot_scan(c, st, e, n, rp, scan);
    pack_obs(c, scan, NULL, 0, rp, rp, zero, goal, obs,
             io->achieved_goal ? io->achieved_goal + 2 * (size_t)e : NULL,
             io->desired_goal ? io->desired_goal + 2 * (size_t)e : NULL);
    st->prev_action[2 * e] = 0.0; st->prev_action[2 * e + 1] = 0.0;        /* env.py:736 */
    st->prev_pose[3 * e] = rp[0]; st->prev_pose[3 * e + 1] = rp[1];
    st->prev_pose[3 * e + 2] = nvr_wrap_pi(rp[2]);
    st->n_hist[e] = (S - 1 < 1) ? S - 1 : 1;                               /* env.py:830 */
    st->steps[e] = 0;                                                      /* env.py:735 */
}

int navsim_step_range_cpu(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                          int32_t e0, int32_t e1) {
    if (!c || !st || !io || !io->action || !io->obs || !io->reward || !io->done || !io->is_success ||
        !io->is_crash || !io->distance) return NAVSIM_E_ARG;
    if (c->max_peds > NAVSIM_MAX_PEDS || c->n_scan_stack < 1) return NAVSIM_E_UNSUPPORTED;
    if (c->field_format != NAVSIM_FIELD_F32) return NAVSIM_E_UNSUPPORTED;   /* oracle reads float32 only */
    if (c->ped_model != NAVSIM_PED_NONE && (c->max_waypoints < 1 || c->max_waypoints > NAVSIM_MAX_WAYPOINTS)) return NAVSIM_E_ARG;
    if (c->march_rule < NAVSIM_MARCH_F64 || c->march_rule > NAVSIM_MARCH_F32_FMA) return NAVSIM_E_ARG;
    if (c->action_kind != NAVSIM_ACTION_TWIST && c->action_kind != NAVSIM_ACTION_WHEELS) return NAVSIM_E_ARG;
    float* scan = (float*)malloc(sizeof(float) * (size_t)c->n_beams);
    for (int e = e0; e < e1; ++e) step_env(c, st, io, e, scan);
    free(scan);
    return NAVSIM_OK;
}

int navsim_step_cpu(const navsim_config* c, const navsim_state* st, const navsim_step_io* io) {
    if (!c) return NAVSIM_E_ARG;
    return navsim_step_range_cpu(c, st, io, 0, c->n_envs);
}

/* The CPU baseline's own loop ("all host cores with envs split statically across threads"; round-4 verdict:
 * the Python thread pool around navsim_step_range_cpu spent its time dispatching -- 5 % parallel efficiency on 256 threads).
 * n_steps steps of every arena on n_threads POSIX threads: thread t owns the arenas [E t / n, E (t + 1) / n) for the whole
 * run and never waits for another thread -- arenas are independent (env.py:80-131: all state is one env's), so there is no
 * barrier between steps, and a thread takes each of its arenas through all n_steps before the next (cache blocking).  actions [n_steps, E, 2]; obs_a holds the current observations and receives those of even-numbered
 * runs (step s reads one buffer and writes the other: n_steps even -> the final rows are in obs_a again); io supplies the
 * output arrays.  Same arithmetic as navsim_step_cpu, arena by arena. */
typedef struct {
    const navsim_config* c; const navsim_state* st; navsim_step_io io; const double* actions;
    float* obs[2]; int32_t e0, e1, n_steps; int rc; int cpu;
    const float* copy_src; float* copy_dst; size_t copy_per_arena;      /* navsim_field_local_copy_cpu */
} step_job;

/* thread t of n runs on the t-th CPU the process may use (`allowed`, taken before any thread is pinned): a thread keeps its
 * arenas' data in ITS core's caches and, after navsim_field_local_copy_cpu, on its own memory node */
static int nth_allowed_cpu(const cpu_set_t* allowed, int t) {
    const int n = CPU_COUNT(allowed);
    if (n <= 0) return -1;
    int want = t % n;
    for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu)
        if (CPU_ISSET(cpu, allowed) && want-- == 0) return cpu;
    return -1;
}
static void pin_self(int cpu) {
    if (cpu < 0) return;
    cpu_set_t one;
    CPU_ZERO(&one); CPU_SET(cpu, &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
}

static void* step_worker(void* arg) {
    step_job* j = (step_job*)arg;
    pin_self(j->cpu);
    if (j->copy_dst) {                                                /* first touch of the copy by the thread that will read it */
        memcpy(j->copy_dst + (size_t)j->e0 * j->copy_per_arena, j->copy_src + (size_t)j->e0 * j->copy_per_arena,
               sizeof(float) * j->copy_per_arena * (size_t)(j->e1 - j->e0));
        j->rc = NAVSIM_OK;
        return NULL;
    }
    const size_t E = (size_t)j->c->n_envs;
    float* scan = (float*)malloc(sizeof(float) * (size_t)j->c->n_beams);
    /* arena by arena, every arena through ALL its steps before the next one: arenas are independent and the actions of every
     * step are there, so the order is free -- and one arena's field (1 MB at 500 x 500 cells) stays in the core's cache for
     * the whole run instead of the thread's eight being cycled through it once per step */
    for (int e = j->e0; e < j->e1; ++e)
        for (int32_t s = 0; s < j->n_steps; ++s) {
            navsim_step_io io = j->io;
            io.action = j->actions + (size_t)s * E * 2;
            io.obs_prev = j->obs[s & 1];
            io.obs = j->obs[1 - (s & 1)];
            step_env(j->c, j->st, &io, e, scan);
        }
    free(scan);
    j->rc = NAVSIM_OK;
    return NULL;
}

/* jobs[t].cpu from the process's allowed CPUs, run every job (the calling thread takes the first), give the caller its
 * affinity back */
static int run_jobs(step_job* jobs, pthread_t* th, int n_threads) {
    cpu_set_t allowed;
    const int have = sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
    for (int t = 0; t < n_threads; ++t) jobs[t].cpu = (have && n_threads > 1) ? nth_allowed_cpu(&allowed, t) : -1;
    int started = 0;
    for (int t = 1; t < n_threads; ++t, ++started)
        if (pthread_create(&th[t], NULL, step_worker, &jobs[t]) != 0) break;
    step_worker(&jobs[0]);
    for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    for (int t = started + 1; t < n_threads; ++t) step_worker(&jobs[t]);  /* a thread that could not be created: run its share here */
    if (have && n_threads > 1) (void)pthread_setaffinity_np(pthread_self(), sizeof(allowed), &allowed);
    int rc = NAVSIM_OK;
    for (int t = 0; t < n_threads; ++t) if (jobs[t].rc != NAVSIM_OK) rc = jobs[t].rc;
    return rc;
}

int navsim_step_threads_cpu(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, const double* actions,
                            float* obs_a, float* obs_b, int32_t n_threads, int32_t n_steps) {
    if (!c || !st || !io || !actions || !obs_a || !obs_b || n_threads < 1 || n_steps < 0) return NAVSIM_E_ARG;
    navsim_step_io probe = *io;
    probe.action = actions; probe.obs = obs_b; probe.obs_prev = obs_a;
    int rc = navsim_step_range_cpu(c, st, &probe, 0, 0);               /* the argument checks, no arena stepped */
    if (rc != NAVSIM_OK) return rc;
    if (n_threads > c->n_envs) n_threads = c->n_envs > 0 ? c->n_envs : 1;
    step_job* jobs = (step_job*)calloc((size_t)n_threads, sizeof(step_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].c = c; jobs[t].st = st; jobs[t].io = *io; jobs[t].actions = actions;
        jobs[t].obs[0] = obs_a; jobs[t].obs[1] = obs_b; jobs[t].n_steps = n_steps;
        jobs[t].e0 = (int32_t)((long long)c->n_envs * t / n_threads);
        jobs[t].e1 = (int32_t)((long long)c->n_envs * (t + 1) / n_threads);
        jobs[t].rc = NAVSIM_E_ARG;
    }
    rc = run_jobs(jobs, th, n_threads);
    free(jobs); free(th);
    return rc;
}

/* A copy of the distance fields [E, H, W] whose pages are first touched by the thread that steps those arenas in
 * navsim_step_threads_cpu (same static split, same CPUs): on a multi-socket host the march then reads its own node's memory
 * instead of the node of whoever built the fields.  Free it with navsim_free_cpu. */
float* navsim_field_local_copy_cpu(const float* field, int32_t n_envs, int32_t map_h, int32_t map_w, int32_t n_threads) {
    if (!field || n_envs <= 0 || map_h <= 0 || map_w <= 0 || n_threads < 1) return NULL;
    if (n_threads > n_envs) n_threads = n_envs;
    const size_t per = (size_t)map_h * map_w;
    /* untouched pages, 2 MB-aligned and advised as huge pages: the march reads the fields at random, and with 4 KB pages
     * over gigabytes of fields nearly every probe of an all-core run pays a page walk on top of its cache miss */
    float* dst = NULL;
    const size_t bytes = sizeof(float) * per * (size_t)n_envs;
    if (posix_memalign((void**)&dst, (size_t)2 << 20, bytes) != 0) dst = NULL;
    if (dst) (void)madvise(dst, bytes, MADV_HUGEPAGE);
    step_job* jobs = (step_job*)calloc((size_t)n_threads, sizeof(step_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!dst || !jobs || !th) { free(dst); free(jobs); free(th); return NULL; }
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].e0 = (int32_t)((long long)n_envs * t / n_threads);
        jobs[t].e1 = (int32_t)((long long)n_envs * (t + 1) / n_threads);
        jobs[t].copy_src = field; jobs[t].copy_dst = dst; jobs[t].copy_per_arena = per;
        jobs[t].rc = NAVSIM_E_ARG;
    }
    (void)run_jobs(jobs, th, n_threads);
    free(jobs); free(th);
    return dst;
}
void navsim_free_cpu(void* p) { free(p); }

/* first observation after reset() (env.py:808-831) */
int navsim_reset_obs_cpu(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                         const uint8_t* mask) {
    if (!c || !st || !io || !io->obs) return NAVSIM_E_ARG;
    if (c->field_format != NAVSIM_FIELD_F32) return NAVSIM_E_UNSUPPORTED;
    const int B = c->n_beams, S = c->n_scan_stack, D = S * B + 7;
    float* scan = (float*)malloc(sizeof(float) * (size_t)B);
    for (int e = 0; e < c->n_envs; ++e) {
        float* obs = io->obs + (size_t)e * D;
        if (mask && !mask[e]) {
            if (io->obs_prev && io->obs_prev != io->obs)
                memcpy(obs, io->obs_prev + (size_t)e * D, sizeof(float) * D);
            continue;
        }
        reset_env(c, st, io, e, scan);
    }
    free(scan);
    return NAVSIM_OK;
}

/* =========================================================================================
 * reset() of finished arenas with a new random map, restated for the device
 * (navsim_regen).  BUILD-DEFINED where the reference relies on NumPy's global RNG and A*:
 *   - uniforms: u(key, i) = (mix64(key + i * 0x9E3779B97F4A7C15) >> 11) * 2^-53, keys from
 *     hash4(seed, global arena, episode, purpose);
 *   - map: create_outdoor_map (map_generator.py:126-143) at map_w x map_w cells;
 *   - start / goal pairs: 64 rejection tries each on clearance and (min, max) goal distance
 *     (env.py:366-383) -- no A* path-length test;
 *   - pedestrians: start >= ped_min_robot_dist from the robot (env.py:372), goal farther than
 *     ped_min_goal_dist (env.py:788-791), v_pref and has_legs re-drawn (env.py:800-803).
 * ======================================================================================= */
static inline double rg_u(uint64_t key, uint64_t i) {
    return (double)(nvr_mix64(key + i * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
}

/* a draw: the supplied one (tests only, navsim_state.regen_draws, NAVSIM_DRAW_* layout) or the hash-keyed one */
static inline double rg_t(const double* tape, int slot, uint64_t key, uint64_t i) {
    return tape ? tape[slot] : rg_u(key, i);
}
/* side of the map an outdoor episode draws (cfg.outdoor_map_size; the reference: 400 inside its 1000-cell arenas) */
static inline int outdoor_size(const navsim_config* c) {
    return (c->outdoor_map_size > 0 && c->outdoor_map_size < c->map_w) ? c->outdoor_map_size : c->map_w;
}

/* create_outdoor_map (map_generator.py:126-143) at size x size cells in the corner [0, size)^2 of the arena's
 * map_w x map_w array (everything outside is occupied: behind the 5-cell border wall no ray and no distance sees it) */
static void regen_map(const navsim_config* c, uint64_t genv, uint64_t ep, uint8_t* occ, const double* tape) {
    const int size = outdoor_size(c), A = c->map_w;
    const uint64_t key = nvr_hash4(c->seed, genv, ep, 0x4D4150ULL);
    uint64_t n = 0;
    double w = c->obstacle_width_lo + (c->obstacle_width_hi - c->obstacle_width_lo) * rg_t(tape, NAVSIM_DRAW_OBSTACLE_WIDTH, key, n++);
    int hw = (int)(10.0 * w);                                              /* map_generator.py:127 */
    if (size < A) memset(occ, 1, (size_t)A * A);
    for (int r = 0; r < size; ++r)
        for (int q = 0; q < size; ++q)
            occ[(size_t)(size - 1 - r) * A + q] = !(r >= 5 && r < size - 5 && q >= 5 && q < size - 5);
    int span = size - 2 * hw - 3;                                          /* range(hw+2, size-hw-1) */
    if (span < 1) span = 1;
    /* env_param['obstacle_number'] (env.py:281-292): uniform over lo..hi inclusive, its own key */
    const int obs_hi = c->obstacle_number_hi > c->obstacle_number ? c->obstacle_number_hi : c->obstacle_number;
    int n_obs = c->obstacle_number + (int)(rg_t(tape, NAVSIM_DRAW_OBSTACLE_NUMBER, nvr_hash4(c->seed, genv, ep, 0x50524DULL), 0) *
                                           (double)(obs_hi - c->obstacle_number + 1));
    if (n_obs > 64) n_obs = 64;
    for (int o = 0; o < n_obs; ++o) {
        int cx = hw + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 2 * o, key, n) * span);
        int cy = hw + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 2 * o + 1, key, n + 1) * span);
        n += 2;
        for (int r = cx - hw; r <= cx + hw; ++r)
            for (int q = cy - hw; q <= cy + hw; ++q)
                if (r >= 0 && r < size && q >= 0 && q < size) occ[(size_t)(size - 1 - r) * A + q] = 1;
    }
}

/* create_indoor_map (map_generator.py:97-123) at size x size, hash-keyed: a random tree of corridors on a
 * coarse grid of G = size/10 cells (0.5 m, the reference's scale for its 1000-cell maps), L1-nearest node,
 * L-shaped paths of half-width r in {3, 4} (env_param corridor_width), iterations in [80, 150] scaled with
 * the area, nearest-neighbour upscaling, vertical flip. */
static void regen_map_indoor(const navsim_config* c, uint64_t genv, uint64_t ep, uint8_t* occ, const double* tape) {
    const int size = c->map_w;
    const uint64_t key = nvr_hash4(c->seed, genv, ep, 0x494E44ULL);
    uint64_t n = 0;
    const int r = c->corridor_width_lo + (int)(rg_t(tape, NAVSIM_DRAW_CORRIDOR_WIDTH, key, n) * (double)(c->corridor_width_hi - c->corridor_width_lo + 1));
    const int it = c->iterations_lo + (int)(rg_t(tape, NAVSIM_DRAW_ITERATIONS, key, n + 1) * (double)(c->iterations_hi - c->iterations_lo + 1));
    n += 2;
    int G = size / 10;
    if (G < 2 * r + 8) G = 2 * r + 8;
    if (G > 100) G = 100;
    int n_it = (it * G * G + 5000) / 10000;
    if (n_it < 4) n_it = 4;
    if (n_it > 150) n_it = 150;
    static __thread uint8_t g[100 * 100];
    int tx[152], ty[152], nt = 1;
    memset(g, 1, (size_t)G * G);
    tx[0] = G / 2; ty[0] = G / 2;
    g[(G / 2) * G + G / 2] = 0;
    const int span = G - 2 * r - 3;                                    /* range(r + 2, G - r - 1) */
    for (int k = 0; k < n_it; ++k) {
        int px = r + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 3 * k, key, n) * span);
        int py = r + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 3 * k + 1, key, n + 1) * span);
        int coin = rg_t(tape, NAVSIM_DRAW_MAP + 3 * k + 2, key, n + 2) >= 0.5;        /* map_generator.py:61, 68 */
        n += 3;
        int best = 0, bd = 1 << 30;
        for (int t = 0; t < nt; ++t) {                                  /* first L1-nearest node */
            int dd = abs(px - tx[t]) + abs(py - ty[t]);
            if (dd < bd) { bd = dd; best = t; }
        }
        int qx = tx[best], qy = ty[best];
        tx[nt] = px; ty[nt] = py; ++nt;
        g[px * G + py] = 0;
        int x1 = px < qx ? px : qx, x2 = px < qx ? qx : px;
        int y1 = py < qy ? py : qy, y2 = py < qy ? qy : py;
        int constellation1 = (px > qx && py < qy) || (px < qx && py > qy);   /* map_generator.py:43-57 */
        int hx, cy;                                                     /* horizontal leg row, vertical leg column */
        if (coin) { hx = x1; cy = constellation1 ? y1 : y2; }
        else      { hx = x2; cy = constellation1 ? y2 : y1; }
        for (int a = hx - r; a <= hx + r; ++a)
            for (int b = y1 - r; b <= y2 + r; ++b)
                if (a >= 0 && a < G && b >= 0 && b < G) g[a * G + b] = 0;
        for (int a = x1 - r; a <= x2 + r; ++a)
            for (int b = cy - r; b <= cy + r; ++b)
                if (a >= 0 && a < G && b >= 0 && b < G) g[a * G + b] = 0;
    }
    for (int yy = 0; yy < size; ++yy) {
        int gy = (int)(((long long)yy * G) / size);
        for (int xx = 0; xx < size; ++xx) {
            int gx = (int)(((long long)xx * G) / size);
            occ[(size_t)(size - 1 - yy) * size + xx] = g[gy * G + gx];
        }
    }
}

static inline void rg_cell_xy(const navsim_config* c, int i, int j, double* x, double* y) {
    *x = ((double)i + 0.5) * c->resolution + c->origin_x;                  /* env.py:1218-1219 */
    *y = ((double)j + 0.5) * c->resolution + c->origin_y;
}

/* The acceptance rules of _sample_start_goal_path (env.py:366-383), shared by every sampler below and by
 * navsim_spawn_decisions_cpu (which tests/ compare with decisions recorded from the reference's own loop):
 *   a start is dropped when it is closer than `min_robot` to the robot:   dist < 4   (env.py:371-373) -> kept at ==
 *   a goal is kept when   min_goal_dist < dist < max_goal_dist   (env.py:379), both strict */
static inline int rg_start_ok(double x, double y, double rx, double ry, double min_robot) {
    double ddx = rx - x, ddy = ry - y;
    return !(sqrt(ddx * ddx + ddy * ddy) < min_robot);
}
static inline int rg_goal_ok(double sx, double sy, double gx, double gy, double dmin, double dmax) {
    double ddx = sx - gx, ddy = sy - gy;
    double dist = sqrt(ddx * ddx + ddy * ddy);
    return dmin < dist && dist < dmax;
}
/* the robot's pair survives reset()'s own test (env.py:756-762) unless path_distance > 2 |goal - start| */
static inline int rg_robot_path_ok(double plen, double sx, double sy, double gx, double gy) {
    double ddx = gx - sx, ddy = gy - sy;
    return !(plen > 2.0 * sqrt(ddx * ddx + ddy * ddy));
}

/* one rejection-sampled free cell: first try (of 64) with field >= clr that passes the rule of its kind (kind 0:
 * none, 1: a start, dropped when closer than dmin to (rx, ry); 2: a goal of the start (rx, ry));
 * fallback = the tried cell with the best clearance.  Cells are drawn in the live map [0, size)^2. */
static void rg_sample(const navsim_config* c, const float* f, int size, uint64_t key, uint64_t* n, double clr,
                      int kind, double rx, double ry, double dmin, double dmax, double* x, double* y) {
    const int W = c->map_w;
    int bi = 0, bj = 0; float bd = -1.0f;
    for (int t = 0; t < 64; ++t) {
        int i = (int)(rg_u(key, (*n)++) * size), j = (int)(rg_u(key, (*n)++) * size);
        float d = f[(size_t)j * W + i];
        double px, py;
        rg_cell_xy(c, i, j, &px, &py);
        int ok = (double)d >= clr;
        if (ok && kind == 1) ok = rg_start_ok(px, py, rx, ry, dmin);
        if (ok && kind == 2) ok = rg_goal_ok(rx, ry, px, py, dmin, dmax);
        if (ok) { *x = px; *y = py; return; }
        if (d > bd) { bd = d; bi = i; bj = j; }
    }
    rg_cell_xy(c, bi, bj, x, y);
}

/* regen with path planning (cfg.regen_plan = 1): _sample_start_goal_path (env.py:342-383) on the
 * costmap -- starts and goals are centres of free COSTMAP cells, a pair is kept only if a path joins
 * it, and for the robot only if that path is not longer than 2x the straight line (env.py:756-762);
 * pedestrians start >= 4 m from the robot and walk > 10 m (env.py:369-379, 788-791) along waypoints every
 * 2 m (env.py:804).  Four rounds of candidates; a slot that never succeeds keeps its last candidate.
 * Each candidate cell: up to 16 uniform tries, fallback = last try. */
static void rgp_cell(const navsim_config* c, const uint8_t* cost, int Wc, int live_w, int live_h, double res_c, uint64_t key,
                     uint64_t* n, int kind, double rx, double ry, double dmin, double dmax, double* x, double* y) {
    for (int t = 0; t < 16; ++t) {
        int I = (int)(rg_u(key, (*n)++) * live_w), J = (int)(rg_u(key, (*n)++) * live_h);
        *x = ((double)I + 0.5) * res_c + c->origin_x;
        *y = ((double)J + 0.5) * res_c + c->origin_y;
        if (cost[(size_t)J * Wc + I]) continue;
        if (kind == 1 && !rg_start_ok(*x, *y, rx, ry, dmin)) continue;
        if (kind == 2 && !rg_goal_ok(rx, ry, *x, *y, dmin, dmax)) continue;
        return;
    }
}

static void regen_planned(const navsim_config* c, const navsim_state* st, int e, uint64_t genv, uint64_t ep,
                          const uint8_t* occ, int size) {
    const int N = c->max_peds, K = c->n_spawn, H = c->map_h, W = c->map_w, P = c->max_waypoints;
    const int Hc = H / 5, Wc = W / 5;
    const double res_c = c->resolution * 5.0;
    const int live_c = size / 5;                /* candidates are cells of the live map's costmap */
    uint8_t* cost = (uint8_t*)malloc((size_t)Hc * Wc);
    navsim_costmap_cpu(occ, 1, H, W, cost);
    double* sp = (double*)st->spawn_pose + (size_t)e * K * 3;
    double* sg = (double*)st->spawn_goal + (size_t)e * K * 2;
    double wp[2 * NAVSIM_MAX_WAYPOINTS];
    int32_t nwp; double plen;
    int resolved[256];
    for (int k = 0; k < K; ++k) resolved[k] = 0;
    for (int round = 0; round < 4; ++round)
        for (int k = 0; k < K && k < 256; ++k) {
            if (resolved[k]) continue;
            uint64_t key = nvr_hash4(c->seed, genv, ep, 0x52504C00ULL + (uint64_t)round * 256 + (uint64_t)k), n = 0;
            double s[2], g[2];
            rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, &n, 0, 0, 0, 0, 0, &s[0], &s[1]);
            rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, &n, 2, s[0], s[1], c->min_goal_dist, c->max_goal_dist, &g[0], &g[1]);
            sp[3 * k] = s[0]; sp[3 * k + 1] = s[1]; sp[3 * k + 2] = NVR_TWO_PI * rg_u(key, n++);
            sg[2 * k] = g[0]; sg[2 * k + 1] = g[1];
            navsim_plan_cpu(cost, NULL, 1, Hc, Wc, res_c, c->origin_x, c->origin_y, s, g, 5.0, P, wp, &nwp, NULL, &plen);
            resolved[k] = nwp > 0 && rg_robot_path_ok(plen, s[0], s[1], g[0], g[1]);      /* env.py:761 */
        }
    int idx = (int)(nvr_hash4(c->seed, genv, ep, 0x5eedULL) % (uint64_t)K);
    {   /* first resolved pair from idx on (cyclic) whose first scan is outside the discomfort zone (env.py:776-781);
           none: the first resolved one; none resolved: idx */
        int first_res = -1, pick = -1;
        for (int s_ = 0; s_ < K && pick < 0; ++s_) {
            int j = (idx + s_) % K;
            if (!resolved[j]) continue;
            if (first_res < 0) first_res = j;
            if (!c->regen_check_discomfort || !spawn_in_discomfort(c, st, e, sp + 3 * j)) pick = j;
        }
        if (pick >= 0) idx = pick; else if (first_res >= 0) idx = first_res;
    }
    double* rp = st->robot_pose + 3 * (size_t)e;
    rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
    st->robot_goal[2 * e] = sg[2 * idx]; st->robot_goal[2 * e + 1] = sg[2 * idx + 1];
    int n = (c->ped_model == NAVSIM_PED_NONE) ? 0 : st->n_peds[e];
    if (n > N) n = N;
    for (int i = 0; i < n; ++i) {
        size_t q = (size_t)e * N + i;
        uint64_t k0 = nvr_hash4(c->seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
        st->ped_pose[q * 3 + 2] = NVR_TWO_PI * rg_u(k0, m++);
        ((double*)st->ped_v_pref)[q] = c->v_pref_lo + (c->v_pref_hi - c->v_pref_lo) * rg_u(k0, m++);
        ((uint8_t*)st->ped_has_legs)[q] = rg_u(k0, m++) < c->has_legs_ratio;
        st->ped_vel[q * 2] = 0.0; st->ped_vel[q * 2 + 1] = 0.0;
        double* w = st->ped_waypoints + (q * P) * 2;
        int done = 0;
        for (int round = 0; round < 4 && !done; ++round) {
            uint64_t key = nvr_hash4(c->seed, genv, ep, 0x50504C00ULL + (uint64_t)round * 256 + (uint64_t)i), nn = 0;
            double s[2], g[2];
            rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, &nn, 1, rp[0], rp[1], c->ped_min_robot_dist, 0, &s[0], &s[1]);
            rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, &nn, 2, s[0], s[1], c->ped_min_goal_dist, 1.0e300, &g[0], &g[1]);
            st->ped_pose[q * 3] = s[0]; st->ped_pose[q * 3 + 1] = s[1];
            int32_t total = 0;
            if (st->ped_goal) { st->ped_goal[q * 2] = g[0]; st->ped_goal[q * 2 + 1] = g[1]; }
            plan_cpu(cost, NULL, 1, Hc, Wc, res_c, c->origin_x, c->origin_y, s, g, 2.0, P, w, &nwp, NULL, NULL, &total);
            if (total > P && st->counters) st->counters[NAVSIM_COUNTER_ROUTES_CUT] += 1;
            if (nwp > 0) { st->ped_n_waypoints[q] = nwp; done = 1; }
            else { w[0] = g[0]; w[1] = g[1]; st->ped_n_waypoints[q] = 1; }
            st->ped_wp_head[q] = 0;
        }
    }
    if (st->ped_due) st->ped_due[e] = 0;
    free(cost);
}

/* ---------------------------------------------------------------------------------------------
 * Pedestrian control block with the HumanPolicy actor (env.py:617-662, human_policy.py:19-52).
 * Specification shared with the HIP kernels: float32, every dot product is the fused-multiply-add
 * chain acc = fmaf(w_k, x_k, acc) in the index order written below starting from acc = 0, the bias is
 * added after the chain, ReLU is max(.,0); sigmoid / tanh are evaluated in float64 on the deterministic
 * exp and rounded once.  fmaf() is exact (one rounding) with or without hardware FMA.
 * ------------------------------------------------------------------------------------------- */
static float policy_sigmoid(float x) {
    double xd = (double)x;
    if (xd >= 0.0) return (float)(1.0 / (1.0 + nvr_exp_neg(-xd)));
    double e = nvr_exp_neg(xd);
    return (float)(e / (1.0 + e));
}
static float policy_tanh(float x) {
    double a = fabs((double)x);
    double e = nvr_exp_neg(-2.0 * a);
    double t = (1.0 - e) / (1.0 + e);
    return (float)(x < 0.0f ? -t : t);
}

static void policy_actor(const navsim_policy_weights* w, const float x[512], const float goal[2], const float speed[2],
                         float mean[2]) {
    static __thread float o1[32][256], feat[4096], h1[260], h2[128];
    /* conv1: 3 identical channels (env.py:647), k = 5, stride 2, zero padding 1 -> 255 outputs */
    for (int o = 0; o < 32; ++o)
        for (int t = 0; t < 255; ++t) {
            float acc = 0.0f;
            for (int ch = 0; ch < 3; ++ch)
                for (int k = 0; k < 5; ++k) {
                    int idx = 2 * t + k - 1;
                    float xv = (idx >= 0 && idx < 512) ? x[idx] : 0.0f;
                    acc = fmaf(w->cv1_w[(o * 3 + ch) * 5 + k], xv, acc);
                }
            acc = acc + w->cv1_b[o];
            o1[o][t] = acc > 0.0f ? acc : 0.0f;
        }
    /* conv2: k = 3, stride 2, zero padding 1 -> 128 outputs; flatten channel-major */
    for (int o = 0; o < 32; ++o)
        for (int t = 0; t < 128; ++t) {
            float acc = 0.0f;
            for (int ch = 0; ch < 32; ++ch)
                for (int k = 0; k < 3; ++k) {
                    int idx = 2 * t + k - 1;
                    float xv = (idx >= 0 && idx < 255) ? o1[ch][idx] : 0.0f;
                    acc = fmaf(w->cv2_w[(o * 32 + ch) * 3 + k], xv, acc);
                }
            acc = acc + w->cv2_b[o];
            feat[o * 128 + t] = acc > 0.0f ? acc : 0.0f;
        }
    for (int j = 0; j < 256; ++j) {
        const float* wr = w->fc1_w + (size_t)j * 4096;
        float acc = 0.0f;
        for (int k = 0; k < 4096; ++k) acc = fmaf(feat[k], wr[k], acc);       /* MFMA order: A = features */
        acc = acc + w->fc1_b[j];
        h1[j] = acc > 0.0f ? acc : 0.0f;
    }
    h1[256] = goal[0]; h1[257] = goal[1]; h1[258] = speed[0]; h1[259] = speed[1];
    for (int j = 0; j < 128; ++j) {
        const float* wr = w->fc2_w + (size_t)j * 260;
        float acc = 0.0f;
        for (int k = 0; k < 260; ++k) acc = fmaf(wr[k], h1[k], acc);
        acc = acc + w->fc2_b[j];
        h2[j] = acc > 0.0f ? acc : 0.0f;
    }
    float a1 = 0.0f, a2 = 0.0f;
    for (int k = 0; k < 128; ++k) { a1 = fmaf(w->a1_w[k], h2[k], a1); a2 = fmaf(w->a2_w[k], h2[k], a2); }
    mean[0] = policy_sigmoid(a1 + w->a1_b[0]);
    mean[1] = policy_tanh(a2 + w->a2_b[0]);
}

int navsim_ped_policy_cpu(const navsim_config* c, const navsim_state* st, const navsim_policy_weights* w,
                          const float* ped_scans, float* prev_actions, double* ped_cmd) {
    if (!c || !st || !w || !ped_scans || !prev_actions || !ped_cmd || !st->ped_pose || !st->ped_waypoints ||
        !st->ped_n_waypoints || !st->ped_wp_head || !st->ped_v_pref || !st->n_peds)
        return NAVSIM_E_ARG;
    if (c->ped_n_beams != 512) return NAVSIM_E_UNSUPPORTED;
    const int N = c->max_peds, P = c->max_waypoints;
    for (int e = 0; e < c->n_envs; ++e) {
        int n = st->n_peds[e] > N ? N : st->n_peds[e];
        for (int i = 0; i < N; ++i) {
            size_t q = (size_t)e * N + i;
            if (i >= n) { ped_cmd[2 * q] = 0.0; ped_cmd[2 * q + 1] = 0.0; continue; }
            const double* pp = st->ped_pose + q * 3;
            const double* wp = st->ped_waypoints + (q * P) * 2;
            pop_waypoints(wp, st->ped_wp_head + q, st->ped_n_waypoints[q], pp);    /* env.py:633-640 */
            wp += 2 * (size_t)st->ped_wp_head[q];
            double s, cs;
            nvr_sincos(pp[2], &s, &cs);                                       /* env.py:644-645 */
            double gx = wp[0] - pp[0], gy = wp[1] - pp[1];
            float goal[2] = {(float)(gx * cs + gy * s), (float)(-gx * s + gy * cs)};
            float x[512];
            const float* scan = ped_scans + q * 512;
            for (int k = 0; k < 512; ++k) {                                   /* env.py:629-630 */
                double v = (double)scan[k];
                v = v < 0.0 ? 0.0 : (v > 6.0 ? 6.0 : v);
                x[k] = (float)(v / 6.0 - 0.5);
            }
            float speed[2] = {prev_actions[2 * q], prev_actions[2 * q + 1]}, mean[2];
            policy_actor(w, x, goal, speed, mean);
            mean[0] = mean[0] < 0.0f ? 0.0f : (mean[0] > 1.0f ? 1.0f : mean[0]);     /* env.py:656-657 */
            mean[1] = mean[1] < -1.0f ? -1.0f : (mean[1] > 1.0f ? 1.0f : mean[1]);
            prev_actions[2 * q] = mean[0]; prev_actions[2 * q + 1] = mean[1];
            ped_cmd[2 * q] = (double)mean[0] * st->ped_v_pref[q];              /* env.py:659-662 */
            ped_cmd[2 * q + 1] = (double)mean[1] * st->ped_v_pref[q];
        }
    }
    return NAVSIM_OK;
}

/* env.py:667-680 (see include/navsim.h navsim_replan) */
int navsim_replan_cpu(const navsim_config* c, const navsim_state* st, int32_t max_queries) {
    if (!c || !st || !st->costmap || !st->ped_pose || !st->ped_waypoints || !st->ped_n_waypoints || !st->ped_wp_head ||
        !st->n_peds || max_queries < 0)
        return NAVSIM_E_ARG;
    if (c->ped_model == NAVSIM_PED_NONE) return NAVSIM_OK;
    if (c->max_waypoints < 1 || c->max_waypoints > NAVSIM_MAX_WAYPOINTS) return NAVSIM_E_ARG;
    const int N = c->max_peds, P = c->max_waypoints, Hc = c->map_h / 5, Wc = c->map_w / 5;
    const double res_c = c->resolution * 5.0;
    int served = 0;
    for (int e = 0; e < c->n_envs; ++e) {
        int n = st->n_peds[e] > N ? N : st->n_peds[e];
        const uint8_t* cost = st->costmap + (size_t)(c->shared_field ? 0 : e) * Hc * Wc;
        const uint64_t genv = (uint64_t)(c->env_index_base + e);
        const uint64_t when = (uint64_t)st->steps[e] + ((uint64_t)st->episode[e] << 40);
        for (int i = 0; i < n; ++i) {
            size_t q = (size_t)e * N + i;
            double* pp = st->ped_pose + q * 3;
            double* w = st->ped_waypoints + (q * P) * 2;
            int nw = st->ped_n_waypoints[q];
            double ddx = pp[0] - w[2 * (nw - 1)], ddy = pp[1] - w[2 * (nw - 1) + 1];
            if (!(sqrt(ddx * ddx + ddy * ddy) < 0.5)) continue;
            if (served >= max_queries) {                /* beyond the cap: waits for a later call, and is counted */
                if (st->counters) st->counters[NAVSIM_COUNTER_REPLAN_UNSERVED] += 1;
                continue;
            }
            ++served;
            if (st->counters) st->counters[NAVSIM_COUNTER_REPLAN_SERVED] += 1;
            /* the end of a route that was stored CUT (its last stored waypoint is not its goal): walk on to the goal it
             * had (round -1, no draw); only if no path joins them draw a new goal like the others */
            double* pg = st->ped_goal ? st->ped_goal + q * 2 : NULL;
            const int cut = pg && (w[2 * (nw - 1)] != pg[0] || w[2 * (nw - 1) + 1] != pg[1]);
            for (int round = cut ? -1 : 0; round < 4; ++round) {
                double g[2], wp[2 * NAVSIM_MAX_WAYPOINTS];
                int32_t nwp, total = 0;
                if (round < 0) { g[0] = pg[0]; g[1] = pg[1]; }
                else {
                    uint64_t key = nvr_hash4(c->seed, genv, when, 0x52504E00ULL + (uint64_t)round * 256 + (uint64_t)i), m = 0;
                    rgp_cell(c, cost, Wc, Wc, Hc, res_c, key, &m, 2, pp[0], pp[1], c->ped_min_goal_dist, 1.0e300, &g[0], &g[1]);
                }
                plan_cpu(cost, NULL, 1, Hc, Wc, res_c, c->origin_x, c->origin_y, pp, g, 2.0, P, wp, &nwp, NULL, NULL, &total);
                if (total > P && st->counters) st->counters[NAVSIM_COUNTER_ROUTES_CUT] += 1;
                if (nwp > 0) {
                    memcpy(w, wp, sizeof(double) * 2 * (size_t)nwp);
                    st->ped_n_waypoints[q] = nwp;
                    st->ped_wp_head[q] = 0;
                    if (pg) { pg[0] = g[0]; pg[1] = g[1]; }
                    if (round < 0 && st->counters) st->counters[NAVSIM_COUNTER_ROUTES_RESUMED] += 1;
                    break;
                }
            }
        }
    }
    return NAVSIM_OK;
}

/* cfg.regen_min_steps (include/navsim.h; build-defined, the reference regenerates at every reset): an arena whose episode
 * ended after fewer steps keeps its map and restarts in place */
static int regen_long_enough(const navsim_config* c, const navsim_state* st, int e) {
    return c->regen_min_steps <= 0 || !st->done_steps || st->done_steps[e] >= c->regen_min_steps;
}

int navsim_regen_cpu(const navsim_config* c, const navsim_state* st, const navsim_step_io* io) {
    if (!c || !st || !io || !io->done || !io->obs) return NAVSIM_E_ARG;
    if (c->regen_min_steps < 0 || (c->regen_min_steps > 0 && !st->done_steps)) return NAVSIM_E_ARG;
    if (c->field_format != NAVSIM_FIELD_F32 || c->map_h != c->map_w || c->n_spawn < 1) return NAVSIM_E_UNSUPPORTED;
    const int E = c->n_envs, N = c->max_peds, K = c->n_spawn, H = c->map_h, W = c->map_w;
    const int P = c->max_waypoints;
    if (c->ped_model != NAVSIM_PED_NONE && (P < 1 || P > NAVSIM_MAX_WAYPOINTS)) return NAVSIM_E_ARG;
    uint8_t* mask = (uint8_t*)calloc((size_t)E, 1);
    uint8_t* occ = (uint8_t*)malloc((size_t)H * W);
    int taken = 0;
    if (st->counters) {                          /* what this call serves, and what its cap leaves waiting */
        int all = 0, n_short = 0;
        for (int e = 0; e < E; ++e) {
            const int lng = regen_long_enough(c, st, e);
            all += io->done[e] != 0 && lng;
            n_short += io->done[e] != 0 && !lng;
        }
        const int served = all < c->regen_cap ? all : c->regen_cap;
        st->counters[NAVSIM_COUNTER_REGEN_SERVED] += (unsigned long long)served;
        st->counters[NAVSIM_COUNTER_REGEN_UNSERVED] += (unsigned long long)(all - served);
        st->counters[NAVSIM_COUNTER_REGEN_SHORT] += (unsigned long long)n_short;
    }
    for (int e = 0; e < E && taken < c->regen_cap; ++e) {
        if (!io->done[e] || !regen_long_enough(c, st, e)) continue;
        ++taken;
        mask[e] = 1;
        const uint64_t genv = (uint64_t)(c->env_index_base + e), ep = (uint64_t)st->episode[e];
        const double* tape = st->regen_draws ? st->regen_draws + (size_t)e * NAVSIM_DRAWS_PER_ARENA : NULL;
        {   /* per-episode env_param draws that are plain state (env.py:281-292, 786, 439) */
            const uint64_t pk = nvr_hash4(c->seed, genv, ep, 0x50524DULL);
            if (c->num_humans_hi > 0 && c->ped_model != NAVSIM_PED_NONE && st->n_peds) {
                int nh = c->num_humans_lo + (int)(rg_t(tape, NAVSIM_DRAW_NUM_HUMANS, pk, 1) * (double)(c->num_humans_hi - c->num_humans_lo + 1));
                st->n_peds[e] = nh > N ? N : nh;
            }
            if (c->scan_noise_std_hi >= 0.0 && st->scan_noise_std)
                st->scan_noise_std[e] = (float)(c->scan_noise_std_lo + (c->scan_noise_std_hi - c->scan_noise_std_lo) *
                                                                           rg_t(tape, NAVSIM_DRAW_SCAN_NOISE_STD, pk, 2));
        }
        float* f = (float*)st->field + (size_t)e * H * W;
        int size = W;                           /* side of the live map: cells are sampled in [0, size)^2 */
        if (c->regen_indoor_ratio > 0.0 &&       /* env.py:295: np.random.random() < indoor_ratio */
            rg_t(tape, NAVSIM_DRAW_KIND, nvr_hash4(c->seed, genv, ep, 0x4B494E44ULL), 0) < c->regen_indoor_ratio) {
            regen_map_indoor(c, genv, ep, occ, tape);
        } else {
            regen_map(c, genv, ep, occ, tape);
            size = outdoor_size(c);
        }
        navsim_build_dt_cpu(occ, 1, H, W, f);
        if (st->costmap) navsim_costmap_cpu(occ, 1, H, W, st->costmap + (size_t)e * (H / 5) * (W / 5));
        if (c->regen_plan) { regen_planned(c, st, e, genv, ep, occ, size); continue; }
        /* start / goal table */
        double* sp = (double*)st->spawn_pose + (size_t)e * K * 3;
        double* sg = (double*)st->spawn_goal + (size_t)e * K * 2;
        const double clr = c->spawn_clearance / c->resolution;
        for (int k = 0; k < K; ++k) {
            uint64_t key = nvr_hash4(c->seed, genv, ep, 0x53504157ULL + (uint64_t)k), n = 0;
            rg_sample(c, f, size, key, &n, clr, 0, 0, 0, 0, 0, &sp[3 * k], &sp[3 * k + 1]);
            sp[3 * k + 2] = NVR_TWO_PI * rg_u(key, n++);
            rg_sample(c, f, size, key, &n, clr, 2, sp[3 * k], sp[3 * k + 1], c->min_goal_dist, c->max_goal_dist,
                      &sg[2 * k], &sg[2 * k + 1]);
        }
        int idx = (int)(nvr_hash4(c->seed, genv, ep, 0x5eedULL) % (uint64_t)K);
        if (c->regen_check_discomfort)           /* env.py:776-781: first table entry from idx on whose first scan is clear */
            for (int s_ = 0; s_ < K; ++s_) {
                int j = (idx + s_) % K;
                if (!spawn_in_discomfort(c, st, e, sp + 3 * j)) { idx = j; break; }
            }
        double* rp = st->robot_pose + 3 * (size_t)e;
        rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
        st->robot_goal[2 * e] = sg[2 * idx]; st->robot_goal[2 * e + 1] = sg[2 * idx + 1];
        /* pedestrians */
        int n = (c->ped_model == NAVSIM_PED_NONE) ? 0 : st->n_peds[e];
        if (n > N) n = N;
        const double pclr = c->ped_clearance / c->resolution;
        for (int i = 0; i < n; ++i) {
            size_t q = (size_t)e * N + i;
            uint64_t key = nvr_hash4(c->seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
            double x, y, gx, gy;
            rg_sample(c, f, size, key, &m, pclr, 1, rp[0], rp[1], c->ped_min_robot_dist, 0, &x, &y);
            double th = NVR_TWO_PI * rg_u(key, m++);
            rg_sample(c, f, size, key, &m, pclr, 2, x, y, c->ped_min_goal_dist, 1.0e300, &gx, &gy);
            st->ped_pose[q * 3] = x; st->ped_pose[q * 3 + 1] = y; st->ped_pose[q * 3 + 2] = th;
            st->ped_vel[q * 2] = 0.0; st->ped_vel[q * 2 + 1] = 0.0;
            ((double*)st->ped_v_pref)[q] = c->v_pref_lo + (c->v_pref_hi - c->v_pref_lo) * rg_u(key, m++);
            ((uint8_t*)st->ped_has_legs)[q] = rg_u(key, m++) < c->has_legs_ratio;
            double* wp = st->ped_waypoints + (q * P) * 2;
            wp[0] = gx; wp[1] = gy;
            st->ped_n_waypoints[q] = 1;
            st->ped_wp_head[q] = 0;
            if (st->ped_goal) { st->ped_goal[q * 2] = gx; st->ped_goal[q * 2 + 1] = gy; }
        }
        if (st->ped_due) st->ped_due[e] = 0;
    }
    /* first observation of the new episodes (env.py:808-831); other arenas keep the row the step wrote.
     * cfg.defer_reset_scan: also of the arenas beyond the cap, which navsim_step restarted in place without scanning */
    navsim_step_io io2 = *io;
    io2.obs_prev = io->obs;
    if (c->defer_reset_scan)
        for (int e = 0; e < E; ++e) mask[e] = io->done[e] != 0;
    int rc = navsim_reset_obs_cpu(c, st, &io2, mask);
    free(mask); free(occ);
    return rc;
}

/* =========================================================================================
 * a16 + reset path: costmap (env.py:312-332), shortest 4-connected path (pyastar2d.astar_path with
 * allow_diagonal=False on uniform costs, called at env.py:343-354) and path_to_waypoints
 * (env.py:1261-1277).
 *   costmap: cv2.resize(INTER_NEAREST) by the integer factor 5 samples occ[5J][5I]; the 9x9 box
 *   filter2D (default border BORDER_REFLECT_101) followed by "> 0 -> 100" is a 4-cell dilation.
 *   path: [UPSTREAM-RECALL] pyastar2d returns a minimum-cost path including both end cells, or None
 *   when the goal is unreachable; with uniform weights every shortest path has the same length, and
 *   WHICH one A* returns depends on its heap order, which is not pinned -- BUILD-DEFINED tie-break:
 *   breadth-first distances from the goal, then from the start always step to the first neighbour in
 *   the order (+i, -i, +j, -j) that is one closer.
 *   path_to_waypoints is pinned by golden vectors (tests/golden/golden_units.npz wp_*).
 * ======================================================================================= */
#define COST_FACTOR 5

static inline int reflect101(int k, int n) {
    if (n == 1) return 0;
    while (k < 0 || k >= n) { if (k < 0) k = -k; if (k >= n) k = 2 * (n - 1) - k; }
    return k;
}

int navsim_costmap_cpu(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, uint8_t* cost) {
    if (!occ || !cost || H < COST_FACTOR || W < COST_FACTOR) return NAVSIM_E_ARG;
    const int Hc = H / COST_FACTOR, Wc = W / COST_FACTOR;
    for (int m = 0; m < n_maps; ++m) {
        const uint8_t* o = occ + (size_t)m * H * W;
        uint8_t* c = cost + (size_t)m * Hc * Wc;
        for (int J = 0; J < Hc; ++J)
            for (int I = 0; I < Wc; ++I) {
                int any = 0;
                for (int dj = -4; dj <= 4 && !any; ++dj)
                    for (int di = -4; di <= 4; ++di) {
                        int jj = reflect101(J + dj, Hc), ii = reflect101(I + di, Wc);
                        if (o[(size_t)(jj * COST_FACTOR) * W + ii * COST_FACTOR]) { any = 1; break; }
                    }
                c[(size_t)J * Wc + I] = (uint8_t)any;
            }
    }
    return NAVSIM_OK;
}

/* env.py:1261-1277 on a path of n points (x, y); returns the number of waypoints of the path.  The first max_wp of them
 * are written (the reference keeps all of them: max_wp is this build's cfg.max_waypoints); the scan continues past
 * max_wp so that the count is exact.  `start` / `len` (optional): path_distance of env.py:757-759, |start - wp0| +
 * sum |wp_k+1 - wp_k| over EVERY waypoint, stored or not. */
static int path_to_waypoints(const double* path, int n, double interval, double* wp, int max_wp,
                             const double* start, double* len) {
    int first = 0, count = 0;
    double L = 0.0, lx = start ? start[0] : 0.0, ly = start ? start[1] : 0.0;
    for (;;) {
        int found = -1;
        for (int k = first; k < n; ++k) {
            double dx = path[2 * first] - path[2 * k], dy = path[2 * first + 1] - path[2 * k + 1];
            if (sqrt(dx * dx + dy * dy) > interval) { found = k; break; }
        }
        int pick = (found >= 0) ? found : n - 1;
        if (count < max_wp) { wp[2 * count] = path[2 * pick]; wp[2 * count + 1] = path[2 * pick + 1]; }
        ++count;
        {
            const double ax = path[2 * pick] - lx, ay = path[2 * pick + 1] - ly;
            L += sqrt(ax * ax + ay * ay);
            lx = path[2 * pick]; ly = path[2 * pick + 1];
        }
        if (found < 0) break;
        first = found;
    }
    if (len) *len = L;
    return count;
}

int navsim_path_to_waypoints_cpu(const double* path, int32_t n, double interval, double* wp, int32_t max_wp) {
    if (!path || !wp || n < 1) return NAVSIM_E_ARG;
    return path_to_waypoints(path, n, interval, wp, max_wp, NULL, NULL);
}

/* n queries: query m plans on costmap map_index[m] (or m when map_index is NULL); cost [*,Hc,Wc]
 * (nonzero = blocked), start/goal [n,2] metres, cost resolution res_c.  Outputs: wp [n,max_wp,2], n_wp [n] (0 = no path), path_cells [n] (cells on the path),
 * path_len [n] = |start - wp0| + sum |wp_k+1 - wp_k| (env.py:757-759). */
int navsim_plan_cpu(const uint8_t* cost, const int32_t* map_index, int32_t n_maps, int32_t Hc, int32_t Wc,
                    double res_c, double ox, double oy, const double* start, const double* goal, double interval,
                    int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len) {
    return plan_cpu(cost, map_index, n_maps, Hc, Wc, res_c, ox, oy, start, goal, interval, max_wp, wp, n_wp, path_cells,
                    path_len, NULL);
}
/* n_total [n] (optional): waypoints of the whole path, of which min(n_total, max_wp) were stored */
static int plan_cpu(const uint8_t* cost, const int32_t* map_index, int32_t n_maps, int32_t Hc, int32_t Wc,
                    double res_c, double ox, double oy, const double* start, const double* goal, double interval,
                    int32_t max_wp, double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len, int32_t* n_total) {
    if (!cost || !start || !goal || !wp || !n_wp) return NAVSIM_E_ARG;
    navsim_config cc;
    memset(&cc, 0, sizeof(cc));
    cc.origin_x = ox; cc.origin_y = oy; cc.resolution = res_c; cc.map_h = Hc; cc.map_w = Wc;
    int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * (size_t)Hc * Wc);
    int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)Hc * Wc);
    double* path = (double*)malloc(sizeof(double) * 2 * (size_t)Hc * Wc);
    static const int DI[4] = {1, -1, 0, 0}, DJ[4] = {0, 0, 1, -1};
    for (int m = 0; m < n_maps; ++m) {
        const uint8_t* c = cost + (size_t)(map_index ? map_index[m] : m) * Hc * Wc;
        int si, sj, gi, gj;
        xy_to_ij(start[2 * m], start[2 * m + 1], &cc, &si, &sj);           /* env.py:348-349 */
        xy_to_ij(goal[2 * m], goal[2 * m + 1], &cc, &gi, &gj);
        n_wp[m] = 0;
        if (n_total) n_total[m] = 0;
        if (path_cells) path_cells[m] = 0;
        if (path_len) path_len[m] = 0.0;
        if (si >= Wc || sj >= Hc || gi >= Wc || gj >= Hc) continue;
        if (c[(size_t)sj * Wc + si] || c[(size_t)gj * Wc + gi]) continue;
        for (size_t k = 0; k < (size_t)Hc * Wc; ++k) dist[k] = -1;
        int head = 0, tail = 0;
        dist[(size_t)gj * Wc + gi] = 0;
        queue[tail++] = gj * Wc + gi;
        while (head < tail) {
            int cur = queue[head++], ci = cur % Wc, cj = cur / Wc;
            for (int d = 0; d < 4; ++d) {
                int ni = ci + DI[d], nj = cj + DJ[d];
                if (ni < 0 || ni >= Wc || nj < 0 || nj >= Hc) continue;
                size_t q = (size_t)nj * Wc + ni;
                if (c[q] || dist[q] >= 0) continue;
                dist[q] = dist[cur] + 1;
                queue[tail++] = (int)q;
            }
        }
        if (dist[(size_t)sj * Wc + si] < 0) continue;                       /* unreachable: None */
        int n = 0, ci = si, cj = sj;
        for (;;) {
            path[2 * n] = ((double)ci + 0.5) * res_c + ox;                 /* env.py:1218-1219 */
            path[2 * n + 1] = ((double)cj + 0.5) * res_c + oy;
            ++n;
            int dcur = dist[(size_t)cj * Wc + ci];
            if (dcur == 0) break;
            for (int d = 0; d < 4; ++d) {
                int ni = ci + DI[d], nj = cj + DJ[d];
                if (ni < 0 || ni >= Wc || nj < 0 || nj >= Hc) continue;
                if (dist[(size_t)nj * Wc + ni] == dcur - 1) { ci = ni; cj = nj; break; }
            }
        }
        double* w = wp + (size_t)m * max_wp * 2;
        double L = 0.0;
        int cnt = path_to_waypoints(path, n, interval, w, max_wp, start + 2 * m, &L);
        n_wp[m] = cnt < max_wp ? cnt : max_wp;
        if (n_total) n_total[m] = cnt;
        if (path_cells) path_cells[m] = n;
        if (path_len) path_len[m] = L;
    }
    free(dist); free(queue); free(path);
    return NAVSIM_OK;
}

/* TESTS ONLY: the acceptance rules of the spawn loops on SUPPLIED candidates, so that they can be compared with the
 * decisions the reference's own _sample_start_goal_path / reset() took on the same candidates
 * (tests/golden/golden_reset.npz).  The very functions the samplers above call.
 * kind[m]: 0 = the robot's pair (env.py:748-762), 1 = a pedestrian's (env.py:786-793; robot [n,2] = the robot's xy).
 * code[m]: 0 kept, 1 start closer than ped_min_robot_dist to the robot, 2 goal distance outside its interval,
 *          3 no path on the costmap, 4 (robot) path_distance > 2 |goal - start|. */
int navsim_spawn_decisions_cpu(const navsim_config* c, const uint8_t* cost, int32_t Hc, int32_t Wc, int32_t n,
                               const int32_t* kind, const double* start, const double* goal, const double* robot,
                               int32_t* code) {
    if (!c || !cost || !kind || !start || !goal || !code || n < 0) return NAVSIM_E_ARG;
    const double res_c = c->resolution * 5.0;
    double wp[2 * NAVSIM_MAX_WAYPOINTS];
    for (int m = 0; m < n; ++m) {
        const double* s = start + 2 * m;
        const double* g = goal + 2 * m;
        const int ped = kind[m] == 1;
        code[m] = 0;
        if (ped && robot && !rg_start_ok(s[0], s[1], robot[2 * m], robot[2 * m + 1], c->ped_min_robot_dist)) { code[m] = 1; continue; }
        if (!rg_goal_ok(s[0], s[1], g[0], g[1], ped ? c->ped_min_goal_dist : c->min_goal_dist,
                        ped ? 1.0e300 : c->max_goal_dist)) { code[m] = 2; continue; }
        int32_t nwp = 0; double plen = 0.0;
        navsim_plan_cpu(cost, NULL, 1, Hc, Wc, res_c, c->origin_x, c->origin_y, s, g, ped ? 2.0 : 5.0,
                        c->max_waypoints, wp, &nwp, NULL, &plen);
        if (nwp <= 0) { code[m] = 3; continue; }
        if (!ped && !rg_robot_path_ok(plen, s[0], s[1], g[0], g[1])) code[m] = 4;
    }
    return NAVSIM_OK;
}

int navsim_math_cpu(int32_t fn, const double* x, const double* x2, double* out, int32_t n) {
    if (!x || !out) return NAVSIM_E_ARG;
    for (int i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = nvr_sin(x[i]); break;
            case 1: out[i] = nvr_cos(x[i]); break;
            case 2: out[i] = nvr_atan2(x[i], x2 ? x2[i] : 1.0); break;
            case 3: out[i] = nvr_exp_neg(x[i]); break;
            case 4: out[i] = nvr_wrap_pi(x[i]); break;
            case 5: out[i] = nvr_mod_2pi(x[i]); break;
            default: return NAVSIM_E_ARG;
        }
    }
    return NAVSIM_OK;
}

/* ---------------------------------------------------------------------------------------------
 * CrowdSim-v0 termination block (crowd_sim.py:808-949), float64 in Python operator order.
 * ------------------------------------------------------------------------------------------- */
/* crowd_sim/envs/utils/utils.py:4-26 */
static double point_to_segment_dist(double x1, double y1, double x2, double y2, double x3, double y3) {
    double px = x2 - x1, py = y2 - y1;
    if (px == 0.0 && py == 0.0) { double a = x3 - x1, b = y3 - y1; return sqrt(a * a + b * b); }
    double u = ((x3 - x1) * px + (y3 - y1) * py) / (px * px + py * py);
    if (u > 1.0) u = 1.0; else if (u < 0.0) u = 0.0;
    double x = x1 + u * px, y = y1 + u * py;
    double a = x - x3, b = y - y3;
    return sqrt(a * a + b * b);
}

/* does the window of half-width `half` cells around (ix, iy) contain an occupied cell?  crowd_sim.py:843-861 */
static int crowd_window_hits(const uint8_t* m, int G, int n_cells, int ix, int iy, int half) {
    int sx = ix - half, ex = sx + half * 2, sy = iy - half, ey = sy + half * 2;
    if (sx < 0) sx = 0;
    if (ex > n_cells) ex = n_cells;
    if (sy < 0) sy = 0;
    if (ey > n_cells) ey = n_cells;
    if (!(ex > sx && ey > sy)) return 0;
    for (int x = sx; x < ex; ++x)
        for (int y = sy; y < ey; ++y)
            if (x < G && y < G && !m[(size_t)x * G + y]) return 1;
    return 0;
}

int navsim_crowd_check_cpu(const navsim_crowd_params* p, int32_t n_envs, int32_t max_agents, int32_t grid,
                           const uint8_t* free_map, const double* robot, const double* agents, const int32_t* n_agents,
                           const double* global_time, double* reward, uint8_t* done, int32_t* info, double* min_dist) {
    if (!p || !free_map || !robot || !global_time || !reward || !done || !info || n_envs < 0 || max_agents < 0 ||
        grid < 1 || (max_agents > 0 && !agents))
        return NAVSIM_E_ARG;
    const int n_cells = (int)nearbyint(p->map_size_m / p->map_resolution);
    for (int e = 0; e < n_envs; ++e) {
        const double* r = robot + (size_t)e * 10;
        const double radius = r[8];
        int na = n_agents ? n_agents[e] : max_agents;
        if (na > max_agents) na = max_agents;
        double dmin = INFINITY;
        int collision = 0;
        for (int a = 0; a < na; ++a) {                                   /* crowd_sim.py:808-826 */
            const double* g = agents + ((size_t)e * max_agents + a) * 5;
            double px = g[0] - r[0], py = g[1] - r[1];
            double vx = g[2] - r[4], vy = g[3] - r[5];
            double ex = px + vx * p->time_step, ey = py + vy * p->time_step;
            double closest = point_to_segment_dist(px, py, ex, ey, 0.0, 0.0) - g[4] - radius;
            if (closest < 0.0) { collision = 1; break; }
            else if (closest < dmin) dmin = closest;
        }
        const uint8_t* m = free_map + (size_t)e * grid * grid;
        const int ix = (int)nearbyint((r[2] + p->map_size_m / 2.0) / p->map_resolution);   /* int(round(.)) */
        const int iy = (int)nearbyint((r[3] + p->map_size_m / 2.0) / p->map_resolution);
        const int half = (int)ceil(radius / sqrt(2.0) / p->map_resolution);
        if (crowd_window_hits(m, grid, n_cells, ix, iy, half)) collision = 1;
        const int half2 = (int)ceil((radius + p->discomfort_dist) / p->map_resolution);
        const int close_to_obstacle = crowd_window_hits(m, grid, n_cells, ix, iy, half2);
        double gx = r[2] - r[6], gy = r[3] - r[7];
        const int reaching_goal = sqrt(gx * gx + gy * gy) < radius;
        double rew; int dn, code; double md = INFINITY;
        if (global_time[e] >= p->time_limit) { rew = p->timeout_penalty; dn = 1; code = NAVSIM_CROWD_TIMEOUT; }
        else if (reaching_goal) { rew = p->success_reward; dn = 1; code = NAVSIM_CROWD_REACH_GOAL; }
        else if (collision) { rew = p->collision_penalty; dn = 1; code = NAVSIM_CROWD_COLLISION; }
        else if (close_to_obstacle) { rew = -p->discomfort_penalty_factor * p->time_step * 0.1; dn = 0; code = NAVSIM_CROWD_DANGER; md = 0.1; }
        else if (dmin < p->discomfort_dist) { rew = (dmin - p->discomfort_dist) * p->discomfort_penalty_factor * p->time_step; dn = 0; code = NAVSIM_CROWD_DANGER; md = dmin; }
        else if (fabs(r[9]) > 0.0) { rew = fabs(r[9]) * p->rotation_penalty_factor; dn = 0; code = NAVSIM_CROWD_NOTHING; }
        else { rew = 0.0; dn = 0; code = NAVSIM_CROWD_NOTHING; }
        reward[e] = rew; done[e] = (uint8_t)dn; info[e] = code;
        if (min_dist) min_dist[e] = md;
    }
    return NAVSIM_OK;
}

/* =========================================================================================
 * CrowdSim-v0 local maps (nav_gym/src/crowd_sim/envs/crowd_sim.py:999-1186), SURVEY.md 8f #4.
 *
 * get_local_map_angular + calculate_angular_map_distances (crowd_sim.py:999-1102): pure NumPy / math in
 * the reference, float64, restated call by call; pinned by tests/golden/golden_crowd_maps.npz (recorded
 * from the reference's own functions).  cos / sin / atan2 are the deterministic nvr_* functions (<= 1 ulp
 * from libm): values agree with the goldens to 1e-12.
 * ======================================================================================= */
typedef struct { int idx; double x, y; } amap_seen;

static void amap_min(double* rdv, int j, double v) { if (v < rdv[j]) rdv[j] = v; }

/* calculate_angular_map_distances (crowd_sim.py:999-1053): `seen` is (rad_indeces, locations) */
static void amap_calc(const navsim_crowd_map_params* p, double vx, double vy, double ex, double ey, double ct, double st,
                      double* rdv, amap_seen* seen, int* n_seen) {
    const int dim = p->angular_dim;
    const double res = (p->angular_max - p->angular_min) / (double)dim;
    double px = (vx - ex) * ct + (vy - ey) * st;
    double py = (vy - ey) * ct - (vx - ex) * st;
    const double phi = nvr_atan2(py, px);
    const int rad_idx = (int)((phi - p->angular_min) / res);               /* int(): toward zero */
    const double distance = sqrt(px * px + py * py);
    if (rad_idx >= 0 && rad_idx < dim) amap_min(rdv, rad_idx, distance);
    for (int s = 0; s < *n_seen; ++s) {
        const int old = seen[s].idx;
        const double lx = seen[s].x, ly = seen[s].y;
        int wrapped, idx_diff;
        if ((double)abs(rad_idx - old) > NVR_PI / res) {
            wrapped = 1;
            idx_diff = (rad_idx > old) ? dim - rad_idx + old : dim - old + rad_idx;
        } else {
            wrapped = 0;
            idx_diff = abs(rad_idx - old);
        }
        for (int i = 0; i < idx_diff; ++i) {
            const double f = (double)i / (double)idx_diff;
            if ((rad_idx < old && !wrapped) || (rad_idx > old && wrapped)) {
                if (rad_idx + i >= 0 && rad_idx + i < dim) {
                    const double X = vx + f * (lx - vx) - ex, Y = vy + f * (ly - vy) - ey;
                    px = X * ct + Y * st;
                    py = Y * ct - X * st;
                    amap_min(rdv, (rad_idx + i) % dim, sqrt(px * px + py * py));
                }
            } else {
                if (old + i >= 0 && old + i < dim) {
                    const double X = lx + f * (vx - lx) - ex, Y = ly + f * (vy - ly) - ey;
                    px = X * ct + Y * st;
                    py = Y * ct - X * st;
                    amap_min(rdv, (old + i) % dim, sqrt(px * px + py * py));
                }
            }
        }
    }
    seen[*n_seen].idx = rad_idx; seen[*n_seen].x = vx; seen[*n_seen].y = vy;
    ++*n_seen;
}

/* robot [E,4] = px, py, theta, radius; verts [E, max_obst, n_vert, 2]; n_obst [E] or NULL (= max_obst);
 * out [E, angular_dim] */
int navsim_crowd_angular_map_cpu(const navsim_crowd_map_params* p, int32_t n_envs, int32_t max_obst, int32_t n_vert,
                                 const double* robot, const double* verts, const int32_t* n_obst, double* out) {
    if (!p || !robot || !out || n_envs < 0 || max_obst < 0 || n_vert < 1 || n_vert > NAVSIM_CROWD_MAX_VERTS ||
        p->angular_dim < 1 || (max_obst > 0 && !verts))
        return NAVSIM_E_ARG;
    const int dim = p->angular_dim;
    static const int s1[4] = {-1, 1, -1, 1}, s2[4] = {-1, -1, 1, 1};       /* crowd_sim.py:1073 */
    for (int e = 0; e < n_envs; ++e) {
        const double* r = robot + (size_t)e * 4;
        double* rdv = out + (size_t)e * dim;
        for (int k = 0; k < dim; ++k) rdv[k] = p->angular_max_range;
        double st, ct;
        nvr_sincos(r[2], &st, &ct);
        double edge[4][2];
        for (int k = 0; k < 4; ++k) { edge[k][0] = r[0] + s1[k] * r[3]; edge[k][1] = r[1] + s2[k] * r[3]; }
        int no = n_obst ? n_obst[e] : max_obst;
        if (no > max_obst) no = max_obst;
        amap_seen seen[NAVSIM_CROWD_MAX_VERTS > 4 ? NAVSIM_CROWD_MAX_VERTS : 4];
        for (int o = 0; o < no; ++o) {                                     /* crowd_sim.py:1077-1083 */
            const double* vv = verts + ((size_t)e * max_obst + o) * n_vert * 2;
            for (int k = 0; k < 4; ++k) {
                int ns = 0;
                for (int v = 0; v < n_vert; ++v)
                    amap_calc(p, vv[2 * v], vv[2 * v + 1], edge[k][0], edge[k][1], ct, st, rdv, seen, &ns);
            }
        }
        for (int o = 0; o < no; ++o) {                                     /* crowd_sim.py:1085-1091 */
            const double* vv = verts + ((size_t)e * max_obst + o) * n_vert * 2;
            for (int v = 0; v < n_vert; ++v) {
                int ns = 0;
                for (int k = 0; k < 4; ++k)
                    amap_calc(p, vv[2 * v], vv[2 * v + 1], edge[k][0], edge[k][1], ct, st, rdv, seen, &ns);
            }
        }
        if (p->normalize)                                                  /* crowd_sim.py:1093-1094 */
            for (int k = 0; k < dim; ++k) rdv[k] = rdv[k] / p->angular_max_range;
    }
    return NAVSIM_OK;
}

/* get_local_map + rotate_grid_around_center (crowd_sim.py:1104-1186).
 *   Window (pure Python in the reference, pinned by goldens recorded with an identity stand-in for the
 *   rotation): centre cell int(round((p + map_size_m/2) / res)) with Python's round-half-even, size
 *   int(round(submap_size_m / res)), one-sided clipping at the map border, and the reference's exclusive
 *   slice ends -- the copied block is (size - 1) cells wide, the last row / column of the grid stays 1.
 *   Rotation: cv2.getRotationMatrix2D(center = (rows/2, cols/2), angle in degrees, 1) + cv2.warpAffine(grid, M,
 *   (rows, cols), borderValue = 1) with the defaults INTER_LINEAR / BORDER_CONSTANT.  cv2 is not installed here:
 *   [UPSTREAM-RECALL, unpinned] OpenCV inverts M, walks the destination pixels with 10-bit fixed-point source
 *   coordinates (AB_BITS = 10, rounded to 1/32 pixel: INTER_BITS = 5) and blends the four neighbours with the
 *   weights (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy; a neighbour outside the source reads borderValue.  Source
 *   values are 0 / 1 and the weights multiples of 1/1024, so the blend is exact and the 0.9 threshold sharp.
 *   free_map [E,G,G] uint8 (1 = free) indexed [x][y] like CrowdSim.map; out [E,S,S] uint8 (1 = free). */
static inline int py_round_int(double v) { return (int)nearbyint(v); }     /* default rounding mode: half-even */

int navsim_crowd_local_map_cpu(const navsim_crowd_map_params* p, int32_t n_envs, int32_t grid, const uint8_t* free_map,
                               const double* robot, int32_t rotate, uint8_t* out) {
    if (!p || !free_map || !robot || !out || n_envs < 0 || grid < 1) return NAVSIM_E_ARG;
    const int S = py_round_int(p->submap_size_m / p->map_resolution);
    if (S < 1 || S > 1024) return NAVSIM_E_UNSUPPORTED;
    double* g = (double*)malloc((size_t)S * S * sizeof(double));
    for (int e = 0; e < n_envs; ++e) {
        const double* r = robot + (size_t)e * 4;
        const uint8_t* m = free_map + (size_t)e * grid * grid;
        uint8_t* o = out + (size_t)e * S * S;
        const int cx = py_round_int((r[0] + p->map_size_m / 2.0) / p->map_resolution);
        const int cy = py_round_int((r[1] + p->map_size_m / 2.0) / p->map_resolution);
        int sx = py_round_int((double)cx - floor((double)S / 2.0)), sy = py_round_int((double)cy - floor((double)S / 2.0));
        int ex = sx + S - 1, ey = sy + S - 1;
        const int mx = grid - 1, my = grid - 1;
        int gsx = 0, gsy = 0, gex = S - 1, gey = S - 1;
        if (sx < 0) { gsx = -sx; sx = 0; } else if (ex > mx) { gex = gex - (ex - mx); ex = mx; }
        if (sy < 0) { gsy = -sy; sy = 0; } else if (ey > my) { gey = gey - (ey - my); ey = my; }
        for (int k = 0; k < S * S; ++k) g[k] = 1.0;
        if (gsy > gey || sy > ey || sx > ex || gsx > gex) {                /* crowd_sim.py:1152-1154: all ones */
            for (int k = 0; k < S * S; ++k) o[k] = 1;
            continue;
        }
        for (int a = 0; a < gex - gsx && a < ex - sx; ++a)                 /* exclusive slice ends */
            for (int b = 0; b < gey - gsy && b < ey - sy; ++b)
                g[(size_t)(gsx + a) * S + (gsy + b)] = (double)m[(size_t)(sx + a) * grid + (sy + b)];
        if (!rotate) {
            for (int k = 0; k < S * S; ++k) o[k] = g[k] > 0.9;
            continue;
        }
        /* getRotationMatrix2D: angle in degrees, positive = counter-clockwise (image coordinates) */
        const double angle = (-r[2] + NVR_PI / 2.0) * 180.0 / NVR_PI;
        double sa, ca;
        nvr_sincos(angle * NVR_PI / 180.0, &sa, &ca);
        const double cxr = (double)S / 2.0, cyr = (double)S / 2.0;
        double M[6] = {ca, sa, (1.0 - ca) * cxr - sa * cyr, -sa, ca, sa * cxr + (1.0 - ca) * cyr};
        /* warpAffine: invert M (it maps source -> destination) */
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0.0 ? 1.0 / D : 0.0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        const double i0 = A11, i1 = M[1] * (-D), i3 = M[3] * (-D), i4 = A22;
        const double b1 = -i0 * M[2] - i1 * M[5], b2 = -i3 * M[2] - i4 * M[5];
        for (int y = 0; y < S; ++y) {                                      /* dst(y, x): row y, column x */
            const int X0 = (int)lrint((i1 * y + b1) * 1024.0) + 16, Y0 = (int)lrint((i4 * y + b2) * 1024.0) + 16;
            for (int x = 0; x < S; ++x) {
                const int X = ((int)lrint(i0 * x * 1024.0) + X0) >> 5, Y = ((int)lrint(i3 * x * 1024.0) + Y0) >> 5;
                const int ix = X >> 5, iy = Y >> 5;
                const double fx = (double)(X & 31) / 32.0, fy = (double)(Y & 31) / 32.0;
                double v[4];
                for (int t = 0; t < 4; ++t) {
                    const int xx = ix + (t & 1), yy = iy + (t >> 1);
                    v[t] = (xx >= 0 && xx < S && yy >= 0 && yy < S) ? g[(size_t)yy * S + xx] : 1.0;
                }
                const double val = v[0] * ((1.0 - fx) * (1.0 - fy)) + v[1] * (fx * (1.0 - fy)) + v[2] * ((1.0 - fx) * fy) +
                                   v[3] * (fx * fy);
                o[(size_t)y * S + x] = val > 0.9;
            }
        }
    }
    free(g);
    return NAVSIM_OK;
}

/* =========================================================================================
 * CrowdSim-v0 pedestrians: ORCA through rvo2 (crowd_sim/envs/policy/orca.py:85-135) + Agent.step
 * (crowd_sim/envs/utils/agent.py:108-141).  SURVEY.md 8f #4.
 *
 * [UPSTREAM-RECALL, unpinned] `rvo2` (pip pyrvo2-danieldugas, nav_gym/setup.py:26) is not installed and its source
 * is not in /root/reference.  What follows restates the published RVO2 Library 2.0 algorithm the binding wraps
 * (Agent::computeNeighbors, Agent::computeNewVelocity, linearProgram1/2/3; van den Berg et al., "Reciprocal n-body
 * collision avoidance"), float32 throughout like the library's Vector2, for the ONE agent whose velocity orca.py
 * reads back (agent 0 of a simulator it rebuilds every step: itself + the other agents in view, their preferred
 * velocity 0).  Stated differences from the library: neighbours are found by a linear scan in index order instead
 * of kd-trees (same sorted bounded lists; ties may order differently), and obstacle edges are not split the way the
 * obstacle kd-tree builder may split them -- the constraints are the same half-planes, their order can differ in
 * degenerate (infeasible) cases.  RVO_EPSILON = 0.00001f.
 * ======================================================================================= */
#define ORCA_EPS 0.00001f
#define ORCA_MAX_LINES (NAVSIM_ORCA_MAX_EDGES + NAVSIM_ORCA_MAX_AGENTS)
typedef struct { float x, y; } ov2;
typedef struct { ov2 point, direction; } oline;
static inline ov2 ov(float x, float y) { ov2 r = {x, y}; return r; }
static inline ov2 oadd(ov2 a, ov2 b) { return ov(a.x + b.x, a.y + b.y); }
static inline ov2 osub(ov2 a, ov2 b) { return ov(a.x - b.x, a.y - b.y); }
static inline ov2 omul(float s, ov2 a) { return ov(s * a.x, s * a.y); }
static inline ov2 odiv(ov2 a, float s) { const float inv = 1.0f / s; return ov(a.x * inv, a.y * inv); }   /* Vector2::operator/ */
static inline ov2 oneg(ov2 a) { return ov(-a.x, -a.y); }
static inline float odot(ov2 a, ov2 b) { return a.x * b.x + a.y * b.y; }
static inline float odet(ov2 a, ov2 b) { return a.x * b.y - a.y * b.x; }
static inline float oabssq(ov2 a) { return odot(a, a); }
static inline float oabs(ov2 a) { return sqrtf(odot(a, a)); }
static inline ov2 onorm(ov2 a) { return odiv(a, oabs(a)); }
static inline float osqr(float a) { return a * a; }

static int orca_lp1(const oline* lines, int lineNo, float radius, ov2 optVelocity, int directionOpt, ov2* result) {
    const float dotProduct = odot(lines[lineNo].point, lines[lineNo].direction);
    const float discriminant = osqr(dotProduct) + osqr(radius) - oabssq(lines[lineNo].point);
    if (discriminant < 0.0f) return 0;
    const float sqrtDiscriminant = sqrtf(discriminant);
    float tLeft = -dotProduct - sqrtDiscriminant;
    float tRight = -dotProduct + sqrtDiscriminant;
    for (int i = 0; i < lineNo; ++i) {
        const float denominator = odet(lines[lineNo].direction, lines[i].direction);
        const float numerator = odet(lines[i].direction, osub(lines[lineNo].point, lines[i].point));
        if (fabsf(denominator) <= ORCA_EPS) {
            if (numerator < 0.0f) return 0;
            continue;
        }
        const float t = numerator / denominator;
        if (denominator >= 0.0f) tRight = tRight < t ? tRight : t;
        else                     tLeft = tLeft > t ? tLeft : t;
        if (tLeft > tRight) return 0;
    }
    if (directionOpt) {
        if (odot(optVelocity, lines[lineNo].direction) > 0.0f) *result = oadd(lines[lineNo].point, omul(tRight, lines[lineNo].direction));
        else                                                  *result = oadd(lines[lineNo].point, omul(tLeft, lines[lineNo].direction));
    } else {
        const float t = odot(lines[lineNo].direction, osub(optVelocity, lines[lineNo].point));
        if (t < tLeft)       *result = oadd(lines[lineNo].point, omul(tLeft, lines[lineNo].direction));
        else if (t > tRight) *result = oadd(lines[lineNo].point, omul(tRight, lines[lineNo].direction));
        else                 *result = oadd(lines[lineNo].point, omul(t, lines[lineNo].direction));
    }
    return 1;
}

static int orca_lp2(const oline* lines, int n, float radius, ov2 optVelocity, int directionOpt, ov2* result) {
    if (directionOpt) *result = omul(radius, optVelocity);
    else if (oabssq(optVelocity) > osqr(radius)) *result = omul(radius, onorm(optVelocity));
    else *result = optVelocity;
    for (int i = 0; i < n; ++i) {
        if (odet(lines[i].direction, osub(lines[i].point, *result)) > 0.0f) {
            const ov2 temp = *result;
            if (!orca_lp1(lines, i, radius, optVelocity, directionOpt, result)) { *result = temp; return i; }
        }
    }
    return n;
}

static void orca_lp3(const oline* lines, int n, int numObstLines, int beginLine, float radius, ov2* result) {
    float distance = 0.0f;
    oline proj[ORCA_MAX_LINES];
    for (int i = beginLine; i < n; ++i) {
        if (odet(lines[i].direction, osub(lines[i].point, *result)) > distance) {
            int np = 0;
            for (int j = 0; j < numObstLines; ++j) proj[np++] = lines[j];
            for (int j = numObstLines; j < i; ++j) {
                oline line;
                const float determinant = odet(lines[i].direction, lines[j].direction);
                if (fabsf(determinant) <= ORCA_EPS) {
                    if (odot(lines[i].direction, lines[j].direction) > 0.0f) continue;
                    line.point = omul(0.5f, oadd(lines[i].point, lines[j].point));
                } else {
                    line.point = oadd(lines[i].point, omul(odet(lines[j].direction, osub(lines[i].point, lines[j].point)) / determinant,
                                                           lines[i].direction));
                }
                line.direction = onorm(osub(lines[j].direction, lines[i].direction));
                proj[np++] = line;
            }
            const ov2 temp = *result;
            if (orca_lp2(proj, np, radius, ov(-lines[i].direction.y, lines[i].direction.x), 1, result) < np) *result = temp;
            distance = odet(lines[i].direction, osub(lines[i].point, *result));
        }
    }
}

/* one processed obstacle vertex (RVO2 Obstacle): point, unit direction to the next vertex, convexity, links */
typedef struct { ov2 point, unitDir; int isConvex, next, prev; } oobst;

/* agents [Q, A, 6] float64: px, py, vx, vy, radius, max_speed (agent 0 = the agent whose velocity is computed);
 * n_agents [Q] or NULL (= A); pref_vel [Q,2]; obstacle polygons verts [S, O, V, 2] (counter-clockwise), n_obst [S] or
 * NULL, obst_set [Q] index of the polygon set of query q (NULL: set 0 for all); theta [Q] heading of agent 0.
 * out_vel [Q,2] = sim.getAgentVelocity(0) after doStep(); out_action [Q,2] = ActionRot(v, r) of orca.py:128-130
 * (may be NULL). */
int navsim_crowd_orca_cpu(const navsim_orca_params* p, int32_t n_queries, int32_t max_agents, const double* agents,
                          const int32_t* n_agents, const double* pref_vel, int32_t max_obst, int32_t n_vert,
                          const double* verts, const int32_t* n_obst, const int32_t* obst_set, const double* theta,
                          double* out_vel, double* out_action) {
    if (!p || !agents || !pref_vel || !out_vel || n_queries < 0 || max_agents < 1 || max_agents > NAVSIM_ORCA_MAX_AGENTS ||
        max_obst < 0 || n_vert < 2 || (size_t)max_obst * n_vert > NAVSIM_ORCA_MAX_EDGES || (max_obst > 0 && !verts))
        return NAVSIM_E_ARG;
    for (int q = 0; q < n_queries; ++q) {
        const double* ag = agents + (size_t)q * max_agents * 6;
        int na = n_agents ? n_agents[q] : max_agents;
        if (na > max_agents) na = max_agents;
        if (na < 1) { out_vel[2 * q] = 0.0; out_vel[2 * q + 1] = 0.0; continue; }
        const ov2 position = ov((float)ag[0], (float)ag[1]), velocity = ov((float)ag[2], (float)ag[3]);
        const float radius = (float)ag[4], maxSpeed = (float)ag[5];
        const ov2 prefVelocity = ov((float)pref_vel[2 * q], (float)pref_vel[2 * q + 1]);
        /* ---- obstacles of this query's set (Simulator::addObstacle) */
        oobst ob[NAVSIM_ORCA_MAX_EDGES];
        int nob = 0;
        const int set = obst_set ? obst_set[q] : 0;
        int no = max_obst ? (n_obst ? n_obst[set] : max_obst) : 0;
        if (no > max_obst) no = max_obst;
        for (int o = 0; o < no; ++o) {
            const double* vv = verts + (((size_t)set * max_obst + o) * n_vert) * 2;
            const int base = nob;
            for (int i = 0; i < n_vert; ++i) {
                const int in = (i + 1) % n_vert, ip = (i + n_vert - 1) % n_vert;
                oobst* t = &ob[nob++];
                t->point = ov((float)vv[2 * i], (float)vv[2 * i + 1]);
                const ov2 pn = ov((float)vv[2 * in], (float)vv[2 * in + 1]), pp = ov((float)vv[2 * ip], (float)vv[2 * ip + 1]);
                t->unitDir = onorm(osub(pn, t->point));
                t->isConvex = (n_vert == 2) ? 1 : (odet(osub(pp, pn), osub(t->point, pp)) >= 0.0f);   /* leftOf(prev, this, next) */
                t->next = base + in; t->prev = base + ip;
            }
        }
        /* ---- Agent::computeNeighbors: obstacle edges within range, agent on their right side, sorted by distance */
        int obn[NAVSIM_ORCA_MAX_EDGES]; float obd[NAVSIM_ORCA_MAX_EDGES]; int nobn = 0;
        {
            const float rangeSq = osqr(p->time_horizon_obst * maxSpeed + radius);
            for (int k = 0; k < nob; ++k) {
                const ov2 a = ob[k].point, b = ob[ob[k].next].point;
                const float agentLeftOfLine = odet(osub(a, position), osub(b, a));          /* leftOf(a, b, position) */
                const float distSqLine = osqr(agentLeftOfLine) / oabssq(osub(b, a));
                if (!(distSqLine < rangeSq) || !(agentLeftOfLine < 0.0f)) continue;
                const float r = odot(osub(position, a), osub(b, a)) / oabssq(osub(b, a));    /* distSqPointLineSegment */
                float distSq;
                if (r < 0.0f) distSq = oabssq(osub(position, a));
                else if (r > 1.0f) distSq = oabssq(osub(position, b));
                else distSq = oabssq(osub(position, oadd(a, omul(r, osub(b, a)))));
                if (distSq < rangeSq) {
                    int i = nobn++;
                    while (i != 0 && distSq < obd[i - 1]) { obn[i] = obn[i - 1]; obd[i] = obd[i - 1]; --i; }
                    obn[i] = k; obd[i] = distSq;
                }
            }
        }
        int agn[NAVSIM_ORCA_MAX_AGENTS]; float agd[NAVSIM_ORCA_MAX_AGENTS]; int nagn = 0;
        if (p->max_neighbors > 0) {
            float rangeSq = osqr(p->neighbor_dist);
            const int maxN = p->max_neighbors < NAVSIM_ORCA_MAX_AGENTS ? p->max_neighbors : NAVSIM_ORCA_MAX_AGENTS;
            for (int k = 1; k < na; ++k) {
                const ov2 op = ov((float)ag[6 * k], (float)ag[6 * k + 1]);
                const float distSq = oabssq(osub(position, op));
                if (distSq < rangeSq) {
                    if (nagn < maxN) ++nagn;
                    int i = nagn - 1;
                    while (i != 0 && distSq < agd[i - 1]) { agn[i] = agn[i - 1]; agd[i] = agd[i - 1]; --i; }
                    agn[i] = k; agd[i] = distSq;
                    if (nagn == maxN) rangeSq = agd[nagn - 1];
                }
            }
        }
        /* ---- Agent::computeNewVelocity */
        oline lines[ORCA_MAX_LINES];
        int nl = 0;
        const float invTimeHorizonObst = 1.0f / p->time_horizon_obst;
        for (int i = 0; i < nobn; ++i) {
            int o1 = obn[i], o2 = ob[o1].next;
            const ov2 rel1 = osub(ob[o1].point, position), rel2 = osub(ob[o2].point, position);
            int covered = 0;
            for (int j = 0; j < nl; ++j)
                if (odet(osub(omul(invTimeHorizonObst, rel1), lines[j].point), lines[j].direction) - invTimeHorizonObst * radius >= -ORCA_EPS &&
                    odet(osub(omul(invTimeHorizonObst, rel2), lines[j].point), lines[j].direction) - invTimeHorizonObst * radius >= -ORCA_EPS) {
                    covered = 1; break;
                }
            if (covered) continue;
            const float distSq1 = oabssq(rel1), distSq2 = oabssq(rel2), radiusSq = osqr(radius);
            const ov2 obstacleVector = osub(ob[o2].point, ob[o1].point);
            const float s = odot(oneg(rel1), obstacleVector) / oabssq(obstacleVector);
            const float distSqLine = oabssq(osub(oneg(rel1), omul(s, obstacleVector)));
            oline line;
            if (s < 0.0f && distSq1 <= radiusSq) {                        /* collision with the left vertex */
                if (ob[o1].isConvex) { line.point = ov(0.0f, 0.0f); line.direction = onorm(ov(-rel1.y, rel1.x)); lines[nl++] = line; }
                continue;
            } else if (s > 1.0f && distSq2 <= radiusSq) {                  /* collision with the right vertex */
                if (ob[o2].isConvex && odet(rel2, ob[o2].unitDir) >= 0.0f) {
                    line.point = ov(0.0f, 0.0f); line.direction = onorm(ov(-rel2.y, rel2.x)); lines[nl++] = line;
                }
                continue;
            } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) {  /* collision with the segment */
                line.point = ov(0.0f, 0.0f); line.direction = oneg(ob[o1].unitDir); lines[nl++] = line;
                continue;
            }
            ov2 leftLeg, rightLeg;
            if (s < 0.0f && distSqLine <= radiusSq) {                      /* viewed obliquely: the left vertex defines the VO */
                if (!ob[o1].isConvex) continue;
                o2 = o1;
                const float leg1 = sqrtf(distSq1 - radiusSq);
                leftLeg = odiv(ov(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), distSq1);
                rightLeg = odiv(ov(rel1.x * leg1 + rel1.y * radius, -rel1.x * radius + rel1.y * leg1), distSq1);
            } else if (s > 1.0f && distSqLine <= radiusSq) {               /* ... the right vertex */
                if (!ob[o2].isConvex) continue;
                o1 = o2;
                const float leg2 = sqrtf(distSq2 - radiusSq);
                leftLeg = odiv(ov(rel2.x * leg2 - rel2.y * radius, rel2.x * radius + rel2.y * leg2), distSq2);
                rightLeg = odiv(ov(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), distSq2);
            } else {                                                       /* usual situation */
                if (ob[o1].isConvex) {
                    const float leg1 = sqrtf(distSq1 - radiusSq);
                    leftLeg = odiv(ov(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), distSq1);
                } else leftLeg = oneg(ob[o1].unitDir);
                if (ob[o2].isConvex) {
                    const float leg2 = sqrtf(distSq2 - radiusSq);
                    rightLeg = odiv(ov(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), distSq2);
                } else rightLeg = ob[o1].unitDir;
            }
            const int leftNeighbor = ob[o1].prev;
            int leftForeign = 0, rightForeign = 0;
            if (ob[o1].isConvex && odet(leftLeg, oneg(ob[leftNeighbor].unitDir)) >= 0.0f) { leftLeg = oneg(ob[leftNeighbor].unitDir); leftForeign = 1; }
            if (ob[o2].isConvex && odet(rightLeg, ob[o2].unitDir) <= 0.0f) { rightLeg = ob[o2].unitDir; rightForeign = 1; }
            const ov2 leftCutoff = omul(invTimeHorizonObst, osub(ob[o1].point, position));
            const ov2 rightCutoff = omul(invTimeHorizonObst, osub(ob[o2].point, position));
            const ov2 cutoffVec = osub(rightCutoff, leftCutoff);
            const float t = (o1 == o2) ? 0.5f : odot(osub(velocity, leftCutoff), cutoffVec) / oabssq(cutoffVec);
            const float tLeft = odot(osub(velocity, leftCutoff), leftLeg);
            const float tRight = odot(osub(velocity, rightCutoff), rightLeg);
            if ((t < 0.0f && tLeft < 0.0f) || (o1 == o2 && tLeft < 0.0f && tRight < 0.0f)) {    /* left cut-off circle */
                const ov2 unitW = onorm(osub(velocity, leftCutoff));
                line.direction = ov(unitW.y, -unitW.x);
                line.point = oadd(leftCutoff, omul(radius * invTimeHorizonObst, unitW));
                lines[nl++] = line;
                continue;
            } else if (t > 1.0f && tRight < 0.0f) {                                              /* right cut-off circle */
                const ov2 unitW = onorm(osub(velocity, rightCutoff));
                line.direction = ov(unitW.y, -unitW.x);
                line.point = oadd(rightCutoff, omul(radius * invTimeHorizonObst, unitW));
                lines[nl++] = line;
                continue;
            }
            const float distSqCutoff = (t < 0.0f || t > 1.0f || o1 == o2) ? INFINITY : oabssq(osub(velocity, oadd(leftCutoff, omul(t, cutoffVec))));
            const float distSqLeft = (tLeft < 0.0f) ? INFINITY : oabssq(osub(velocity, oadd(leftCutoff, omul(tLeft, leftLeg))));
            const float distSqRight = (tRight < 0.0f) ? INFINITY : oabssq(osub(velocity, oadd(rightCutoff, omul(tRight, rightLeg))));
            if (distSqCutoff <= distSqLeft && distSqCutoff <= distSqRight) {                     /* cut-off line */
                line.direction = oneg(ob[o1].unitDir);
                line.point = oadd(leftCutoff, omul(radius * invTimeHorizonObst, ov(-line.direction.y, line.direction.x)));
                lines[nl++] = line;
            } else if (distSqLeft <= distSqRight) {                                              /* left leg */
                if (leftForeign) continue;
                line.direction = leftLeg;
                line.point = oadd(leftCutoff, omul(radius * invTimeHorizonObst, ov(-line.direction.y, line.direction.x)));
                lines[nl++] = line;
            } else {                                                                             /* right leg */
                if (rightForeign) continue;
                line.direction = oneg(rightLeg);
                line.point = oadd(rightCutoff, omul(radius * invTimeHorizonObst, ov(-line.direction.y, line.direction.x)));
                lines[nl++] = line;
            }
        }
        const int numObstLines = nl;
        const float invTimeHorizon = 1.0f / p->time_horizon;
        for (int i = 0; i < nagn; ++i) {
            const double* o = ag + 6 * agn[i];
            const ov2 relativePosition = osub(ov((float)o[0], (float)o[1]), position);
            const ov2 relativeVelocity = osub(velocity, ov((float)o[2], (float)o[3]));
            const float distSq = oabssq(relativePosition);
            const float combinedRadius = radius + (float)o[4];
            const float combinedRadiusSq = osqr(combinedRadius);
            oline line;
            ov2 u;
            if (distSq > combinedRadiusSq) {
                const ov2 w = osub(relativeVelocity, omul(invTimeHorizon, relativePosition));
                const float wLengthSq = oabssq(w);
                const float dotProduct1 = odot(w, relativePosition);
                if (dotProduct1 < 0.0f && osqr(dotProduct1) > combinedRadiusSq * wLengthSq) {    /* cut-off circle */
                    const float wLength = sqrtf(wLengthSq);
                    const ov2 unitW = odiv(w, wLength);
                    line.direction = ov(unitW.y, -unitW.x);
                    u = omul(combinedRadius * invTimeHorizon - wLength, unitW);
                } else {                                                                          /* legs */
                    const float leg = sqrtf(distSq - combinedRadiusSq);
                    if (odet(relativePosition, w) > 0.0f)
                        line.direction = odiv(ov(relativePosition.x * leg - relativePosition.y * combinedRadius,
                                                 relativePosition.x * combinedRadius + relativePosition.y * leg), distSq);
                    else
                        line.direction = oneg(odiv(ov(relativePosition.x * leg + relativePosition.y * combinedRadius,
                                                      -relativePosition.x * combinedRadius + relativePosition.y * leg), distSq));
                    const float dotProduct2 = odot(relativeVelocity, line.direction);
                    u = osub(omul(dotProduct2, line.direction), relativeVelocity);
                }
            } else {                                                                              /* already colliding */
                const float invTimeStep = 1.0f / p->time_step;
                const ov2 w = osub(relativeVelocity, omul(invTimeStep, relativePosition));
                const float wLength = oabs(w);
                const ov2 unitW = odiv(w, wLength);
                line.direction = ov(unitW.y, -unitW.x);
                u = omul(combinedRadius * invTimeStep - wLength, unitW);
            }
            line.point = oadd(velocity, omul(0.5f, u));
            lines[nl++] = line;
        }
        ov2 newVelocity;
        const int lineFail = orca_lp2(lines, nl, maxSpeed, prefVelocity, 0, &newVelocity);
        if (lineFail < nl) orca_lp3(lines, nl, numObstLines, lineFail, maxSpeed, &newVelocity);
        out_vel[2 * q] = (double)newVelocity.x; out_vel[2 * q + 1] = (double)newVelocity.y;
        if (out_action) {                                                  /* orca.py:128-130 */
            const double vx = (double)newVelocity.x, vy = (double)newVelocity.y;
            out_action[2 * q] = sqrt(vx * vx + vy * vy);
            out_action[2 * q + 1] = nvr_atan2(vy, vx) - (theta ? theta[q] : 0.0);
        }
    }
    return NAVSIM_OK;
}

/* Agent.step with an ActionRot (crowd_sim/envs/utils/agent.py:108-141): pose [n,3] in/out, action [n,2] = (v, r),
 * vel [n,2] out */
int navsim_crowd_agent_step_cpu(double* pose, const double* action, double* vel, int32_t n, double time_step) {
    if (!pose || !action || n < 0) return NAVSIM_E_ARG;
    for (int i = 0; i < n; ++i) {
        double* ps = pose + 3 * i;
        const double v = action[2 * i], r = action[2 * i + 1];
        const double theta = ps[2] + r;
        double s, c;
        nvr_sincos(theta, &s, &c);
        ps[0] = ps[0] + c * v * time_step;
        ps[1] = ps[1] + s * v * time_step;
        if (vel) { vel[2 * i] = v * c; vel[2 * i + 1] = v * s; }
        ps[2] = nvr_mod_2pi(ps[2] + r);
    }
    return NAVSIM_OK;
}
