// navsim_device.hpp -- device-side building blocks of the batched NavGym step (gfx950).
//
// Every float32 / float64 expression below is written in the exact operation order the
// specification in DESIGN.md section 3 gives (the CPU oracle implements the same specification
// independently); the translation unit is compiled with -ffp-contract=off so that v_mul + v_add
// pairs are never fused.  Reference call sites are cited per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/navsim.h"
#include "navmath.hpp"

namespace nv {

#pragma clang fp contract(off)

constexpr float kLegRadius = 0.03f;   // CSimAgent leg radius (CMap2D, called from env.py:402)

// np.linspace(angle_min, angle_max - angle_increment, n)[k]   (env.py:388-390)
__device__ __forceinline__ double linspace_k(const navsim_config& c, int k, double step) {
    if (c.n_beams == 1) return c.angle_min;
    if (k == c.n_beams - 1) return c.angle_last;
    return (double)k * step + c.angle_min;
}
__device__ __forceinline__ double linspace_step(const navsim_config& c) {
    return (c.n_beams > 1) ? (c.angle_last - c.angle_min) / (double)(c.n_beams - 1) : 0.0;
}

// beam direction: sin/cos of the float32 heading evaluated in double, rounded once
__device__ __forceinline__ void beam_dir(float heading, float& dx, float& dy) {
    double s, c;
    sincos((double)heading, s, c);
    dx = (float)c;
    dy = (float)s;
}

// Same direction from the beam table: cos/sin(lin_k + th + delta) by one angle addition, accepted only when the
// float32 rounding is PROVABLY the one beam_dir() produces.  Error budget of the candidate against the true cos/sin
// of the float32 heading (all values at most 1 in magnitude, 1 ulp = 1.1e-16): the table entries and cos/sin(th) are
// within 1 ulp; C = fma(ct, cT, -(st sT)) inherits 2.2e-16 per product and adds one rounding each: <= 6.6e-16; the
// second-order series in delta (|delta| <= 5e-7) is exact to 2e-20 and adds two fma roundings: <= 8.8e-16 in all;
// beam_dir() itself is within 1.1e-16.  So both round alike whenever the candidate sits more than 3e-15 inside its
// float32 rounding interval; otherwise the caller falls back to beam_dir() -- a few beams in 1e7.
//
// The interval: f = fl32(v) and r = v - f (exact).  The float32 neighbours of v are spaced 2^(e-23) apart with e the
// binary exponent of v ITSELF (if f rounded up to the power of two above, the spacing on v's side is still that
// of v's binade), so v is half_ulp - |r| away from the nearest rounding boundary, half_ulp = 2^(e-24), built from
// v's exponent field.  A component below 2^-23 has half_ulp < 3e-15 and never passes; a zero exponent field (v = 0)
// makes half_ulp a negative number and does not pass either: those beams go to beam_dir().
__device__ __forceinline__ bool round_if_safe(double v, float& out) {
    const float f = (float)v;
    const double r = v - (double)f;
    const int hi = __double2hiint(v);
    const double half_ulp = __hiloint2double((hi & 0x7FF00000) - (24 << 20), 0);
    out = f;
    return half_ulp - __builtin_fabs(r) >= 3.0e-15;
}
__device__ __forceinline__ bool beam_dir_from_table(double ct, double st, double cT, double sT, double delta,
                                                    float& dx, float& dy) {
    const double C = __builtin_fma(ct, cT, -(st * sT));
    const double S = __builtin_fma(st, cT, ct * sT);
    const double h = (0.5 * delta) * delta;
    const double c = __builtin_fma(-S, delta, __builtin_fma(-C, h, C));
    const double s = __builtin_fma(C, delta, __builtin_fma(-S, h, S));
    return round_if_safe(c, dx) & round_if_safe(s, dy);
}

// range_libc RayMarching::calc_range (PyRayMarching.calc_range_many, env.py:425): sphere tracing
// through the distance field, float32, C truncation of the sample position.
__device__ __forceinline__ float trace_ray(const float* __restrict__ f, int H, int W, float x0, float y0,
                                           float dx, float dy, float max_range, int march_rule) {
    float t = 0.0f;
    const bool fma_pos = march_rule == NAVSIM_MARCH_F32_FMA;            // include/navsim.h: the contracted sample position
    while (t < max_range) {
        float fx = fma_pos ? __builtin_fmaf(dx, t, x0) : x0 + dx * t;
        float fy = fma_pos ? __builtin_fmaf(dy, t, y0) : y0 + dy * t;
        int px = (int)fx;
        int py = (int)fy;
        if (px >= W || px < 0 || py < 0 || py >= H) return max_range;
        float d = f[(size_t)py * W + px];
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

// CMap2D.render_contours_in_lidar inner test (env.py:431): ray / segment, min-merge
__device__ __forceinline__ void seg_merge(float& r, float ox, float oy, float c, float s,
                                          float px, float py, float qx, float qy) {
    float ex = qx - px, ey = qy - py;
    float wx = px - ox, wy = py - oy;
    float denom = c * ey - s * ex;
    if (denom == 0.0f) return;
    float t = (wx * ey - wy * ex) / denom;
    float u = (wx * s - wy * c) / denom;
    if (t >= 0.0f && u >= 0.0f && u <= 1.0f && t < r) r = t;
}

// CMap2D.render_agents_in_lidar inner test (env.py:432): ray / leg disc first hit, min-merge
__device__ __forceinline__ void circle_merge(float& r, float ox, float oy, float c, float s,
                                             float cx, float cy, float rad) {
    float wx = cx - ox, wy = cy - oy;
    float b = wx * c + wy * s;
    float x = wx * s - wy * c;
    float disc = rad * rad - x * x;
    if (disc < 0.0f) return;
    float t = b - sqrtf(disc);
    if (t >= 0.0f && t < r) r = t;
}

// CSimAgent leg centres (CMap2D; env.py:399-402): a = pos(3) dist(3) as float32 values
__device__ __forceinline__ void leg_centres(float apx, float apy, float ath, float adx, float ady,
                                            float adth, float out[4]) {
    double px = (double)apx, py = (double)apy, th = (double)ath;
    double dxx = (double)adx, dyy = (double)ady, dth = (double)adth;
    double sf, cf, ss, cs;
    sincos(dxx * 2.0 / 0.3 + dth, sf, cf);
    sincos(dyy * 2.0 / 0.1 + dth, ss, cs);
    double front = 0.3 * cf;
    double side = 0.1 * cs;
    double s, c;
    sincos(th, s, c);
    double rx = front, ry = side + 0.1;
    double lx = -front, ly = -side - 0.1;
    out[0] = (float)((c * rx - s * ry) + px);
    out[1] = (float)((s * rx + c * ry) + py);
    out[2] = (float)((c * lx - s * ly) + px);
    out[3] = (float)((s * lx + c * ly) + py);
}

// Human.set_vel (human.py:32-41) / KetiRobot.set_vel (keti_robot.py:64-93), `off` = axle offset.
// set_vel_with: the same update with (s0, c0) = sincos(p[2]) and (s1, c1) = sincos(p[2] + w dt) supplied (the fused step
// evaluates its three sincos of phase 0 on three lanes at once)
__device__ __forceinline__ void set_vel_with(double p[3], double v, double w, double dt, double off, double s0, double c0,
                                             double s1, double c1, double* vel) {
    if (vel) { vel[0] = v * c0; vel[1] = v * s0; }
    double rx = p[0] + off * c0;
    double ry = p[1] + off * s0;
    rx = rx + c1 * v * dt;
    ry = ry + s1 * v * dt;
    p[0] = rx + (-off) * c1;
    p[1] = ry + (-off) * s1;
    p[2] = mod_2pi(p[2] + w * dt);
}
__device__ __forceinline__ void set_vel(double p[3], double v, double w, double dt, double off,
                                        double* vel) {
    double s0, c0, s1, c1;
    sincos(p[2], s0, c0);
    double th = p[2] + w * dt;
    sincos(th, s1, c1);
    set_vel_with(p, v, w, dt, off, s0, c0, s1, c1, vel);
}

// batch_xy_to_ij (env.py:1228-1253), float64 inputs
__device__ __forceinline__ void xy_to_ij(double x, double y, const navsim_config& c, int& i, int& j) {
    float fi = (float)((x - c.origin_x) / c.resolution);
    float fj = (float)((y - c.origin_y) / c.resolution);
    if (fi >= (float)c.map_h) fi = (float)(c.map_h - 1);
    if (fj >= (float)c.map_w) fj = (float)(c.map_w - 1);
    if (fi < 0.0f) fi = 0.0f;
    if (fj < 0.0f) fj = 0.0f;
    i = (int)fi;
    j = (int)fj;
}
// same with float32 inputs (the scan origin, env.py:419): float32 arithmetic (NumPy >= 2 scalars)
__device__ __forceinline__ void xy_to_ij_f32(float x, float y, const navsim_config& c, int& i, int& j) {
    float fi = (x - (float)c.origin_x) / (float)c.resolution;
    float fj = (y - (float)c.origin_y) / (float)c.resolution;
    if (fi >= (float)c.map_h) fi = (float)(c.map_h - 1);
    if (fj >= (float)c.map_w) fj = (float)(c.map_w - 1);
    if (fi < 0.0f) fi = 0.0f;
    if (fj < 0.0f) fj = 0.0f;
    i = (int)fi;
    j = (int)fj;
}

// _update_dist_travelled (env.py:237-255) for one pedestrian
__device__ __forceinline__ void leg_odometry(const double pose[3], const double vel[2], double prev_yaw,
                                             double dt, double dist[3]) {
    double vrot = (pose[2] - prev_yaw) / dt;
    double s, c;
    sincos(-pose[2], s, c);
    double bx = c * vel[0] - s * vel[1];
    double by = s * vel[0] + c * vel[1];
    dist[0] += bx * dt;
    dist[1] += by * dt;
    dist[2] += vrot * dt;
}

// scalar part of compute_rewards / compute_terminals / compute_info (env.py:464-589)
struct RewardOut { double reward; int done; float success, crash; double distance; };

__device__ __forceinline__ RewardOut reward_scalar(const navsim_config& c, const double prev_xy[2],
                                                   const double pose[2], const double vel[2],
                                                   const double goal[2], int crash, int discomfort,
                                                   double ratio_min) {
    RewardOut o;
    double dx = goal[0] - pose[0], dy = goal[1] - pose[1];
    double distance = sqrt(dx * dx + dy * dy);
    double px = goal[0] - prev_xy[0], py = goal[1] - prev_xy[1];
    double prev_distance = sqrt(px * px + py * py);
    int success = distance < c.distance_threshold;
    if (crash) discomfort = 0;
    double r_success = success ? 1.0 * c.reward_success_factor * c.reward_scale : 0.0;
    double r_crash = crash ? -1.0 * c.reward_crash_factor * c.reward_scale : 0.0;
    double r_progress = (prev_distance - distance) * c.reward_progress_factor * c.reward_scale;
    double r_forward = vel[0] * c.reward_forward_factor * c.reward_scale;
    double r_rotation = -1.0 * (vel[1] * vel[1]) * c.reward_rotation_factor * c.reward_scale;
    double r_discomfort = discomfort ? -(1.0 - ratio_min) * c.reward_discomfort_factor * c.reward_scale : 0.0;
    o.reward = r_success + r_crash + r_progress + r_forward + r_rotation + r_discomfort;
    o.done = success || crash;
    o.success = (float)success;
    o.crash = (float)crash;
    o.distance = distance;
    return o;
}

// discomfort ratio of one beam (env.py:563-567): float64 numerator over a float32 denominator
__device__ __forceinline__ double discomfort_ratio(double s, float thr, float dthr) {
    float den = (dthr - thr) + 1e-6f;
    return (s - (double)thr) / (double)den;
}

// Gaussian scan noise (env.py:437-440).  Counter-based -- keyed by (seed, global arena, episode / step / scan, beam)
// -- so results do not depend on the launch geometry; not bit-comparable with numpy's global Mersenne stream
// (tested statistically).  noise_stream() is the per-scan part of the key, evaluated once per thread.
__device__ __forceinline__ uint64_t noise_stream(uint64_t seed, uint64_t genv, uint64_t scan_key) {
    return hash4(seed ^ scan_key, genv, scan_key, 0x6E6F697365ULL);
}
__device__ __forceinline__ float gauss_noise(uint64_t stream, uint32_t beam) {
    // two 32-bit finalisers (lowbias32) on (stream word ^ beam * golden ratio): four 32-bit multiplies per beam
    // instead of the four 64-bit ones of a mix64 -- the per-beam noise was 6 % of the c2 step
    uint32_t a = (uint32_t)stream ^ (beam * 0x9E3779B9u);
    a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16;
    uint32_t b = (uint32_t)(stream >> 32) ^ a;
    b ^= b >> 16; b *= 0x7FEB352Du; b ^= b >> 15;
    float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);              // (0, 1]
    float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);   // noise: 1-ulp sqrt is plenty
}

}  // namespace nv
