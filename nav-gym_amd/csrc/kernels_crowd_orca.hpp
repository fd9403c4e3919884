// kernels_crowd_orca.hpp -- CrowdSim-v0 pedestrians: ORCA (rvo2 restated) and Agent.step.
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace).
// Specification: oracle/navsim_ref.c navsim_crowd_orca_cpu (RVO2 Library 2.0: Agent::computeNeighbors,
// Agent::computeNewVelocity, linearProgram1/2/3; float32 like the library's Vector2; unpinned -- rvo2 is absent).
// One thread per query: the linear programs are short sequential loops over <= max_neighbors + edges half-planes;
// the half-plane lists live in the thread's private memory.

namespace orca {

constexpr float kEps = 0.00001f;                                     // RVO_EPSILON
constexpr int kMaxLines = NAVSIM_ORCA_MAX_EDGES + NAVSIM_ORCA_MAX_AGENTS;

struct V2 { float x, y; };
struct Line { V2 point, direction; };
__device__ __forceinline__ V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ V2 operator-(V2 a) { return v2(-a.x, -a.y); }
__device__ __forceinline__ V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
__device__ __forceinline__ V2 operator/(V2 a, float s) { const float inv = 1.0f / s; return v2(a.x * inv, a.y * inv); }
__device__ __forceinline__ float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float det(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float abs_sq(V2 a) { return dot(a, a); }
__device__ __forceinline__ float length(V2 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V2 normalize(V2 a) { return a / length(a); }
__device__ __forceinline__ float sqr(float a) { return a * a; }

__device__ bool lp1(const Line* lines, int line_no, float radius, V2 opt, bool direction_opt, V2& result) {
    const Line& L = lines[line_no];
    const float dp = dot(L.point, L.direction);
    const float disc = sqr(dp) + sqr(radius) - abs_sq(L.point);
    if (disc < 0.0f) return false;
    const float sd = sqrtf(disc);
    float t_left = -dp - sd, t_right = -dp + sd;
    for (int i = 0; i < line_no; ++i) {
        const float den = det(L.direction, lines[i].direction);
        const float num = det(lines[i].direction, L.point - lines[i].point);
        if (fabsf(den) <= kEps) {
            if (num < 0.0f) return false;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) t_right = t_right < t ? t_right : t;
        else             t_left = t_left > t ? t_left : t;
        if (t_left > t_right) return false;
    }
    if (direction_opt) {
        result = (dot(opt, L.direction) > 0.0f) ? L.point + t_right * L.direction : L.point + t_left * L.direction;
    } else {
        const float t = dot(L.direction, opt - L.point);
        if (t < t_left)       result = L.point + t_left * L.direction;
        else if (t > t_right) result = L.point + t_right * L.direction;
        else                  result = L.point + t * L.direction;
    }
    return true;
}

__device__ int lp2(const Line* lines, int n, float radius, V2 opt, bool direction_opt, V2& result) {
    if (direction_opt) result = radius * opt;
    else if (abs_sq(opt) > sqr(radius)) result = radius * normalize(opt);
    else result = opt;
    for (int i = 0; i < n; ++i) {
        if (det(lines[i].direction, lines[i].point - result) > 0.0f) {
            const V2 keep = result;
            if (!lp1(lines, i, radius, opt, direction_opt, result)) { result = keep; return i; }
        }
    }
    return n;
}

__device__ void lp3(const Line* lines, int n, int n_obst_lines, int begin, float radius, V2& result, Line* proj) {
    float distance = 0.0f;
    for (int i = begin; i < n; ++i) {
        if (det(lines[i].direction, lines[i].point - result) > distance) {
            int np = 0;
            for (int j = 0; j < n_obst_lines; ++j) proj[np++] = lines[j];
            for (int j = n_obst_lines; j < i; ++j) {
                Line l;
                const float d = det(lines[i].direction, lines[j].direction);
                if (fabsf(d) <= kEps) {
                    if (dot(lines[i].direction, lines[j].direction) > 0.0f) continue;
                    l.point = 0.5f * (lines[i].point + lines[j].point);
                } else {
                    l.point = lines[i].point + (det(lines[j].direction, lines[i].point - lines[j].point) / d) * lines[i].direction;
                }
                l.direction = normalize(lines[j].direction - lines[i].direction);
                proj[np++] = l;
            }
            const V2 keep = result;
            if (lp2(proj, np, radius, v2(-lines[i].direction.y, lines[i].direction.x), true, result) < np) result = keep;
            distance = det(lines[i].direction, lines[i].point - result);
        }
    }
}

// the obstacle vertices of one polygon set, addressed by a flat edge index k = polygon * n_vert + vertex
struct Obstacles {
    const double* verts; int n_vert;
    __device__ __forceinline__ V2 point(int k) const { return v2((float)verts[2 * k], (float)verts[2 * k + 1]); }
    __device__ __forceinline__ int next(int k) const { const int o = k / n_vert, i = k - o * n_vert; return o * n_vert + (i + 1) % n_vert; }
    __device__ __forceinline__ int prev(int k) const { const int o = k / n_vert, i = k - o * n_vert; return o * n_vert + (i + n_vert - 1) % n_vert; }
    __device__ __forceinline__ V2 unit_dir(int k) const { return normalize(point(next(k)) - point(k)); }
    __device__ __forceinline__ bool convex(int k) const {
        if (n_vert == 2) return true;
        const V2 pp = point(prev(k)), pt = point(k), pn = point(next(k));
        return det(pp - pn, pt - pp) >= 0.0f;                        // leftOf(prev, this, next)
    }
};

}  // namespace orca

__global__ __launch_bounds__(64) void crowd_orca_kernel(navsim_orca_params p, int n_queries, int max_agents,
                                                        const double* __restrict__ agents, const int32_t* __restrict__ n_agents,
                                                        const double* __restrict__ pref_vel, int max_obst, int n_vert,
                                                        const double* __restrict__ verts, const int32_t* __restrict__ n_obst,
                                                        const int32_t* __restrict__ obst_set, const double* __restrict__ theta,
                                                        double* __restrict__ out_vel, double* __restrict__ out_action) {
    using namespace orca;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const double* ag = agents + (size_t)q * max_agents * 6;
    int na = n_agents ? n_agents[q] : max_agents;
    na = na > max_agents ? max_agents : na;
    if (na < 1) { out_vel[2 * q] = 0.0; out_vel[2 * q + 1] = 0.0; return; }
    const V2 position = v2((float)ag[0], (float)ag[1]), velocity = v2((float)ag[2], (float)ag[3]);
    const float radius = (float)ag[4], max_speed = (float)ag[5];
    const V2 pref = v2((float)pref_vel[2 * q], (float)pref_vel[2 * q + 1]);
    const int set = obst_set ? obst_set[q] : 0;
    int no = max_obst ? (n_obst ? n_obst[set] : max_obst) : 0;
    no = no > max_obst ? max_obst : no;
    const Obstacles ob = {verts + (size_t)set * max_obst * n_vert * 2, n_vert};
    const int n_edges = no * n_vert;
    // ---- Agent::computeNeighbors
    int obn[NAVSIM_ORCA_MAX_EDGES]; float obd[NAVSIM_ORCA_MAX_EDGES]; int n_obn = 0;
    {
        const float range_sq = sqr(p.time_horizon_obst * max_speed + radius);
        for (int k = 0; k < n_edges; ++k) {
            const V2 a = ob.point(k), b = ob.point(ob.next(k));
            const float left_of = det(a - position, b - a);
            const float d_line = sqr(left_of) / abs_sq(b - a);
            if (!(d_line < range_sq) || !(left_of < 0.0f)) continue;
            const float r = dot(position - a, b - a) / abs_sq(b - a);
            float d;
            if (r < 0.0f) d = abs_sq(position - a);
            else if (r > 1.0f) d = abs_sq(position - b);
            else d = abs_sq(position - (a + r * (b - a)));
            if (d < range_sq) {
                int i = n_obn++;
                while (i != 0 && d < obd[i - 1]) { obn[i] = obn[i - 1]; obd[i] = obd[i - 1]; --i; }
                obn[i] = k; obd[i] = d;
            }
        }
    }
    int agn[NAVSIM_ORCA_MAX_AGENTS]; float agd[NAVSIM_ORCA_MAX_AGENTS]; int n_agn = 0;
    if (p.max_neighbors > 0) {
        float range_sq = sqr(p.neighbor_dist);
        const int max_n = p.max_neighbors < NAVSIM_ORCA_MAX_AGENTS ? p.max_neighbors : NAVSIM_ORCA_MAX_AGENTS;
        for (int k = 1; k < na; ++k) {
            const float d = abs_sq(position - v2((float)ag[6 * k], (float)ag[6 * k + 1]));
            if (d < range_sq) {
                if (n_agn < max_n) ++n_agn;
                int i = n_agn - 1;
                while (i != 0 && d < agd[i - 1]) { agn[i] = agn[i - 1]; agd[i] = agd[i - 1]; --i; }
                agn[i] = k; agd[i] = d;
                if (n_agn == max_n) range_sq = agd[n_agn - 1];
            }
        }
    }
    // ---- Agent::computeNewVelocity: obstacle half-planes
    Line lines[kMaxLines];
    int nl = 0;
    const float inv_tho = 1.0f / p.time_horizon_obst;
    for (int i = 0; i < n_obn; ++i) {
        int o1 = obn[i], o2 = ob.next(o1);
        const V2 p1 = ob.point(o1), p2 = ob.point(o2);
        const V2 rel1 = p1 - position, rel2 = p2 - position;
        bool covered = false;
        for (int j = 0; j < nl; ++j)
            if (det(inv_tho * rel1 - lines[j].point, lines[j].direction) - inv_tho * radius >= -kEps &&
                det(inv_tho * rel2 - lines[j].point, lines[j].direction) - inv_tho * radius >= -kEps) { covered = true; break; }
        if (covered) continue;
        const float d1 = abs_sq(rel1), d2 = abs_sq(rel2), rsq = sqr(radius);
        const V2 ovec = p2 - p1;
        const float s = dot(-rel1, ovec) / abs_sq(ovec);
        const float d_line = abs_sq(-rel1 - s * ovec);
        const bool convex1 = ob.convex(o1), convex2 = ob.convex(o2);
        const V2 dir1 = ob.unit_dir(o1);
        Line line;
        if (s < 0.0f && d1 <= rsq) {
            if (convex1) { line.point = v2(0.0f, 0.0f); line.direction = normalize(v2(-rel1.y, rel1.x)); lines[nl++] = line; }
            continue;
        } else if (s > 1.0f && d2 <= rsq) {
            if (convex2 && det(rel2, ob.unit_dir(o2)) >= 0.0f) {
                line.point = v2(0.0f, 0.0f); line.direction = normalize(v2(-rel2.y, rel2.x)); lines[nl++] = line;
            }
            continue;
        } else if (s >= 0.0f && s < 1.0f && d_line <= rsq) {
            line.point = v2(0.0f, 0.0f); line.direction = -dir1; lines[nl++] = line;
            continue;
        }
        V2 left_leg, right_leg;
        bool cv1 = convex1, cv2 = convex2;             // convexity of the (possibly merged) end vertices
        if (s < 0.0f && d_line <= rsq) {
            if (!convex1) continue;
            o2 = o1; cv2 = convex1;
            const float leg1 = sqrtf(d1 - rsq);
            left_leg = v2(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1) / d1;
            right_leg = v2(rel1.x * leg1 + rel1.y * radius, -rel1.x * radius + rel1.y * leg1) / d1;
        } else if (s > 1.0f && d_line <= rsq) {
            if (!convex2) continue;
            o1 = o2; cv1 = convex2;
            const float leg2 = sqrtf(d2 - rsq);
            left_leg = v2(rel2.x * leg2 - rel2.y * radius, rel2.x * radius + rel2.y * leg2) / d2;
            right_leg = v2(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2) / d2;
        } else {
            if (convex1) {
                const float leg1 = sqrtf(d1 - rsq);
                left_leg = v2(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1) / d1;
            } else left_leg = -dir1;
            if (convex2) {
                const float leg2 = sqrtf(d2 - rsq);
                right_leg = v2(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2) / d2;
            } else right_leg = dir1;
        }
        const V2 dir_o1 = ob.unit_dir(o1), dir_o2 = ob.unit_dir(o2), dir_left = ob.unit_dir(ob.prev(o1));
        bool left_foreign = false, right_foreign = false;
        if (cv1 && det(left_leg, -dir_left) >= 0.0f) { left_leg = -dir_left; left_foreign = true; }
        if (cv2 && det(right_leg, dir_o2) <= 0.0f) { right_leg = dir_o2; right_foreign = true; }
        const V2 left_cut = inv_tho * (ob.point(o1) - position), right_cut = inv_tho * (ob.point(o2) - position);
        const V2 cut_vec = right_cut - left_cut;
        const float t = (o1 == o2) ? 0.5f : dot(velocity - left_cut, cut_vec) / abs_sq(cut_vec);
        const float t_left = dot(velocity - left_cut, left_leg), t_right = dot(velocity - right_cut, right_leg);
        if ((t < 0.0f && t_left < 0.0f) || (o1 == o2 && t_left < 0.0f && t_right < 0.0f)) {
            const V2 w = normalize(velocity - left_cut);
            line.direction = v2(w.y, -w.x);
            line.point = left_cut + (radius * inv_tho) * w;
            lines[nl++] = line;
            continue;
        } else if (t > 1.0f && t_right < 0.0f) {
            const V2 w = normalize(velocity - right_cut);
            line.direction = v2(w.y, -w.x);
            line.point = right_cut + (radius * inv_tho) * w;
            lines[nl++] = line;
            continue;
        }
        const float inf = __builtin_inff();
        const float ds_cut = (t < 0.0f || t > 1.0f || o1 == o2) ? inf : abs_sq(velocity - (left_cut + t * cut_vec));
        const float ds_left = (t_left < 0.0f) ? inf : abs_sq(velocity - (left_cut + t_left * left_leg));
        const float ds_right = (t_right < 0.0f) ? inf : abs_sq(velocity - (right_cut + t_right * right_leg));
        if (ds_cut <= ds_left && ds_cut <= ds_right) {
            line.direction = -dir_o1;
            line.point = left_cut + (radius * inv_tho) * v2(-line.direction.y, line.direction.x);
            lines[nl++] = line;
        } else if (ds_left <= ds_right) {
            if (left_foreign) continue;
            line.direction = left_leg;
            line.point = left_cut + (radius * inv_tho) * v2(-line.direction.y, line.direction.x);
            lines[nl++] = line;
        } else {
            if (right_foreign) continue;
            line.direction = -right_leg;
            line.point = right_cut + (radius * inv_tho) * v2(-line.direction.y, line.direction.x);
            lines[nl++] = line;
        }
    }
    const int n_obst_lines = nl;
    const float inv_th = 1.0f / p.time_horizon;
    for (int i = 0; i < n_agn; ++i) {
        const double* o = ag + 6 * agn[i];
        const V2 rel_p = v2((float)o[0], (float)o[1]) - position;
        const V2 rel_v = velocity - v2((float)o[2], (float)o[3]);
        const float dist_sq = abs_sq(rel_p);
        const float comb = radius + (float)o[4], comb_sq = sqr(comb);
        Line line;
        V2 u;
        if (dist_sq > comb_sq) {
            const V2 w = rel_v - inv_th * rel_p;
            const float w_sq = abs_sq(w);
            const float dp1 = dot(w, rel_p);
            if (dp1 < 0.0f && sqr(dp1) > comb_sq * w_sq) {
                const float wl = sqrtf(w_sq);
                const V2 uw = w / wl;
                line.direction = v2(uw.y, -uw.x);
                u = (comb * inv_th - wl) * uw;
            } else {
                const float leg = sqrtf(dist_sq - comb_sq);
                if (det(rel_p, w) > 0.0f)
                    line.direction = v2(rel_p.x * leg - rel_p.y * comb, rel_p.x * comb + rel_p.y * leg) / dist_sq;
                else
                    line.direction = -(v2(rel_p.x * leg + rel_p.y * comb, -rel_p.x * comb + rel_p.y * leg) / dist_sq);
                const float dp2 = dot(rel_v, line.direction);
                u = dp2 * line.direction - rel_v;
            }
        } else {
            const float inv_dt = 1.0f / p.time_step;
            const V2 w = rel_v - inv_dt * rel_p;
            const float wl = length(w);
            const V2 uw = w / wl;
            line.direction = v2(uw.y, -uw.x);
            u = (comb * inv_dt - wl) * uw;
        }
        line.point = velocity + 0.5f * u;
        lines[nl++] = line;
    }
    V2 nv;
    const int fail = lp2(lines, nl, max_speed, pref, false, nv);
    if (fail < nl) {
        Line proj[kMaxLines];
        lp3(lines, nl, n_obst_lines, fail, max_speed, nv, proj);
    }
    out_vel[2 * q] = (double)nv.x; out_vel[2 * q + 1] = (double)nv.y;
    if (out_action) {                                                       // orca.py:128-130
        const double vx = (double)nv.x, vy = (double)nv.y;
        out_action[2 * q] = sqrt(vx * vx + vy * vy);
        out_action[2 * q + 1] = nv::atan2_(vy, vx) - (theta ? theta[q] : 0.0);
    }
}

// Agent.step with an ActionRot (crowd_sim/envs/utils/agent.py:108-141)
__global__ __launch_bounds__(256) void crowd_agent_step_kernel(double* __restrict__ pose, const double* __restrict__ action,
                                                               double* __restrict__ vel, int n, double dt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* ps = pose + 3 * (size_t)i;
    const double v = action[2 * i], r = action[2 * i + 1];
    double s, c;
    nv::sincos(ps[2] + r, s, c);
    ps[0] = ps[0] + c * v * dt;
    ps[1] = ps[1] + s * v * dt;
    if (vel) { vel[2 * i] = v * c; vel[2 * i + 1] = v * s; }
    ps[2] = nv::mod_2pi(ps[2] + r);
}
