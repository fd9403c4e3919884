// kernels_crowd_maps.hpp -- CrowdSim-v0 local maps (nav_gym/src/crowd_sim/envs/crowd_sim.py:999-1186).
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace).
// Specification: oracle/navsim_ref.c navsim_crowd_angular_map_cpu / navsim_crowd_local_map_cpu.

// ---- get_local_map_angular: one wavefront per env.  The reference runs 2 * n_obst * 4 short SEQUENCES of
// calculate_angular_map_distances calls, each with its own (rad_indeces, locations) history: per obstacle and
// robot-corner "edge" the obstacle's vertices in order (crowd_sim.py:1077-1083), then per obstacle and vertex the
// four edges in order (crowd_sim.py:1085-1091).  A sequence is sequential, sequences are independent, and the only
// shared state is radial_dist_vector, updated by min: one lane per sequence, a 64-bit LDS atomicMin on the bits of
// the (non-negative) float64 distances -- the minimum does not depend on the order, so the result is the
// reference's.
struct AmapSeen { int idx; double x, y; };

__device__ __forceinline__ void amap_min(unsigned long long* rdv, int j, double v) {
    atomicMin(&rdv[j], (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ void amap_calc(const navsim_crowd_map_params& p, double vx, double vy, double ex, double ey,
                                          double ct, double st, unsigned long long* rdv, AmapSeen* seen, int& n_seen) {
    const int dim = p.angular_dim;
    const double res = (p.angular_max - p.angular_min) / (double)dim;
    double px = (vx - ex) * ct + (vy - ey) * st;
    double py = (vy - ey) * ct - (vx - ex) * st;
    const double phi = nv::atan2_(py, px);
    const int rad_idx = (int)((phi - p.angular_min) / res);
    const double distance = sqrt(px * px + py * py);
    if (rad_idx >= 0 && rad_idx < dim) amap_min(rdv, rad_idx, distance);
    for (int s = 0; s < n_seen; ++s) {
        const int old = seen[s].idx;
        const double lx = seen[s].x, ly = seen[s].y;
        const int ad = rad_idx > old ? rad_idx - old : old - rad_idx;
        bool wrapped;
        int idx_diff;
        if ((double)ad > nv::kPi / res) {
            wrapped = true;
            idx_diff = (rad_idx > old) ? dim - rad_idx + old : dim - old + rad_idx;
        } else {
            wrapped = false;
            idx_diff = ad;
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
    seen[n_seen].idx = rad_idx; seen[n_seen].x = vx; seen[n_seen].y = vy;
    ++n_seen;
}

__global__ __launch_bounds__(64) void crowd_angular_map_kernel(navsim_crowd_map_params p, int max_obst, int n_vert,
                                                               const double* __restrict__ robot,
                                                               const double* __restrict__ verts,
                                                               const int32_t* __restrict__ n_obst,
                                                               double* __restrict__ out) {
    extern __shared__ unsigned long long amap_rdv[];                      // [angular_dim] float64 bits
    const int e = blockIdx.x, lane = threadIdx.x, dim = p.angular_dim;
    const double* r = robot + (size_t)e * 4;
    for (int k = lane; k < dim; k += 64) amap_rdv[k] = (unsigned long long)__double_as_longlong(p.angular_max_range);
    __syncthreads();
    double st, ct;
    nv::sincos(r[2], st, ct);
    int no = n_obst ? n_obst[e] : max_obst;
    no = no > max_obst ? max_obst : no;
    const int seq_a = no * 4, seq_b = no * n_vert;
    for (int q = lane; q < seq_a + seq_b; q += 64) {
        AmapSeen seen[NAVSIM_CROWD_MAX_VERTS > 4 ? NAVSIM_CROWD_MAX_VERTS : 4];
        int ns = 0;
        if (q < seq_a) {                                                  // obstacle o, edge k: vertices in order
            const int o = q >> 2, k = q & 3;
            const double ex = r[0] + ((k & 1) ? 1.0 : -1.0) * r[3], ey = r[1] + ((k & 2) ? 1.0 : -1.0) * r[3];
            const double* vv = verts + ((size_t)e * max_obst + o) * n_vert * 2;
            for (int v = 0; v < n_vert; ++v) amap_calc(p, vv[2 * v], vv[2 * v + 1], ex, ey, ct, st, amap_rdv, seen, ns);
        } else {                                                          // obstacle o, vertex v: edges in order
            const int qq = q - seq_a, o = qq / n_vert, v = qq - o * n_vert;
            const double* vv = verts + ((size_t)e * max_obst + o) * n_vert * 2;
            for (int k = 0; k < 4; ++k) {
                const double ex = r[0] + ((k & 1) ? 1.0 : -1.0) * r[3], ey = r[1] + ((k & 2) ? 1.0 : -1.0) * r[3];
                amap_calc(p, vv[2 * v], vv[2 * v + 1], ex, ey, ct, st, amap_rdv, seen, ns);
            }
        }
    }
    __syncthreads();
    for (int k = lane; k < dim; k += 64) {
        double v = __longlong_as_double((long long)amap_rdv[k]);
        out[(size_t)e * dim + k] = p.normalize ? v / p.angular_max_range : v;
    }
}

// ---- get_local_map + rotate_grid_around_center: one workgroup per env, the window in LDS (1 byte per cell: the
// grid holds 0 / 1 only), then one thread per destination cell (bilinear blend of exact 0 / 1 values, see the oracle).
__device__ __forceinline__ int py_round_int(double v) { return (int)rint(v); }      // round-half-even, like Python 3

__global__ __launch_bounds__(256) void crowd_local_map_kernel(navsim_crowd_map_params p, int grid, int S,
                                                              const uint8_t* __restrict__ free_map,
                                                              const double* __restrict__ robot, int rotate,
                                                              uint8_t* __restrict__ out) {
    extern __shared__ uint8_t lm_grid[];                                  // [S][S]
    const int e = blockIdx.x, tid = threadIdx.x;
    const double* r = robot + (size_t)e * 4;
    const uint8_t* m = free_map + (size_t)e * grid * grid;
    uint8_t* o = out + (size_t)e * S * S;
    const int cx = py_round_int((r[0] + p.map_size_m / 2.0) / p.map_resolution);
    const int cy = py_round_int((r[1] + p.map_size_m / 2.0) / p.map_resolution);
    int sx = py_round_int((double)cx - floor((double)S / 2.0)), sy = py_round_int((double)cy - floor((double)S / 2.0));
    int ex = sx + S - 1, ey = sy + S - 1;
    const int mx = grid - 1, my = grid - 1;
    int gsx = 0, gsy = 0, gex = S - 1, gey = S - 1;
    if (sx < 0) { gsx = -sx; sx = 0; } else if (ex > mx) { gex = gex - (ex - mx); ex = mx; }
    if (sy < 0) { gsy = -sy; sy = 0; } else if (ey > my) { gey = gey - (ey - my); ey = my; }
    const bool all_ones = gsy > gey || sy > ey || sx > ex || gsx > gex;   // crowd_sim.py:1152-1154
    const int na = (gex - gsx < ex - sx) ? gex - gsx : ex - sx, nb = (gey - gsy < ey - sy) ? gey - gsy : ey - sy;
    for (int k = tid; k < S * S; k += 256) {
        const int a = k / S - gsx, b = k % S - gsy;                       // exclusive slice ends: na x nb cells copied
        uint8_t v = 1;
        if (!all_ones && a >= 0 && a < na && b >= 0 && b < nb) v = m[(size_t)(sx + a) * grid + (sy + b)] != 0;
        lm_grid[k] = v;
    }
    __syncthreads();
    if (all_ones || !rotate) {
        for (int k = tid; k < S * S; k += 256) o[k] = lm_grid[k];
        return;
    }
    const double angle = (-r[2] + nv::kPi / 2.0) * 180.0 / nv::kPi;
    double sa, ca;
    nv::sincos(angle * nv::kPi / 180.0, sa, ca);
    const double cxr = (double)S / 2.0, cyr = (double)S / 2.0;
    const double M0 = ca, M1 = sa, M2 = (1.0 - ca) * cxr - sa * cyr, M3 = -sa, M4 = ca, M5 = sa * cxr + (1.0 - ca) * cyr;
    double D = M0 * M4 - M1 * M3;
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double i0 = M4 * D, i4 = M0 * D, i1 = M1 * (-D), i3 = M3 * (-D);
    const double b1 = -i0 * M2 - i1 * M5, b2 = -i3 * M2 - i4 * M5;
    for (int k = tid; k < S * S; k += 256) {
        const int y = k / S, x = k - y * S;
        const int X0 = (int)llrint((i1 * y + b1) * 1024.0) + 16, Y0 = (int)llrint((i4 * y + b2) * 1024.0) + 16;
        const int X = ((int)llrint(i0 * x * 1024.0) + X0) >> 5, Y = ((int)llrint(i3 * x * 1024.0) + Y0) >> 5;
        const int ix = X >> 5, iy = Y >> 5;
        const double fx = (double)(X & 31) / 32.0, fy = (double)(Y & 31) / 32.0;
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xx = ix + (t & 1), yy = iy + (t >> 1);
            v[t] = (xx >= 0 && xx < S && yy >= 0 && yy < S) ? (double)lm_grid[yy * S + xx] : 1.0;
        }
        const double val = v[0] * ((1.0 - fx) * (1.0 - fy)) + v[1] * (fx * (1.0 - fy)) + v[2] * ((1.0 - fx) * fy) +
                           v[3] * (fx * fy);
        o[k] = val > 0.9;
    }
}
