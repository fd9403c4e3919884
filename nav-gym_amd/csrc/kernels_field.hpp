// kernels_field.hpp -- distance transform, field formats, tile table and the mirror primitives (rows a3-a6, a8/a9, a12-a14).
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).

// ============================================================================================
// a3: exact Euclidean distance transform (replaces range_libc.PyOMap + PyRayMarching.__init__,
// env.py:337-340).  Pass 1: per column, distance to the nearest occupied cell of that column
// (uint16, 0xFFFF = none).  Pass 2: per row, d2(x) = min_i (x-i)^2 + g(i)^2 by an outward search
// that stops as soon as (x-i)^2 alone exceeds the best value: exact, integer, and the search
// radius is the answer itself, so cells near obstacles (most of them) cost a handful of reads.
// ============================================================================================
constexpr int kDtInf = 32768;

// One workgroup = 64 adjacent columns x kColSeg row segments (a wavefront per segment, so a row of loads is
// 64 contiguous bytes).  Segments are scanned independently and stitched through LDS: the downward
// distance entering a segment is min over the segments above of (their last local value + rows in
// between), the upward one likewise from their first occupied row.  A single thread per column would walk
// H rows twice with one memory latency per chunk -- 190 us when only a few maps are live (navsim_regen).
constexpr int kColSeg = 8;
__global__ __launch_bounds__(64 * kColSeg) void dt_columns_kernel(const uint8_t* __restrict__ occ,
                                                                 uint16_t* __restrict__ g, int H, int W,
                                                                 const int* __restrict__ n_live,
                                                                 const int* __restrict__ kind = nullptr) {
    __shared__ int down_last[kColSeg][64], up_first[kColSeg][64];
    const int cx = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + cx;
    const size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;          // navsim_regen: only the first *n_live maps are live
    if (kind && kind[m] == 0) return;                 // ... and outdoor maps got their field from the geometry
    const int rows = (H + kColSeg - 1) / kColSeg;
    const int y0 = seg * rows < H ? seg * rows : H, y1 = (y0 + rows < H) ? y0 + rows : H;
    const bool live = x < W;
    const uint8_t* o = occ + m * (size_t)H * W;
    uint16_t* gg = g + m * (size_t)H * W;
    constexpr int CH = 16;                            // loads of a chunk are independent and issued together
    int d = kDtInf, first = kDtInf;
    if (live)
        for (int ya = y0; ya < y1; ya += CH) {
            uint8_t v[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = (ya + j < y1) ? o[(size_t)(ya + j) * W + x] : 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (ya + j < y1) {
                    if (v[j] && first == kDtInf) first = ya + j - y0;
                    d = v[j] ? 0 : (d >= kDtInf ? kDtInf : d + 1);
                    gg[(size_t)(ya + j) * W + x] = (uint16_t)(d >= kDtInf ? 0xFFFF : d);
                }
            }
        }
    down_last[seg][cx] = d;
    up_first[seg][cx] = first;
    __syncthreads();
    if (!live) return;
    int cd = kDtInf, cu = kDtInf;                     // distance at the row just above / just below the segment
    for (int s2 = 0; s2 < seg; ++s2) {
        int r0 = s2 * rows < H ? s2 * rows : H, r1 = (r0 + rows < H) ? r0 + rows : H;
        int through = cd >= kDtInf ? kDtInf : cd + (r1 - r0);
        cd = down_last[s2][cx] < through ? down_last[s2][cx] : through;
    }
    for (int s2 = kColSeg - 1; s2 > seg; --s2) {
        int r0 = s2 * rows < H ? s2 * rows : H, r1 = (r0 + rows < H) ? r0 + rows : H;
        int through = cu >= kDtInf ? kDtInf : cu + (r1 - r0);
        cu = up_first[s2][cx] < through ? up_first[s2][cx] : through;
    }
    int u = cu;
    for (int yb = y1 - 1; yb >= y0; yb -= CH) {
        uint16_t v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = (yb - j >= y0) ? gg[(size_t)(yb - j) * W + x] : 0;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int y = yb - j;
            if (y >= y0) {
                const int cur = (v[j] == 0xFFFF) ? kDtInf : v[j];
                u = (cur == 0) ? 0 : (u >= kDtInf ? kDtInf : u + 1);
                const int from_above = cd >= kDtInf ? kDtInf : cd + (y - y0 + 1);
                int best = cur < from_above ? cur : from_above;
                best = best < u ? best : u;
                if (best != cur) gg[(size_t)y * W + x] = (uint16_t)(best >= kDtInf ? 0xFFFF : best);
            }
        }
    }
}

// The march step of a free sample at distance d (cells): t += max(0.999 * d, 1), RayMarching::calc_range of
// range_libc (env.py:425).  How the product is rounded is NOT pinned by anything in /root/reference (the
// package's source is absent; include/navsim.h NAVSIM_MARCH_* states the candidates): NAVSIM_MARCH_F64 keeps the
// coefficient a double, fl32(fl64(d) * 0.999); NAVSIM_MARCH_F32 keeps it a float member, d * 0.999f (the default);
// NAVSIM_MARCH_F32_FMA is the float member with the sample position contracted into an FMA (march_pos).
//
// kMarchF64Exact32 is the SAME function as NAVSIM_MARCH_F64 evaluated without float64 instructions (v_cvt_f64_f32,
// v_mul_f64, v_cvt_f32_f64 cost the probe loop 5 % of the c2 step): 0.999 = c_hi + c_lo in float32, the product's
// rounding error recovered exactly by an FMA, one final addition.  Identical to the float64 form for every
// d = sqrtf(n), n an integer below 2^22 (exhaustive device check in tests/test_gpu_parity.py), which is every
// distance the packed field or a rect record can produce on maps up to 1448 cells per side; the host selects it
// only there (march_rule_variant).  Not identical for arbitrary floats, so the float32 field (caller-supplied
// values) and larger maps keep the float64 instructions.
constexpr int kMarchF64Exact32 = 3;
template <int RULE>
__device__ __forceinline__ float march_step(float d) {
    float stp;
    if (RULE == kMarchF64Exact32) {
        const float c_hi = 0.999f, c_lo = (float)(0.999 - (double)0.999f);
        const float hi = d * c_hi;
        stp = hi + __builtin_fmaf(d, c_lo, __builtin_fmaf(d, c_hi, -hi));
    } else {
        stp = (RULE == NAVSIM_MARCH_F32 || RULE == NAVSIM_MARCH_F32_FMA) ? d * 0.999f : (float)((double)d * 0.999);
    }
    return (stp > 1.0f) ? stp : 1.0f;
}
// the sample position x0 + dx * t of calc_range: two roundings (v_mul_f32 + v_add_f32; the translation unit is
// -ffp-contract=off) or, under NAVSIM_MARCH_F32_FMA, the single rounding of the FMA GCC contracts it into
template <int RULE>
__device__ __forceinline__ float march_pos(float x0, float dx, float t) {
    if (RULE == NAVSIM_MARCH_F32_FMA) return __builtin_fmaf(dx, t, x0);
    return x0 + dx * t;
}

// distance-field accessors -------------------------------------------------------------------
// FieldF32: float32 row-major (what range_libc keeps).  FieldU16T: uint16 squared distances in
// 8x8-cell tiles, one tile = one 128-B line = one HBM fill (profiles/gather_granularity.py): a
// fan of adjacent beams touches ~2.2x fewer lines than with float32 rows, and sqrtf(d2) is the
// very float the float32 field holds.
// Which slot of the per-map arrays (field, overflow plane, rect records, rect index rows, costmap) holds arena e's map:
// navsim_state.map_slot (NULL: the arena's own index).  navsim_step_install exchanges the live and the staged state's entries
// instead of copying a map.
__device__ __forceinline__ int map_slot_of(const navsim_config& c, const navsim_state& st, int e) {
    return c.shared_field ? 0 : (st.map_slot ? st.map_slot[e] : e);
}

struct FieldF32 {
    const float* p; int W;
    __device__ __forceinline__ FieldF32(const void* base, const float*, int e, int H, int W_)
        : p((const float*)base + (size_t)e * H * W_), W(W_) {}
    typedef float raw_t;
    // byte offsets stay below 4 GiB per arena, so a 32-bit lane offset on a uniform base suffices
    __device__ __forceinline__ raw_t load(int px, int py) const {
        unsigned off = ((unsigned)py * (unsigned)W + (unsigned)px) * 4u;
        return *(const float*)((const char*)p + off);
    }
    __device__ __forceinline__ bool occupied(raw_t v) const { return v <= 0.0f; }
    __device__ __forceinline__ float decode(raw_t v, int, int) const { return v; }
    __device__ __forceinline__ float decode_nz(raw_t v, int, int) const { return v; }
    __device__ __forceinline__ float at(int px, int py) const { return load(px, py); }
    // the float this field holds for an integer squared distance (dt_rows_kernel: sqrtf((float)d2))
    __device__ __forceinline__ static float sqrt_d2(int d2) { return sqrtf((float)d2); }
};
// OVF = false: the caller guarantees that no cell is saturated (navsim_build_field reported none and gave no
// overflow plane), so decoding needs no test for the 0xFFFF escape -- one divergent branch less per probe
template <bool OVF>
struct FieldU16TT {
    const uint16_t* p; const float* ovf; int W, tpr;
    __device__ __forceinline__ FieldU16TT(const void* base, const float* overflow, int e, int H, int W_)
        : W(W_), tpr((W_ + 7) >> 3) {
        size_t per_map = (size_t)((H + 7) >> 3) * tpr * 64;
        p = (const uint16_t*)base + (size_t)e * per_map;
        ovf = overflow ? overflow + (size_t)e * H * W_ : nullptr;
    }
    __device__ __forceinline__ static size_t index(int px, int py, int tpr) {
        return ((size_t)((py >> 3) * tpr + (px >> 3)) << 6) + ((py & 7) << 3) + (px & 7);
    }
    typedef unsigned raw_t;
    // load and decode are split so that a thread can issue the loads of all its rays back to back
    // before the (rare, divergent) overflow read of any of them
    __device__ __forceinline__ raw_t load(int px, int py) const {
        unsigned upx = (unsigned)px, upy = (unsigned)py;
        unsigned off = ((((upy >> 3) * (unsigned)tpr + (upx >> 3)) << 6) | ((upy & 7u) << 3) | (upx & 7u)) * 2u;
        return *(const uint16_t*)((const char*)p + off);
    }
    __device__ __forceinline__ bool occupied(raw_t v) const { return v == 0u; }
    __device__ __forceinline__ float decode(raw_t v, int px, int py) const {
        if (OVF && v == 0xFFFFu) return ovf[(size_t)py * W + px];      // d2 >= 65535: exact float plane
        return nv::sqrt_small_int((float)v);
    }
    // the march's form: the value of an OCCUPIED sample (v = 0) is never used there, so the 0 guard is dropped
    __device__ __forceinline__ float decode_nz(raw_t v, int px, int py) const {
        if (OVF && v == 0xFFFFu) return ovf[(size_t)py * W + px];
        return nv::sqrt_small_int_nz((float)v);
    }
    __device__ __forceinline__ float at(int px, int py) const { return decode(load(px, py), px, py); }
    // d of an integer squared distance from a rect record, as decode() would give it (the exact short sqrt holds
    // for every integer below 2^22, i.e. maps up to 1024 cells per side); NaN at d2 = 0, which the march never uses
    __device__ __forceinline__ static float sqrt_d2(int d2) { return nv::sqrt_small_int_nz((float)d2); }
};
typedef FieldU16TT<true> FieldU16T;
typedef FieldU16TT<false> FieldU16TN;
// How far a ray has to be marched.  The reference marches up to H*W cells (env.py:337) and clips the
// result to range_max afterwards (env.py:434).  A hit found at parameter t lies at least t - sqrt(2) cells
// from the origin, so once t exceeds range_max / resolution + 4 every possible outcome -- a later hit,
// leaving the map, or the H*W limit -- clips to range_max: stopping there returns the same scan.
__device__ __forceinline__ float march_limit(int H, int W, double range_max, double resolution) {
    const float full = (float)((long long)H * W);
    const float lim = (float)(floor(range_max / resolution) + 4.0);
    return lim < full ? lim : full;
}

// FORMAT 0: float32 row-major to `field`; 1: uint16 tiles to `field` (+ float32 to `overflow` if
// given, + saturation count)
template <int FORMAT>
__global__ __launch_bounds__(256) void dt_rows_kernel(const uint16_t* __restrict__ g,
                                                      void* __restrict__ field_v, float* __restrict__ overflow,
                                                      int32_t* __restrict__ n_saturated, int H, int W,
                                                      const int* __restrict__ n_live,
                                                      const int* __restrict__ kind = nullptr) {
    extern __shared__ int32_t row[];                 // W entries of g(i)^2-ready distances
    size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;
    if (kind && kind[m] == 0) return;
    int y = blockIdx.x;
    const uint16_t* gr = g + (m * (size_t)H + y) * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int v = gr[x];
        row[x] = (v == 0xFFFF) ? kDtInf : v;
    }
    __syncthreads();
    const int tpr = (W + 7) >> 3;
    const size_t per_map_t = (size_t)((H + 7) >> 3) * tpr * 64;
    int sat = 0;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int g0 = row[x];
        int best = g0 * g0;
        // eight distances per round: the 16 LDS reads do not depend on `best`, only the exit test does, and
        // candidates past the exit point (dx^2 >= best) can never win, so running a round to its end is
        // result-neutral.  (One distance per round pays an LDS latency per step: 25 us for an open row.)
        constexpr int UR = 8;
        for (int dx0 = 1; dx0 < W; dx0 += UR) {
            if (dx0 * dx0 >= best) break;
            int vl[UR], vr[UR];
#pragma unroll
            for (int j = 0; j < UR; ++j) {
                int xl = x - (dx0 + j), xr = x + (dx0 + j);
                vl[j] = (xl >= 0) ? row[xl] : kDtInf;
                vr[j] = (xr < W) ? row[xr] : kDtInf;
            }
#pragma unroll
            for (int j = 0; j < UR; ++j) {
                int dx2 = (dx0 + j) * (dx0 + j);
                int cl = dx2 + vl[j] * vl[j], cr = dx2 + vr[j] * vr[j];
                best = cl < best ? cl : best;
                best = cr < best ? cr : best;
            }
        }
        if (FORMAT == 0) {
            ((float*)field_v)[(m * (size_t)H + y) * W + x] = sqrtf((float)best);
        } else if (FORMAT == 1) {
            uint16_t* out = (uint16_t*)field_v + m * per_map_t;
            out[FieldU16T::index(x, y, tpr)] = (uint16_t)(best >= 65535 ? 0xFFFF : best);
            if (overflow) overflow[(m * (size_t)H + y) * W + x] = sqrtf((float)best);
            sat += best >= 65535;
        }
    }
    if (FORMAT != 0 && n_saturated && sat) atomicAdd(n_saturated, sat);
}

// ============================================================================================
// a4: PyRayMarching.calc_range_many (env.py:425): one thread per query
// ============================================================================================
__global__ __launch_bounds__(256) void cast_static_kernel(const float* __restrict__ field, int H, int W,
                                                          const float* __restrict__ q, int n_per_env,
                                                          long long n_total, float max_range, int march_rule,
                                                          float* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    long long e = i / n_per_env;
    const float* f = field + (size_t)e * H * W;
    float dx, dy;
    nv::beam_dir(q[3 * i + 2], dx, dy);
    out[i] = nv::trace_ray(f, H, W, q[3 * i], q[3 * i + 1], dx, dy, max_range, march_rule);
}

// ============================================================================================
// a5: CMap2D.render_contours_in_lidar (env.py:431): one thread per (env, beam)
// ============================================================================================
__global__ __launch_bounds__(256) void render_polys_kernel(float* __restrict__ ranges,
                                                           const double* __restrict__ angles, int B,
                                                           const float* __restrict__ verts,
                                                           const int32_t* __restrict__ n_verts, int V,
                                                           const float* __restrict__ origin) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int e = blockIdx.y;
    if (k >= B) return;
    const float* vv = verts + (size_t)e * V * 3;
    int nv_ = n_verts[e];
    float ox = origin[2 * e], oy = origin[2 * e + 1];
    float c, s;
    nv::beam_dir((float)angles[(size_t)e * B + k], c, s);
    float r = ranges[(size_t)e * B + k];
    int start = 0;
    while (start < nv_) {
        int end = start;
        while (end + 1 < nv_ && vv[3 * (end + 1)] == vv[3 * start]) ++end;
        for (int v = start; v <= end; ++v) {
            int w = (v == end) ? start : v + 1;      // polygons are closed automatically
            nv::seg_merge(r, ox, oy, c, s, vv[3 * v + 1], vv[3 * v + 2], vv[3 * w + 1], vv[3 * w + 2]);
        }
        start = end + 1;
    }
    ranges[(size_t)e * B + k] = r;
}

// ============================================================================================
// a6: CMap2D.render_agents_in_lidar (env.py:432): one thread per (env, beam)
// ============================================================================================
__global__ __launch_bounds__(256) void render_legs_kernel(float* __restrict__ ranges,
                                                          const double* __restrict__ angles, int B,
                                                          const float* __restrict__ agents,
                                                          const int32_t* __restrict__ n_agents, int A,
                                                          const float* __restrict__ origin) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int e = blockIdx.y;
    if (k >= B) return;
    float ox = origin[2 * e], oy = origin[2 * e + 1];
    float c, s;
    nv::beam_dir((float)angles[(size_t)e * B + k], c, s);
    float r = ranges[(size_t)e * B + k];
    int na = n_agents[e];
    for (int i = 0; i < na; ++i) {
        const float* a = agents + ((size_t)e * A + i) * 8;
        float cc[4];
        nv::leg_centres(a[0], a[1], a[2], a[3], a[4], a[5], cc);
        nv::circle_merge(r, ox, oy, c, s, cc[0], cc[1], nv::kLegRadius);
        nv::circle_merge(r, ox, oy, c, s, cc[2], cc[3], nv::kLegRadius);
    }
    ranges[(size_t)e * B + k] = r;
}

// ============================================================================================
// a8 / a9: set_vel
// ============================================================================================
__global__ __launch_bounds__(256) void integrate_kernel(double* __restrict__ pose,
                                                        const double* __restrict__ cmd,
                                                        double* __restrict__ vel_out, int n, double dt,
                                                        double off) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3] = {pose[3 * i], pose[3 * i + 1], pose[3 * i + 2]};
    double v[2];
    nv::set_vel(p, cmd[2 * i], cmd[2 * i + 1], dt, off, v);
    pose[3 * i] = p[0]; pose[3 * i + 1] = p[1]; pose[3 * i + 2] = p[2];
    if (vel_out) { vel_out[2 * i] = v[0]; vel_out[2 * i + 1] = v[1]; }
}

// ============================================================================================
// block-level helpers
// ============================================================================================
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = (o < v) ? o : v;
    }
    return v;
}

// ============================================================================================
// a12 / a13: compute_rewards / compute_terminals / compute_info on arbitrary obs rows (HER API)
// one workgroup per row
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void reward_done_kernel(navsim_config c, const T* __restrict__ obs,
                                                          const T* __restrict__ goals,
                                                          const float* __restrict__ thr,
                                                          const float* __restrict__ dthr,
                                                          double* reward, uint8_t* done, float* is_success,
                                                          float* is_crash, double* distance) {
    __shared__ double s_ratio[kMaxWaves];
    const int B = c.n_beams, S = c.n_scan_stack, D = S * B + 7;
    const int row = blockIdx.x;
    const T* o = obs + (size_t)row * D;
    const T* scan = o + (size_t)(S - 1) * B;
    int crash = 0, disc = 0;
    double rmin = 1.0e300;
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        double s = (double)scan[k];
        if (s - (double)thr[k] < 0.0) crash = 1;
        if (s - (double)dthr[k] < 0.0) disc = 1;
        double ratio = nv::discomfort_ratio(s, thr[k], dthr[k]);
        rmin = ratio < rmin ? ratio : rmin;
    }
    crash = __syncthreads_or(crash);
    disc = __syncthreads_or(disc);
    rmin = wave_min_f64(rmin);
    if ((threadIdx.x & 63) == 0) s_ratio[threadIdx.x >> 6] = rmin;
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) rmin = s_ratio[w] < rmin ? s_ratio[w] : rmin;
        const T* tail = o + (size_t)S * B;
        double prev_xy[2] = {(double)tail[0], (double)tail[1]};
        double pose[2] = {(double)tail[2], (double)tail[3]};
        double vel[2] = {(double)tail[4], (double)tail[5]};
        double goal[2] = {(double)goals[2 * row], (double)goals[2 * row + 1]};
        nv::RewardOut r = nv::reward_scalar(c, prev_xy, pose, vel, goal, crash != 0, disc != 0, rmin);
        if (reward) reward[row] = r.reward;
        if (done) done[row] = (uint8_t)r.done;
        if (is_success) is_success[row] = r.success;
        if (is_crash) is_crash[row] = r.crash;
        if (distance) distance[row] = r.distance;
    }
}

// ============================================================================================
// a14: _make_scan_threshold (env.py:162-180)
// ============================================================================================
__global__ __launch_bounds__(256) void scan_threshold_kernel(navsim_config c, const float* __restrict__ fp,
                                                             int nvert, float* __restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.n_beams) return;
    double step = nv::linspace_step(c);
    double ang = nv::linspace_k(c, k, step) + (double)0.0f;
    float dx, dy;
    nv::beam_dir((float)ang, dx, dy);
    float rmax = (float)c.range_max;
    float r = rmax;
    for (int v = 0; v < nvert; ++v) {
        int w = (v + 1 == nvert) ? 0 : v + 1;
        nv::seg_merge(r, 0.0f, 0.0f, dx, dy, fp[2 * v], fp[2 * v + 1], fp[2 * w], fp[2 * w + 1]);
    }
    r = r < 0.0f ? 0.0f : r;
    r = r > rmax ? rmax : r;
    out[k] = r;
}
