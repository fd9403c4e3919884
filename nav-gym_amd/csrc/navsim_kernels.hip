// navsim_kernels.hip -- hand-written gfx950 kernels + the C ABI of include/navsim.h.
//
// Hot path: NavGymEnv.step (nav_gym/src/nav_gym_env/env.py:591-728 of leekwoon/nav-gym), batched
// over E independent arenas.  One workgroup owns one arena for the whole step: pedestrian update,
// robot integration, lidar ray-march over the arena's distance field, reward / done / info,
// crash revert (or respawn) with re-scan and observation packing all happen in ONE launch, so the
// only HBM traffic is the distance-field sectors the rays touch, the arena's small state and the
// observation row written once.  No MFMA: there is no dense contraction on this path.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see nav-gym_amd/csrc/build.sh).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/navsim.h"
#include "navmath.hpp"
#include "navsim_device.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxWaves = 16;

// Diagnostic build only (-DNAVSIM_STAMPS, profiles/stamp_phases.py): s_memtime at the phase
// boundaries of each arena's workgroup, written to a buffer nothing else reads.  The shipped library
// is built without it (no stamp executes in the measured kernel).
#ifdef NAVSIM_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#ifdef NAVSIM_STAMPS_REALTIME      // chip-wide 100 MHz clock (comparable across XCDs) instead of the per-XCD shader clock
#define NAVSIM_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NAVSIM_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define NAVSIM_STAMP(i) do { } while (0)
#endif

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
                                                                 const int* __restrict__ n_live) {
    __shared__ int down_last[kColSeg][64], up_first[kColSeg][64];
    const int cx = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + cx;
    const size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;          // navsim_regen: only the first *n_live maps are live
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

// distance-field accessors -------------------------------------------------------------------
// FieldF32: float32 row-major (what range_libc keeps).  FieldU16T: uint16 squared distances in
// 8x8-cell tiles, one tile = one 128-B line = one HBM fill (profiles/gather_granularity.py): a
// fan of adjacent beams touches ~2.2x fewer lines than with float32 rows, and sqrtf(d2) is the
// very float the float32 field holds.
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
    __device__ __forceinline__ float at(int px, int py) const { return load(px, py); }
    // march step of a non-occupied sample: t += max(fl32(fl64(d) * 0.999), 1)
    __device__ __forceinline__ float step_of(raw_t v, int, int) const {
        float stp = (float)((double)v * 0.999);
        return (stp > 1.0f) ? stp : 1.0f;
    }
};
struct FieldU16T {
    const uint16_t* p; const float* ovf; int W, tpr;
    __device__ __forceinline__ FieldU16T(const void* base, const float* overflow, int e, int H, int W_)
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
        if (v == 0xFFFFu) return ovf[(size_t)py * W + px];      // d2 >= 65535: exact float plane
        return nv::sqrt_small_int((float)v);
    }
    __device__ __forceinline__ float at(int px, int py) const { return decode(load(px, py), px, py); }
    __device__ __forceinline__ float step_of(raw_t v, int px, int py) const {
        float stp = (float)((double)decode(v, px, py) * 0.999);
        return (stp > 1.0f) ? stp : 1.0f;
    }
};
// float32 march steps in 8x4-cell tiles (one tile = one 128-B line): the loop adds the loaded value
struct FieldF32S {
    const float* p; const float* ovf; int W, tpr;
    __device__ __forceinline__ FieldF32S(const void* base, const float* overflow, int e, int H, int W_)
        : W(W_), tpr((W_ + 7) >> 3) {
        size_t per_map = (size_t)((H + 3) >> 2) * tpr * 32;
        p = (const float*)base + (size_t)e * per_map;
        ovf = overflow ? overflow + (size_t)e * H * W_ : nullptr;
    }
    __device__ __forceinline__ static size_t index(int px, int py, int tpr) {
        return ((size_t)((py >> 2) * tpr + (px >> 3)) << 5) + ((py & 3) << 3) + (px & 7);
    }
    typedef float raw_t;
    __device__ __forceinline__ raw_t load(int px, int py) const {
        unsigned upx = (unsigned)px, upy = (unsigned)py;
        unsigned off = ((((upy >> 2) * (unsigned)tpr + (upx >> 3)) << 5) | ((upy & 3u) << 3) | (upx & 7u)) * 4u;
        return *(const float*)((const char*)p + off);
    }
    __device__ __forceinline__ bool occupied(raw_t v) const { return v == 0.0f; }
    __device__ __forceinline__ float step_of(raw_t v, int, int) const { return v; }
    // exact distance (first probe, social force): from the float32 plane
    __device__ __forceinline__ float decode(raw_t, int px, int py) const { return ovf[(size_t)py * W + px]; }
    __device__ __forceinline__ float at(int px, int py) const { return ovf[(size_t)py * W + px]; }
};

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
                                                      const int* __restrict__ n_live) {
    extern __shared__ int32_t row[];                 // W entries of g(i)^2-ready distances
    size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;
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
        } else {
            const size_t per_map_s = (size_t)((H + 3) >> 2) * tpr * 32;
            float d = sqrtf((float)best);
            float stp = (float)((double)d * 0.999);
            stp = (stp > 1.0f) ? stp : 1.0f;
            ((float*)field_v)[m * per_map_s + FieldF32S::index(x, y, tpr)] = (best == 0) ? 0.0f : stp;
            if (overflow) overflow[(m * (size_t)H + y) * W + x] = d;
            sat += 1;                                  // "saturated": the overflow plane is always needed
        }
    }
    if (FORMAT != 0 && n_saturated && sat) atomicAdd(n_saturated, sat);
}

// ============================================================================================
// analytic tile records (navsim_build_tiles): feature transform + per-tile verification
// ============================================================================================
constexpr unsigned kTileValid = 1u << 31, kTileDx0 = 1u << 30, kTileDy0 = 1u << 29;

// nearest occupied row per cell of a column (-1: none); ties go to the row above
__global__ __launch_bounds__(256) void ft_columns_kernel(const uint8_t* __restrict__ occ,
                                                         int16_t* __restrict__ nr, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    size_t m = blockIdx.y;
    if (x >= W) return;
    const uint8_t* o = occ + m * (size_t)H * W;
    int16_t* r = nr + m * (size_t)H * W;
    int last = -1;
    for (int y = 0; y < H; ++y) {
        if (o[(size_t)y * W + x]) last = y;
        r[(size_t)y * W + x] = (int16_t)last;
    }
    last = -1;
    for (int y = H - 1; y >= 0; --y) {
        if (o[(size_t)y * W + x]) last = y;
        int up = r[(size_t)y * W + x];
        if (last >= 0 && (up < 0 || last - y < y - up)) r[(size_t)y * W + x] = (int16_t)last;
    }
}

// per row: exact d2 and the obstacle cell (ox, oy) that realises it
__global__ __launch_bounds__(256) void ft_rows_kernel(const int16_t* __restrict__ nr, int32_t* __restrict__ d2out,
                                                      int16_t* __restrict__ oxy, int H, int W) {
    extern __shared__ int32_t row[];                 // vertical distance g(i) of this row
    size_t m = blockIdx.y;
    int y = blockIdx.x;
    const int16_t* r = nr + (m * (size_t)H + y) * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int v = r[x];
        row[x] = (v < 0) ? kDtInf : (v > y ? v - y : y - v);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int g0 = row[x];
        int best = g0 * g0, arg = x;
        for (int dx = 1; dx < W; ++dx) {
            int dx2 = dx * dx;
            if (dx2 >= best) break;
            int xl = x - dx, xr = x + dx;
            if (xl >= 0) { int v = row[xl]; int c = dx2 + v * v; if (c < best) { best = c; arg = xl; } }
            if (xr < W)  { int v = row[xr]; int c = dx2 + v * v; if (c < best) { best = c; arg = xr; } }
        }
        size_t i = (m * (size_t)H + y) * W + x;
        d2out[i] = best;
        oxy[2 * i] = (int16_t)arg;
        oxy[2 * i + 1] = r[arg];
    }
}

// one wave per tile, one lane per cell: try the four forms with the feature of the tile's first
// cell and keep the first that reproduces d2 on every in-map cell of the tile
__global__ __launch_bounds__(64) void tile_table_kernel(const int32_t* __restrict__ d2in,
                                                        const int16_t* __restrict__ oxy,
                                                        uint32_t* __restrict__ tiles, int H, int W) {
    const int tpr = (W + 7) >> 3, tpc = (H + 7) >> 3;
    size_t m = blockIdx.y;
    int tile = blockIdx.x;
    int ty = tile / tpr, tx = tile - ty * tpr;
    int lane = threadIdx.x;
    int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    bool in_map = px < W && py < H;
    size_t base = m * (size_t)H * W;
    size_t i0 = base + (size_t)(ty * 8) * W + tx * 8;                      // first cell is always in the map
    int ox = oxy[2 * i0], oy = oxy[2 * i0 + 1];
    int d2 = in_map ? d2in[base + (size_t)py * W + px] : 0;
    uint32_t rec = 0;
    if (ox >= 0 && oy >= 0 && d2in[i0] < kDtInf * kDtInf) {
        int ddx = px - ox, ddy = py - oy;
        bool ok00 = !in_map || d2 == 0;                                    // solid tile
        bool ok01 = !in_map || d2 == ddy * ddy;                            // horizontal wall: dx == 0
        bool ok10 = !in_map || d2 == ddx * ddx;                            // vertical wall:   dy == 0
        bool ok11 = !in_map || d2 == ddx * ddx + ddy * ddy;                // corner cell
        const unsigned long long full = ~0ull;
        uint32_t feat = ((uint32_t)oy << 14) | (uint32_t)ox;
        if (__ballot(ok00) == full)      rec = kTileValid | kTileDx0 | kTileDy0;
        else if (__ballot(ok01) == full) rec = kTileValid | kTileDx0 | feat;
        else if (__ballot(ok10) == full) rec = kTileValid | kTileDy0 | feat;
        else if (__ballot(ok11) == full) rec = kTileValid | feat;
    }
    const size_t stride = ((size_t)tpr * tpc + 3) & ~(size_t)3;             // 16-byte granular per arena
    if (lane == 0) tiles[m * stride + tile] = rec;
}

// ============================================================================================
// a4: PyRayMarching.calc_range_many (env.py:425): one thread per query
// ============================================================================================
__global__ __launch_bounds__(256) void cast_static_kernel(const float* __restrict__ field, int H, int W,
                                                          const float* __restrict__ q, int n_per_env,
                                                          long long n_total, float max_range,
                                                          float* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    long long e = i / n_per_env;
    const float* f = field + (size_t)e * H * W;
    float dx, dy;
    nv::beam_dir(q[3 * i + 2], dx, dy);
    out[i] = nv::trace_ray(f, H, W, q[3 * i], q[3 * i + 1], dx, dy, max_range);
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

// ============================================================================================
// a1: the fused step.  One workgroup = one arena.
// ============================================================================================
struct StepShared {
    double rp[3];                 // robot pose being scanned
    double old_rp[3];             // robot pose at the start of the step (social force input)
    double act[2];                // action after the turning-radius clamp
    float lx, ly, lth;            // float32 lidar pose (env.py:386)
    double cT, sT;                // cos / sin of (double)lth for the beam-table fast path
    int i0, j0;                   // integer ray origin (env.py:419)
    int nseg, ndisc;
    int rescan;
    int respawn;
    float t1;                     // march parameter after the shared first probe (origin cell)
    float r_all;                  // >= 0: every beam has this raw range (origin occupied / no march)
    unsigned long long step_key;  // scan-noise counter of this step
    double wave_ratio[kMaxWaves];
};
constexpr size_t kPoolEnvBytes = 512;     // StepShared slot per arena in the step workspace
static_assert(sizeof(StepShared) <= kPoolEnvBytes, "StepShared must fit its workspace slot");
// Pedestrian scratch of the pedestrian variants of the kernel, carved out of dynamic LDS behind the scan's
// dir / rng area and sized by cfg.max_peds (N), not by the compiled maximum: 136 N + 32 bytes, so that a
// 20-pedestrian world still fits 8 arenas per CU (the static 64-pedestrian layout allowed 7).
struct PedShared {
    double *ax, *ay, *avx, *avy;             // [N + 1] agent positions / velocities at time t (robot last)
    float (*seg)[4];                         // [4 N] rectangle edges seen by the lidar ...
    float (*disc)[2];                        // [2 N] ... leg discs (stored right behind seg)
    float* info;                             // [6 N] merge_prims_culled_core scratch: in-range flag per primitive
};
__host__ __device__ inline size_t ped_lds_bytes(int N) { return (size_t)136 * N + 32; }
__device__ __forceinline__ PedShared ped_lds_carve(char* base, int N) {
    PedShared ps;
    double* d = (double*)base;
    ps.ax = d; ps.ay = d + (N + 1); ps.avx = d + 2 * (N + 1); ps.avy = d + 3 * (N + 1);
    float* f = (float*)(d + 4 * (N + 1));
    ps.seg = (float(*)[4])f;
    ps.disc = (float(*)[2])(f + 16 * N);
    ps.info = f + 20 * N;
    return ps;
}
struct Prims { const float (*seg)[4]; const float (*disc)[2]; float* info; };

template <int BLOCK>
__device__ __forceinline__ void finish_beams(const navsim_config& c, const StepShared& sh, const Prims pr,
                                             const double* __restrict__ tab, const float2* __restrict__ dir,
                                             const float* __restrict__ rng, float* __restrict__ rng_rw,
                                             const float* __restrict__ thr, const float* __restrict__ dthr,
                                             float* __restrict__ obs_row, int n_hist, float noise_std,
                                             uint64_t noise_key, uint64_t genv, int& crash, int& discomfort);

// robot scan (env.py:385-441 with other_agents = all pedestrians).  Writes the latest-scan slot
// of the observation row and every "not yet filled" stack slot (env.py:262-265).
//
// The march is latency-bound (each probe of the distance field is a dependent HBM/L2 access), so
// every thread advances R independent rays in lock-step: the R loads of one round are issued back
// to back before any of them is consumed, which multiplies the lines in flight per CU by R.
// Beam k of round-slot q is base + q*BLOCK + tid, so lanes of a wave hold adjacent beams (their
// probes fall on neighbouring cells and their range stores coalesce).
template <int BLOCK, int R, typename Field, bool TO_LDS>
__device__ __forceinline__ void scan_beams(const navsim_config& c, const StepShared& sh,
                                           const Field& field, const double* __restrict__ tab,
                                           const Prims pr, float2* __restrict__ dir_lds, float* __restrict__ rng_lds,
                                           const uint32_t* __restrict__ tiles,
                                           const float* __restrict__ thr, const float* __restrict__ dthr,
                                           float* __restrict__ obs_row, int n_hist, float noise_std,
                                           uint64_t noise_key, uint64_t genv,
                                           int& crash, int& discomfort) {
    const int B = c.n_beams, S = c.n_scan_stack, H = c.map_h, W = c.map_w;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const float res = (float)c.resolution;
    const float rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    const float x0 = (float)sh.i0, y0 = (float)sh.j0;
    const float lx = sh.lx, ly = sh.ly;
    const double lth = (double)sh.lth;
    const int nseg = sh.nseg, ndisc = sh.ndisc;
    int cr = 0, dc = 0;
#ifdef NAVSIM_CONTIGUOUS_FANS
    // each wave owns ONE contiguous fan of beams and walks it in 64-beam slices
    const int n_waves = BLOCK / 64;
    const int per_wave = (B + n_waves - 1) / n_waves;
    const int fan0 = ((int)threadIdx.x >> 6) * per_wave;
    const int fan1 = (fan0 + per_wave < B) ? fan0 + per_wave : B;
#define NAVSIM_BEAM_OF(base_, q_) (fan0 + (base_) / n_waves + (q_) * 64 + ((int)threadIdx.x & 63))
#define NAVSIM_BEAM_OK(k_) ((k_) < fan1)
    for (int base = 0; base < per_wave * n_waves; base += BLOCK * R) {
#else
#define NAVSIM_BEAM_OF(base_, q_) ((base_) + (q_) * BLOCK + (int)threadIdx.x)
#define NAVSIM_BEAM_OK(k_) ((k_) < B)
    for (int base = 0; base < B; base += BLOCK * R) {
#endif
        float dx[R], dy[R], t[R], r[R];
        unsigned active = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            int k = NAVSIM_BEAM_OF(base, q);
            bool valid = NAVSIM_BEAM_OK(k);
            const int kk = valid ? k : 0;
            const double lin = nv::linspace_k(c, kk, step);
            double ang = lin + lth;                                     // env.py:388-390
            float heading = (float)ang;                                 // env.py:424
            bool fast = false;
            if (tab) {
                // heading = lin + lth + delta EXACTLY: (heading - ang) is exact (Sterbenz), and the
                // rounding error of the float64 sum is recovered by TwoSum
                double bb = ang - lin;
                double eps = (lin - (ang - bb)) + (lth - bb);
                double delta = ((double)heading - ang) + eps;
                double2 cs = ((const double2*)tab)[kk];
                fast = nv::beam_dir_from_table(cs.x, cs.y, sh.cT, sh.sT, delta, dx[q], dy[q]);
            }
            if (!fast) nv::beam_dir(heading, dx[q], dy[q]);
            t[q] = sh.t1;                              // the t = 0 probe (origin cell) was taken once
            r[q] = (sh.r_all >= 0.0f) ? sh.r_all : max_range;
            active |= (valid && sh.r_all < 0.0f) ? (1u << q) : 0u;
        }
        // range_libc RayMarching::calc_range (env.py:425), R rays per thread in lock-step.  The hit
        // distance is evaluated once after the march (hx, hy), not speculatively in every round.
        int hx[R], hy[R];
        unsigned hit = 0;
        const unsigned uW = (unsigned)W, uH = (unsigned)H;
        if (R == 1 && tiles) {                               // LDS tile table: two-phase march
            r[0] = march_ray_tiles(field, tiles, (W + 7) >> 3, x0, y0, dx[0], dy[0], t[0], max_range, uW, uH,
                                   (active & 1u) != 0u);
            if (sh.r_all >= 0.0f) r[0] = sh.r_all;
            active = 0;
        }
        while (active) {
            int px[R], py[R];
            typename Field::raw_t raw[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                float fx = x0 + dx[q] * t[q];
                float fy = y0 + dy[q] * t[q];
                px[q] = (int)fx;
                py[q] = (int)fy;
                // px >= W || px < 0 || py < 0 || py >= H, as two unsigned compares
                bool inb = ((unsigned)px[q] < uW) & ((unsigned)py[q] < uH);
                bool a = (active >> q) & 1u;
                if (a && !inb) active &= ~(1u << q);                    // left the map: max_range
                bool live = a && inb;
                px[q] = live ? px[q] : 0;
                py[q] = live ? py[q] : 0;
                raw[q] = field.load(px[q], py[q]);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if ((active >> q) & 1u) {
                    if (field.occupied(raw[q])) {
                        hx[q] = px[q]; hy[q] = py[q];
                        hit |= 1u << q;
                        active &= ~(1u << q);
                    } else {
                        t[q] += field.step_of(raw[q], px[q], py[q]);
                        if (!(t[q] < max_range)) active &= ~(1u << q);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if ((hit >> q) & 1u) {
                float xd = (float)hx[q] - x0;
                float yd = (float)hy[q] - y0;
                r[q] = sqrtf(xd * xd + yd * yd);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            int k = NAVSIM_BEAM_OF(base, q);
            if (TO_LDS) {                                               // pedestrian variants: merge later, culled
                if (NAVSIM_BEAM_OK(k)) { rng_lds[k] = r[q]; dir_lds[k] = make_float2(dx[q], dy[q]); }
            } else if (NAVSIM_BEAM_OK(k)) {
                float rr = r[q] * res;                                  // env.py:426
                for (int p = 0; p < nseg; ++p)
                    nv::seg_merge(rr, lx, ly, dx[q], dy[q], pr.seg[p][0], pr.seg[p][1], pr.seg[p][2], pr.seg[p][3]);
                for (int p = 0; p < ndisc; ++p)
                    nv::circle_merge(rr, lx, ly, dx[q], dy[q], pr.disc[p][0], pr.disc[p][1], nv::kLegRadius);
                rr = rr < 0.0f ? 0.0f : rr;                             // env.py:435
                rr = rr > rmax ? rmax : rr;
                if (noise_std > 0.0f && rr != rmax)                     // env.py:437-440
                    rr = rr + noise_std * nv::gauss_noise(c.seed ^ noise_key, genv, noise_key, (uint32_t)k);
                cr |= (rr < thr[k]);
                dc |= (rr < dthr[k]);
                obs_row[(size_t)(S - 1) * B + k] = rr;
                for (int j = 0; j < S - 1; ++j)
                    if (S - 1 - j > n_hist) obs_row[(size_t)j * B + k] = rr;
            }
        }
    }
    if (TO_LDS) {
        __syncthreads();
        finish_beams<BLOCK>(c, sh, pr, tab, dir_lds, rng_lds, rng_lds, thr, dthr, obs_row, n_hist, noise_std,
                            noise_key, genv, cr, dc);
    }
    crash = cr;
    discomfort = dc;
}

// one social-force term of the build-defined pedestrian model: force on agent
// i from agent j; (0, 0) when the pair is skipped
__device__ __forceinline__ void sfm_pair(const navsim_config& c, double xi, double yi, double vxi, double vyi,
                                         double xj, double yj, double vxj, double vyj, double& fx, double& fy) {
    fx = 0.0; fy = 0.0;
    double dxx = xj - xi, dyy = yj - yi;
    double dist = sqrt(dxx * dxx + dyy * dyy);
    if (dist < 1e-9) return;
    double ddx = dxx / dist, ddy = dyy / dist;
    double ivx = c.sfm_lambda * (vxi - vxj) + ddx;
    double ivy = c.sfm_lambda * (vyi - vyj) + ddy;
    double il = sqrt(ivx * ivx + ivy * ivy);
    if (il < 1e-9) return;
    double idx = ivx / il, idy = ivy / il;
    double theta = nv::atan2_(idx * ddy - idy * ddx, idx * ddx + idy * ddy);
    double Bq = c.sfm_gamma * il;
    double a1 = c.sfm_n_prime * Bq * theta;
    double a2 = c.sfm_n * Bq * theta;
    double fv = -nv::exp_neg(-dist / Bq - a1 * a1);
    double sgn = (theta > 0.0) ? 1.0 : ((theta < 0.0) ? -1.0 : 0.0);
    double fa = -sgn * nv::exp_neg(-dist / Bq - a2 * a2);
    fx = fv * idx + fa * (-idy);
    fy = fv * idy + fa * idx;
}

// direction of beam k (env.py:388-390, 424): table fast path with proven rounding, else full sincos
__device__ __forceinline__ void beam_dir_k(const navsim_config& c, const double* __restrict__ tab, int k,
                                           double step, double lth, double cT, double sT,
                                           float& dx, float& dy) {
    const double lin = nv::linspace_k(c, k, step);
    double ang = lin + lth;
    float heading = (float)ang;
    bool fast = false;
    if (tab) {
        // heading = lin + lth + delta EXACTLY: (heading - ang) is exact (Sterbenz), and the
        // rounding error of the float64 sum is recovered by TwoSum
        double bb = ang - lin;
        double eps = (lin - (ang - bb)) + (lth - bb);
        double delta = ((double)heading - ang) + eps;
        double2 cs = ((const double2*)tab)[k];
        fast = nv::beam_dir_from_table(cs.x, cs.y, cT, sT, delta, dx, dy);
    }
    if (!fast) nv::beam_dir(heading, dx, dy);
}

// the t = 0 sample of calc_range is the origin cell for every beam: take it once per scan
template <typename Field>
__device__ __forceinline__ void first_probe(const Field& field, int i0, int j0, float max_range,
                                            float& t1, float& r_all) {
    t1 = 0.0f; r_all = -1.0f;
    typename Field::raw_t raw0 = field.load(i0, j0);               // origin is clipped into the map
    if (field.occupied(raw0)) { r_all = 0.0f; return; }            // sqrtf(0): starts inside an obstacle
    float d0 = field.decode(raw0, i0, j0);
    float stp = (float)((double)d0 * 0.999);
    t1 = (stp > 1.0f) ? stp : 1.0f;
    if (!(t1 < max_range)) r_all = max_range;
}

// Same march with the arena's analytic tile table in LDS (navsim_build_tiles): a probe whose tile
// has a valid record gets its exact d2 from one LDS read and integer arithmetic; only probes in
// mixed tiles read the field (~35 % of the probes, ~3x fewer distinct lines per arena).
// d2 from a record is the exact integer the field holds, so the sampled sequence is unchanged.
// (A two-phase form -- lanes run ahead through valid tiles, then load together -- was measured
// slower: in lock-step the run-ahead iterations of a few lanes stall the whole wave.)
template <typename Field>
__device__ __forceinline__ float march_ray_tiles(const Field& field, const uint32_t* __restrict__ tiles, int tpr,
                                                 float x0, float y0, float dx, float dy, float t,
                                                 float max_range, unsigned uW, unsigned uH, bool alive) {
    float result = max_range;
    while (alive) {
        float fx = x0 + dx * t;
        float fy = y0 + dy * t;
        int px = (int)fx, py = (int)fy;
        if (!(((unsigned)px < uW) & ((unsigned)py < uH))) break;           // left the map
        unsigned rec = tiles[(py >> 3) * tpr + (px >> 3)];
        float d;
        bool occ;
        if (rec & kTileValid) {                                            // analytic: no memory access
            int ddx = (rec & kTileDx0) ? 0 : px - (int)(rec & 0x3FFFu);
            int ddy = (rec & kTileDy0) ? 0 : py - (int)((rec >> 14) & 0x3FFFu);
            int d2 = ddx * ddx + ddy * ddy;
            occ = d2 == 0;
            d = nv::sqrt_small_int((float)(d2 | (int)occ));
        } else {                                                           // mixed tile: read the field
            typename Field::raw_t raw = field.load(px, py);
            occ = field.occupied(raw);
            d = occ ? 1.0f : field.decode(raw, px, py);
        }
        if (occ) {
            float xd = (float)px - x0, yd = (float)py - y0;
            result = sqrtf(xd * xd + yd * yd);
            break;
        }
        float stp = (float)((double)d * 0.999);
        t += (stp > 1.0f) ? stp : 1.0f;
        if (!(t < max_range)) break;
    }
    return result;
}

// one ray of calc_range from t = t1 on (env.py:425); returns the raw range in cells
template <typename Field>
__device__ __forceinline__ float march_ray(const Field& field, float x0, float y0, float dx, float dy,
                                           float t, float max_range, unsigned uW, unsigned uH) {
    for (;;) {
        float fx = x0 + dx * t;
        float fy = y0 + dy * t;
        int px = (int)fx, py = (int)fy;
        if (!(((unsigned)px < uW) & ((unsigned)py < uH))) return max_range;     // left the map
        typename Field::raw_t raw = field.load(px, py);
        if (field.occupied(raw)) {
            float xd = (float)px - x0;
            float yd = (float)py - y0;
            return sqrtf(xd * xd + yd * yd);
        }
        t += field.step_of(raw, px, py);
        if (!(t < max_range)) return max_range;
    }
}

// Pedestrians into the scan (env.py:428-432), culled by bearing.  A rectangle side or a leg disc is
// seen under a small angle, so instead of testing every beam against every primitive (B x P ray
// tests: 3x the cost of the whole map march at 20 pedestrians) each wave takes one primitive, derives
// the beam-index interval that can possibly hit it (bearing +- half-width, two beams of margin, all
// three 2*pi aliases) and runs the SAME float32 seg_merge / circle_merge on those beams only;
// results land with an LDS atomicMin on the (non-negative) float bits, so they do not depend on the
// order of primitives.  rng[] holds metres, dir[] the beam directions.
// One lane per primitive: can it change the clipped scan at all?  Every point of a segment is at least
// |u| - |v - u| from the lidar (a disc: |u| - r); beyond the clip range a hit cannot matter, because
// clip(min(r, t)) = clip(r) for t >= range_max.  info[p] < 0 marks such a primitive.
template <int BLOCK>
__device__ __forceinline__ void prim_in_range(int nprim, int nseg, float lx, float ly, float rcull, const Prims pr) {
    for (int p = (int)threadIdx.x; p < nprim; p += BLOCK) {
        bool skip;
        if (p < nseg) {
            float ux = pr.seg[p][0] - lx, uy = pr.seg[p][1] - ly, vx = pr.seg[p][2] - lx, vy = pr.seg[p][3] - ly;
            skip = sqrtf(ux * ux + uy * uy) - sqrtf((vx - ux) * (vx - ux) + (vy - uy) * (vy - uy)) > rcull;
        } else {
            float ux = pr.disc[p - nseg][0] - lx, uy = pr.disc[p - nseg][1] - ly;
            skip = sqrtf(ux * ux + uy * uy) - nv::kLegRadius > rcull;
        }
        pr.info[p] = skip ? -1.0f : 1.0f;
    }
}

template <int BLOCK>
__device__ __forceinline__ void merge_prims_culled_core(int B, float lx, float ly, float stepf, float beta0,
                                                        int nseg, int ndisc, const Prims pr,
                                                        const float2* __restrict__ dir, float* __restrict__ rng,
                                                        float rcull) {
    const int lane = (int)threadIdx.x & 63;
    const float kTwoPiF = 6.2831853f;
    const float Kf = (stepf > 0.0f) ? kTwoPiF / stepf : 0.0f;
    const int nprim = nseg + ndisc;
    prim_in_range<BLOCK>(nprim, nseg, lx, ly, rcull, pr);
    __syncthreads();
    // eight lanes per primitive, eight primitives per wavefront at a time (a pedestrian a few metres away
    // spans 10-50 beams; measured 4 / 8 / 16 / 32 / 64 lanes: c3 11.81 / 11.80 / 11.68 / 11.13 / 10.26 M env-steps/s)
#ifndef NAVSIM_MERGE_G
#define NAVSIM_MERGE_G 8
#endif
    constexpr int G = NAVSIM_MERGE_G;
    const int sub = lane & (G - 1);
    for (int p = ((int)threadIdx.x) / G; p < nprim; p += BLOCK / G) {
        if (pr.info[p] < 0.0f) continue;                            // beyond the clip range
        const bool is_seg = p < nseg;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float ac = 0.0f, w = 0.0f;
        bool full = (stepf <= 0.0f);
        if (is_seg) {
            a0 = pr.seg[p][0]; a1 = pr.seg[p][1]; a2 = pr.seg[p][2]; a3 = pr.seg[p][3];
            float ux = a0 - lx, uy = a1 - ly, vx = a2 - lx, vy = a3 - ly;
            float b1 = atan2f(uy, ux), b2 = atan2f(vy, vx);
            float d = b2 - b1;
            d -= kTwoPiF * floorf(d / kTwoPiF + 0.5f);              // (-pi, pi]
            ac = b1 + 0.5f * d;
            w = 0.5f * fabsf(d);
            if (fabsf(d) > 3.0f || ux * ux + uy * uy < 1e-6f || vx * vx + vy * vy < 1e-6f) full = true;
        } else {
            a0 = pr.disc[p - nseg][0]; a1 = pr.disc[p - nseg][1];
            float ux = a0 - lx, uy = a1 - ly;
            float dist = sqrtf(ux * ux + uy * uy);
            if (dist <= nv::kLegRadius * 1.05f) full = true;
            else { ac = atan2f(uy, ux); w = asinf(fminf(1.0f, nv::kLegRadius / dist)); }
        }
        float rel = ac - beta0;
        rel -= kTwoPiF * floorf(rel / kTwoPiF + 0.5f);              // [-pi, pi)
        const float klo = (rel - w) / stepf - 2.0f, khi = (rel + w) / stepf + 2.0f;
        for (int m = full ? 0 : -1; m <= (full ? 0 : 1); ++m) {
            int k0 = full ? 0 : (int)floorf(klo + (float)m * Kf);
            int k1 = full ? B - 1 : (int)ceilf(khi + (float)m * Kf);
            k0 = k0 < 0 ? 0 : k0;
            k1 = k1 > B - 1 ? B - 1 : k1;
            for (int k = k0 + sub; k <= k1; k += G) {
                float2 d = dir[k];
                float old = rng[k], rr = old;
                if (is_seg) nv::seg_merge(rr, lx, ly, d.x, d.y, a0, a1, a2, a3);
                else        nv::circle_merge(rr, lx, ly, d.x, d.y, a0, a1, nv::kLegRadius);
                if (rr < old) atomicMin((int*)&rng[k], __float_as_int(rr));
            }
        }
    }
}

template <int BLOCK>
__device__ __forceinline__ void merge_prims_culled(const navsim_config& c, const StepShared& sh, const Prims pr,
                                                   const float2* __restrict__ dir, float* __restrict__ rng) {
    merge_prims_culled_core<BLOCK>(c.n_beams, sh.lx, sh.ly, (float)nv::linspace_step(c),
                                   (float)(c.angle_min + (double)sh.lth), sh.nseg, sh.ndisc, pr, dir, rng,
                                   (float)c.range_max * 1.0001f + 0.01f);
}

// raw ranges (cells) -> metres, pedestrians, clip, noise, crash / discomfort flags, observation row
// (env.py:426-440 + the stack fill of env.py:262-265).  `dir` may be NULL: directions are then
// recomputed (same function, same values) for the beams that need them.
template <int BLOCK>
__device__ __forceinline__ void finish_beams(const navsim_config& c, const StepShared& sh, const Prims pr,
                                             const double* __restrict__ tab, const float2* __restrict__ dir,
                                             const float* __restrict__ rng, float* __restrict__ rng_rw,
                                             const float* __restrict__ thr, const float* __restrict__ dthr,
                                             float* __restrict__ obs_row, int n_hist, float noise_std,
                                             uint64_t noise_key, uint64_t genv, int& crash, int& discomfort) {
    const int B = c.n_beams, S = c.n_scan_stack;
    const float res = (float)c.resolution;
    const float rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    const float lx = sh.lx, ly = sh.ly;
    const int nseg = sh.nseg, ndisc = sh.ndisc;
    const float r_all = sh.r_all;
    int cr = 0, dc = 0;
    const bool culled = dir && rng_rw && (nseg | ndisc);           // LDS-resident: bearing-culled merge
    if (culled) {
        for (int k = (int)threadIdx.x; k < B; k += BLOCK)
            rng_rw[k] = ((r_all >= 0.0f) ? r_all : rng[k]) * res;   // env.py:426
        __syncthreads();
        merge_prims_culled<BLOCK>(c, sh, pr, dir, rng_rw);
        __syncthreads();
    }
    for (int k = (int)threadIdx.x; k < B; k += BLOCK) {
        float rr = culled ? rng[k] : ((r_all >= 0.0f) ? r_all : rng[k]) * res;
        if (!culled && (nseg | ndisc)) {
            float dx, dy;
            if (dir) { float2 d = dir[k]; dx = d.x; dy = d.y; }
            else beam_dir_k(c, tab, k, step, (double)sh.lth, sh.cT, sh.sT, dx, dy);
            for (int p = 0; p < nseg; ++p)
                nv::seg_merge(rr, lx, ly, dx, dy, pr.seg[p][0], pr.seg[p][1], pr.seg[p][2], pr.seg[p][3]);
            for (int p = 0; p < ndisc; ++p)
                nv::circle_merge(rr, lx, ly, dx, dy, pr.disc[p][0], pr.disc[p][1], nv::kLegRadius);
        }
        rr = rr < 0.0f ? 0.0f : rr;                                 // env.py:435
        rr = rr > rmax ? rmax : rr;
        if (noise_std > 0.0f && rr != rmax)                         // env.py:437-440
            rr = rr + noise_std * nv::gauss_noise(c.seed ^ noise_key, genv, noise_key, (uint32_t)k);
        cr |= (rr < thr[k]);
        dc |= (rr < dthr[k]);
        obs_row[(size_t)(S - 1) * B + k] = rr;
        for (int j = 0; j < S - 1; ++j)
            if (S - 1 - j > n_hist) obs_row[(size_t)j * B + k] = rr;
    }
    crash = cr;
    discomfort = dc;
}

// Predicated one-ray-per-lane scan (R == 11 variant): the march loop has ONE wave-level branch
// (any lane still marching?) instead of a divergent if-ladder per probe; finished or out-of-map
// lanes keep executing with their updates masked off.  Same results as scan_beams.
template <int BLOCK, typename Field>
__device__ __forceinline__ void scan_beams_pred(const navsim_config& c, const StepShared& sh,
                                                const Field& field, const double* __restrict__ tab,
                                                const Prims pr,
                                                const float* __restrict__ thr, const float* __restrict__ dthr,
                                                float* __restrict__ obs_row, int n_hist, float noise_std,
                                                uint64_t noise_key, uint64_t genv,
                                                int& crash, int& discomfort) {
    const int B = c.n_beams, S = c.n_scan_stack, H = c.map_h, W = c.map_w;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const float res = (float)c.resolution;
    const float rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    const float x0 = (float)sh.i0, y0 = (float)sh.j0;
    const float lx = sh.lx, ly = sh.ly;
    const int nseg = sh.nseg, ndisc = sh.ndisc;
    const unsigned uW = (unsigned)W, uH = (unsigned)H;
    const float t1 = sh.t1, r_all = sh.r_all;
    int cr = 0, dc = 0;
    for (int k = (int)threadIdx.x; k < B; k += BLOCK) {
        float dx, dy;
        beam_dir_k(c, tab, k, step, (double)sh.lth, sh.cT, sh.sT, dx, dy);
        float t = t1;
        bool active = r_all < 0.0f;
        bool hit = false;
        int hx = 0, hy = 0;
        while (__any(active)) {
            float fx = x0 + dx * t;
            float fy = y0 + dy * t;
            int px = (int)fx, py = (int)fy;
            bool live = active & ((unsigned)px < uW) & ((unsigned)py < uH);
            px = live ? px : 0;
            py = live ? py : 0;
            typename Field::raw_t raw = field.load(px, py);
            bool occ = live & field.occupied(raw);
            hx = occ ? px : hx;
            hy = occ ? py : hy;
            hit |= occ;
            float tn = t + field.step_of(raw, px, py);
            bool go = live & !occ;
            t = go ? tn : t;
            active = go & (tn < max_range);
        }
        float rr = (r_all >= 0.0f) ? r_all : max_range;
        if (hit) {
            float xd = (float)hx - x0;
            float yd = (float)hy - y0;
            rr = sqrtf(xd * xd + yd * yd);
        }
        rr = rr * res;                                          // env.py:426
        for (int p = 0; p < nseg; ++p)
            nv::seg_merge(rr, lx, ly, dx, dy, pr.seg[p][0], pr.seg[p][1], pr.seg[p][2], pr.seg[p][3]);
        for (int p = 0; p < ndisc; ++p)
            nv::circle_merge(rr, lx, ly, dx, dy, pr.disc[p][0], pr.disc[p][1], nv::kLegRadius);
        rr = rr < 0.0f ? 0.0f : rr;                             // env.py:435
        rr = rr > rmax ? rmax : rr;
        if (noise_std > 0.0f && rr != rmax)                     // env.py:437-440
            rr = rr + noise_std * nv::gauss_noise(c.seed ^ noise_key, genv, noise_key, (uint32_t)k);
        cr |= (rr < thr[k]);
        dc |= (rr < dthr[k]);
        obs_row[(size_t)(S - 1) * B + k] = rr;
        for (int j = 0; j < S - 1; ++j)
            if (S - 1 - j > n_hist) obs_row[(size_t)j * B + k] = rr;
    }
    crash = cr;
    discomfort = dc;
}

// Wave-dynamic robot scan (R == 0 variants; the default).  Same results as scan_beams, different
// schedule, built on what the profiles showed (profiles/README.md):
//   * lanes of a wave finish their rays after very different numbers of probes (mean 7.9, wave
//     maximum 13.5), so with one fixed ray per lane 45 % of the issue slots idle.  Here every wave
//     owns a contiguous fan of beams and a lane that finishes a ray immediately takes the next
//     unassigned beam of its wave (ballot + popcount, no atomics) -- "persistent lanes";
//   * beam directions are produced once per scan by a coalesced pass into LDS, so a refill is one
//     ds_read_b64; ranges go back to LDS and a second coalesced pass merges pedestrians, clips,
//     adds noise, raises the crash / discomfort flags and stores the observation row;
//   * the first probe (the robot's own cell) is identical for every beam and is taken once.
template <int BLOCK, typename Field>
__device__ __forceinline__ void scan_beams_dyn(const navsim_config& c, StepShared& sh,
                                               const Field& field, const double* __restrict__ tab,
                                               const Prims pr, float2* __restrict__ dir, float* __restrict__ rng,
                                               const float* __restrict__ thr, const float* __restrict__ dthr,
                                               float* __restrict__ obs_row, int n_hist, float noise_std,
                                               uint64_t noise_key, uint64_t genv,
                                               int& crash, int& discomfort) {
    const int B = c.n_beams, H = c.map_h, W = c.map_w;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const double step = nv::linspace_step(c);
    const float x0 = (float)sh.i0, y0 = (float)sh.j0;
    const int tid = (int)threadIdx.x;

    // ---- pass 1: beam directions -> LDS
    for (int k = tid; k < B; k += BLOCK) {
        float dx, dy;
        beam_dir_k(c, tab, k, step, (double)sh.lth, sh.cT, sh.sT, dx, dy);
        dir[k] = make_float2(dx, dy);
    }
    const float t1 = sh.t1, r_all = sh.r_all;                // first probe, taken by thread 0 earlier
    __syncthreads();

    // ---- pass 2: march (env.py:425), persistent lanes
    if (r_all < 0.0f) {
        const int lane = tid & 63, wave = tid >> 6;
        const int per = (B + (BLOCK / 64) - 1) / (BLOCK / 64);
        int next = wave * per;
        const int end = (next + per < B) ? next + per : B;
        const unsigned long long lt = (1ull << lane) - 1ull;
        const unsigned uW = (unsigned)W, uH = (unsigned)H;
        bool active = false;
        int k = 0;
        float t = 0.0f, dx = 0.0f, dy = 0.0f;
        for (;;) {
            unsigned long long idle = __ballot(!active);
            if (next < end && idle) {
                int my = next + __popcll(idle & lt);
                if (!active && my < end) {
                    k = my;
                    float2 d = dir[k];
                    dx = d.x; dy = d.y;
                    t = t1;
                    active = true;
                }
                next += __popcll(idle);
            }
            if (!__any(active)) break;
            if (active) {
                float fx = x0 + dx * t;
                float fy = y0 + dy * t;
                int px = (int)fx, py = (int)fy;
                if (!(((unsigned)px < uW) & ((unsigned)py < uH))) {
                    rng[k] = max_range;                             // left the map
                    active = false;
                } else {
                    typename Field::raw_t raw = field.load(px, py);
                    if (field.occupied(raw)) {
                        float xd = (float)px - x0;
                        float yd = (float)py - y0;
                        rng[k] = sqrtf(xd * xd + yd * yd);
                        active = false;
                    } else {
                        t += field.step_of(raw, px, py);
                        if (!(t < max_range)) { rng[k] = max_range; active = false; }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- pass 3
    finish_beams<BLOCK>(c, sh, pr, tab, dir, rng, rng, thr, dthr, obs_row, n_hist, noise_std, noise_key, genv,
                        crash, discomfort);
}

// ============================================================================================
// Pool scan: the march of ALL arenas as one flat pool of 64-beam wave tasks.  No workgroup barrier,
// no idle waves waiting for an arena's slowest fan: a CU always holds 32 marching waves, whatever
// the number of arenas.  Reads each arena's scan request (StepShared slot in the workspace), writes
// raw ranges in cells.  `only_flagged`: the re-scan after a crash revert / respawn (env.py:718-723).
// ============================================================================================
template <typename Field>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void pool_scan_kernel(navsim_config c, navsim_state st, char* __restrict__ ws_env, float* __restrict__ ranges,
                      int only_flagged, unsigned n_blocks_logical) {
    // XCD-aware block order: consecutive logical blocks (the fans of one arena) share an XCD's L2
    const unsigned nb = gridDim.x;
    unsigned bid = blockIdx.x;
    if ((nb & 7u) == 0u) bid = (bid & 7u) * (nb >> 3) + (bid >> 3);
    if (bid >= n_blocks_logical) return;
    const int B = c.n_beams;
    const int G = (B + 63) >> 6;
    const unsigned task = bid * 4u + (threadIdx.x >> 6);
    const int e = (int)(task / (unsigned)G);
    if (e >= c.n_envs) return;
    const int g = (int)(task - (unsigned)e * (unsigned)G);
    const StepShared* __restrict__ s = (const StepShared*)(ws_env + (size_t)e * kPoolEnvBytes);
    if (only_flagged && !s->rescan) return;
    const float r_all = s->r_all;
    if (r_all >= 0.0f) return;                                   // finish_beams supplies the range
    const int k = g * 64 + (int)(threadIdx.x & 63);
    if (k >= B) return;
    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, c.map_h, c.map_w);
    float dx, dy;
    beam_dir_k(c, st.beam_table, k, nv::linspace_step(c), (double)s->lth, s->cT, s->sT, dx, dy);
    const float max_range = march_limit(c.map_h, c.map_w, c.range_max, c.resolution);
    ranges[(size_t)e * B + k] = march_ray(field, (float)s->i0, (float)s->j0, dx, dy, s->t1, max_range,
                                          (unsigned)c.map_w, (unsigned)c.map_h);
}

// MODE 0: the whole step in one launch.  MODE 1 / 2 / 3: the same code cut at the scan, for the
// pooled schedule (navsim_step with a workspace): 1 = everything before the scan, then the arena's
// StepShared (+ pedestrian primitives) is parked in the workspace; pool_scan_kernel marches;
// 2 = flags, reward / done / info, relocation decision and -- unless the arena must be re-scanned --
// the observation row and state; 3 = the same tail for re-scanned arenas.
enum { kModeFused = 0, kModePre = 1, kModePost = 2, kModeFinal = 3 };

// Phase 1 for the pedestrians of one arena (env.py:617-693 with the build-defined social force or external
// commands): waypoint pop, forces / integration, new goal, leg odometry, state.  Called by every thread of the
// workgroup (it synchronises); pedestrian i lives on thread i.  Shared by the fused step kernel and by
// ped_update_kernel, which runs it on one wavefront per arena ahead of the step.
template <int BLOCK, typename Field>
__device__ __forceinline__ void ped_phase(const navsim_config& c, const navsim_state& st, const Field& field, int e,
                                          int n, int tid, bool is_ped, size_t pq, double dt, uint64_t genv,
                                          uint64_t steps_now, const double* old_rp, double prev_v, const PedShared& ps,
                                          char* pair_scratch, unsigned pair_bytes, double (&pp)[3], double (&pvel)[2]) {
    const int N = c.max_peds, P = NAVSIM_MAX_WAYPOINTS;
    (void)N;
    double* wp = st.ped_waypoints ? st.ped_waypoints + (pq * P) * 2 : nullptr;
    int nw = 1;
    if (is_ped) {
        nw = st.ped_n_waypoints[pq];
        while (nw > 1) {                                   // env.py:633-642
            double ddx = pp[0] - wp[0], ddy = pp[1] - wp[1];
            if (sqrt(ddx * ddx + ddy * ddy) < 1.0) {
                for (int k = 0; k + 1 < nw; ++k) { wp[2 * k] = wp[2 * k + 2]; wp[2 * k + 1] = wp[2 * k + 3]; }
                nw -= 1;
            } else break;
        }
    }
    if (c.ped_model == NAVSIM_PED_SFM) {
        // stage every agent's position / velocity at time t (pedestrians, then the robot)
        if (is_ped) { ps.ax[tid] = pp[0]; ps.ay[tid] = pp[1]; ps.avx[tid] = pvel[0]; ps.avy[tid] = pvel[1]; }
        if (tid == 0) {
            double s, cs;
            nv::sincos(old_rp[2], s, cs);
            ps.ax[n] = old_rp[0]; ps.ay[n] = old_rp[1];
            ps.avx[n] = prev_v * cs; ps.avy[n] = prev_v * s;
        }
        __syncthreads();
        // the n*(n+1) pair terms are independent: spread them over the whole workgroup (LDS scratch
        // = the scan's dir/rng area, free until the march), then every pedestrian adds its row in
        // partner order -- the same sums, in the same order, as a sequential loop
        double2* pair = (double2*)pair_scratch;
        const bool pair_par = pair_bytes >= (unsigned)(n * (n + 1)) * sizeof(double2) && n > 1;
        if (pair_par) {
            for (int t = tid; t < n * (n + 1); t += BLOCK) {
                int i = t / (n + 1), j = t - i * (n + 1);
                double fx = 0.0, fy = 0.0;
                if (j != i)
                    sfm_pair(c, ps.ax[i], ps.ay[i], ps.avx[i], ps.avy[i], ps.ax[j], ps.ay[j], ps.avx[j], ps.avy[j], fx, fy);
                pair[t] = make_double2(fx, fy);
            }
            __syncthreads();
        }
        if (is_ped) {
            const int i = tid;
            double vpref = st.ped_v_pref[pq];
            double ex = wp[0] - ps.ax[i], ey = wp[1] - ps.ay[i];
            double L = sqrt(ex * ex + ey * ey);
            if (L > 1e-9) { ex = ex / L; ey = ey / L; } else { ex = 0.0; ey = 0.0; }
            double fdx = (vpref * ex - ps.avx[i]) / c.sfm_tau;
            double fdy = (vpref * ey - ps.avy[i]) / c.sfm_tau;
            double fsx = 0.0, fsy = 0.0;
            for (int j = 0; j <= n; ++j) {
                if (j == i) continue;
                double fx, fy;
                if (pair_par) { double2 f = pair[i * (n + 1) + j]; fx = f.x; fy = f.y; }
                else sfm_pair(c, ps.ax[i], ps.ay[i], ps.avx[i], ps.avy[i], ps.ax[j], ps.ay[j], ps.avx[j], ps.avy[j], fx, fy);
                fsx += fx;
                fsy += fy;
            }
            double fox = 0.0, foy = 0.0;
            {
                const int H = c.map_h, W = c.map_w;
                int ci, cj;
                nv::xy_to_ij(ps.ax[i], ps.ay[i], c, ci, cj);
                ci = ci > W - 1 ? W - 1 : ci;
                cj = cj > H - 1 ? H - 1 : cj;
                int il_ = ci > 0 ? ci - 1 : 0, ir = ci < W - 1 ? ci + 1 : W - 1;
                int jl = cj > 0 ? cj - 1 : 0, jr = cj < H - 1 ? cj + 1 : H - 1;
                double d = (double)field.at(ci, cj) * c.resolution;
                double gx = (double)field.at(ir, cj) - (double)field.at(il_, cj);
                double gy = (double)field.at(ci, jr) - (double)field.at(ci, jl);
                double gl = sqrt(gx * gx + gy * gy);
                if (gl > 0.0) {
                    double mag = nv::exp_neg(-(d - c.sfm_agent_radius) / c.sfm_sigma_obstacle);
                    fox = mag * (gx / gl);
                    foy = mag * (gy / gl);
                }
            }
            double accx = c.sfm_k_desired * fdx + c.sfm_k_social * fsx + c.sfm_k_obstacle * fox;
            double accy = c.sfm_k_desired * fdy + c.sfm_k_social * fsy + c.sfm_k_obstacle * foy;
            double vx = ps.avx[i] + accx * dt;
            double vy = ps.avy[i] + accy * dt;
            double sp = sqrt(vx * vx + vy * vy);
            if (sp > vpref) {
                double k = (sp > 0.0) ? vpref / sp : 0.0;
                vx = vx * k; vy = vy * k;
            }
            pp[0] = pp[0] + vx * dt;
            pp[1] = pp[1] + vy * dt;
            double sp2 = sqrt(vx * vx + vy * vy);
            if (sp2 > 1e-6) pp[2] = nv::mod_2pi(nv::atan2_(vy, vx));
            pvel[0] = vx; pvel[1] = vy;
        }
    } else if (c.ped_model == NAVSIM_PED_EXTERNAL && is_ped) {
        const double* cmd = st.ped_cmd + pq * 2;
        nv::set_vel(pp, cmd[0], cmd[1], dt, 0.0, pvel);     // env.py:662
    }
    if (is_ped) {
        // ---- new goal at the final waypoint (env.py:667-680): table draw, or wait for navsim_replan
        double ddx = pp[0] - wp[2 * (nw - 1)], ddy = pp[1] - wp[2 * (nw - 1) + 1];
        if (sqrt(ddx * ddx + ddy * ddy) < 0.5 && c.n_spawn > 0 && st.spawn_pose && !st.costmap) {
            uint64_t h = nv::hash4(c.seed, genv, (uint64_t)tid + 1000, steps_now);
            for (int tries = 0; tries < c.n_spawn; ++tries) {
                int idx = (int)((h + (uint64_t)tries) % (uint64_t)c.n_spawn);
                const double* cand = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
                double gx = cand[0] - pp[0], gy = cand[1] - pp[1];
                if (sqrt(gx * gx + gy * gy) > 10.0) {
                    wp[0] = cand[0]; wp[1] = cand[1]; nw = 1;
                    break;
                }
            }
        }
        st.ped_n_waypoints[pq] = nw;
        // ---- leg odometry, then the pedestrian's obs yaw (env.py:683-693)
        double dist[3] = {st.ped_dist[pq * 3], st.ped_dist[pq * 3 + 1], st.ped_dist[pq * 3 + 2]};
        nv::leg_odometry(pp, pvel, st.ped_prev_yaw[pq], dt, dist);
        st.ped_dist[pq * 3] = dist[0]; st.ped_dist[pq * 3 + 1] = dist[1]; st.ped_dist[pq * 3 + 2] = dist[2];
        st.ped_prev_yaw[pq] = nv::wrap_pi(pp[2]);
        st.ped_pose[pq * 3] = pp[0]; st.ped_pose[pq * 3 + 1] = pp[1]; st.ped_pose[pq * 3 + 2] = pp[2];
        st.ped_vel[pq * 2] = pvel[0]; st.ped_vel[pq * 2 + 1] = pvel[1];
    }
}

// The pedestrians of every arena, one wavefront per arena, launched ahead of the fused step.  Inside the step
// this phase is a third of a workgroup's lifetime during which three of its four wavefronts only hold their
// slots; here an arena costs one wavefront.  Same device function, same results.
template <typename Field>
__global__ __launch_bounds__(64) void ped_update_kernel(navsim_config c, navsim_state st) {
    extern __shared__ __attribute__((aligned(16))) char ped_dyn[];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int N = c.max_peds;
    int n = st.n_peds[e];
    n = n > N ? N : n;
    if (n <= 0) return;
    const unsigned pair_bytes = (unsigned)(((size_t)N * (N + 1) * sizeof(double2) + 15) & ~(size_t)15);
    const PedShared ps = ped_lds_carve(ped_dyn + pair_bytes, N);
    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, c.map_h, c.map_w);
    const double old_rp[3] = {st.robot_pose[3 * (size_t)e], st.robot_pose[3 * (size_t)e + 1], st.robot_pose[3 * (size_t)e + 2]};
    const double prev_v = st.prev_action[2 * (size_t)e];
    const size_t pq = (size_t)e * N + (tid < n ? tid : 0);
    const bool is_ped = tid < n;
    double pp[3] = {0.0, 0.0, 0.0}, pvel[2] = {0.0, 0.0};
    if (is_ped) {
        pp[0] = st.ped_pose[pq * 3]; pp[1] = st.ped_pose[pq * 3 + 1]; pp[2] = st.ped_pose[pq * 3 + 2];
        pvel[0] = st.ped_vel[pq * 2]; pvel[1] = st.ped_vel[pq * 2 + 1];
    }
    // the step increments steps[e] before anything else (env.py:592); it has not run yet
    ped_phase<64, Field>(c, st, field, e, n, tid, is_ped, pq, c.time_step, (uint64_t)(c.env_index_base + e),
                         (uint64_t)st.steps[e] + 1, old_rp, prev_v, ps, ped_dyn, pair_bytes, pp, pvel);
}

template <int BLOCK, int R, bool PEDS, typename Field, int MODE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void navsim_step_kernel(navsim_config c, navsim_state st,
                                                            navsim_step_io io, int reset_only,
                                                            const uint8_t* __restrict__ reset_mask,
                                                            char* __restrict__ ws_env, char* __restrict__ ws_prims,
                                                            float* __restrict__ ws_ranges, unsigned dyn_lds_bytes,
                                                            unsigned tile_lds_bytes) {
    __shared__ StepShared sh;
    const int peds_done = (reset_only >> 1) & 1;     // the pedestrians were advanced by ped_update_kernel
    reset_only &= 1;
    // dynamic LDS: [analytic tile table of the arena, tile_lds_bytes][float2 dir[B], float rng[B]]
    extern __shared__ __attribute__((aligned(16))) char dyn_lds_all[];
    char* dyn_lds = dyn_lds_all + tile_lds_bytes;
    const uint32_t* tiles_lds = tile_lds_bytes ? (const uint32_t*)dyn_lds_all : nullptr;
    PedShared ps = {};
    if constexpr (PEDS) ps = ped_lds_carve(dyn_lds + ((dyn_lds_bytes + 15u) & ~15u), c.max_peds);
    const Prims prims = {ps.seg, ps.disc, ps.info};
    // longest-first launch order (a scheduling hint: which arena a workgroup takes never changes a result)
    const int e = (MODE == kModeFused && st.launch_order) ? st.launch_order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x;
    unsigned long long t_begin = 0;
    if (MODE == kModeFused && st.arena_cost && tid == 0) t_begin = __builtin_amdgcn_s_memrealtime();
    if (tile_lds_bytes) {                                       // stage the arena's tile table (coalesced)
        const uint4* src = (const uint4*)((const char*)st.tile_table + (size_t)(c.shared_field ? 0 : e) * tile_lds_bytes);
        uint4* dst = (uint4*)dyn_lds_all;
        for (int i = tid; i < (int)(tile_lds_bytes / 16); i += BLOCK) dst[i] = src[i];
        // visibility: every path reaches a __syncthreads() before the first scan
    }
    const int B = c.n_beams, S = c.n_scan_stack, N = c.max_peds, D = S * B + 7;
    const double dt = c.time_step;
    const uint64_t genv = (uint64_t)(c.env_index_base + e);
    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, c.map_h, c.map_w);
    float* obs_row = io.obs + (size_t)e * D;
    const float* obs_prev = io.obs_prev ? io.obs_prev + (size_t)e * D : nullptr;
    double* rp_g = st.robot_pose + 3 * (size_t)e;
    double* goal_g = st.robot_goal + 2 * (size_t)e;
    double* pa_g = st.prev_action + 2 * (size_t)e;
    double* pv_g = st.prev_pose + 3 * (size_t)e;

    if (reset_only && reset_mask && !reset_mask[e]) {          // untouched env: carry the row over
        if ((MODE == kModeFused || MODE == kModePre) && obs_prev)
            for (int k = tid; k < D; k += BLOCK) obs_row[k] = obs_prev[k];
        return;
    }
    StepShared* slot = (MODE == kModeFused) ? nullptr : (StepShared*)(ws_env + (size_t)e * kPoolEnvBytes);
    constexpr int kPrimWords = (int)((sizeof(float) * 4 * 4 * NAVSIM_MAX_PEDS + sizeof(float) * 2 * 2 * NAVSIM_MAX_PEDS) / 4);
    if constexpr (MODE == kModePost || MODE == kModeFinal) {
        if (MODE == kModeFinal && !slot->rescan) return;        // nothing was re-scanned for this arena
        for (int i = tid; i < (int)(sizeof(StepShared) / 4); i += BLOCK) ((int*)&sh)[i] = ((const int*)slot)[i];
        if constexpr (PEDS) {
            const int* src = (const int*)(ws_prims + (size_t)e * kPrimWords * 4);
            for (int i = tid; i < 20 * c.max_peds; i += BLOCK) ((int*)ps.seg)[i] = src[i];   // seg, then disc
        }
        __syncthreads();
    }

    int n = (!PEDS || c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    const float noise_std = (c.add_scan_noise && st.scan_noise_std) ? st.scan_noise_std[e] : 0.0f;

    NAVSIM_STAMP(0);
    if constexpr (MODE == kModeFused || MODE == kModePre) {
    // ---------------------------------------------------------------- phase 0: scalars
    if (tid == 0) {
        sh.old_rp[0] = rp_g[0]; sh.old_rp[1] = rp_g[1]; sh.old_rp[2] = rp_g[2];
        sh.nseg = 0; sh.ndisc = 0; sh.rescan = 0; sh.respawn = 0;
        if (!reset_only) {
            double a0 = io.action[2 * e], a1 = io.action[2 * e + 1];
            st.steps[e] += 1;                                  // env.py:592
            if (c.min_turning_radius > 0.0) {                  // env.py:595-600
                double lim = fabs(a1) * c.min_turning_radius;
                if (a0 >= 0.0) a0 = (a0 > lim) ? a0 : lim;
                else           a0 = (a0 < -lim) ? a0 : -lim;
            }
            sh.act[0] = a0; sh.act[1] = a1;
        }
    }
    __syncthreads();

    NAVSIM_STAMP(1);
    // ---------------------------------------------------------------- phase 1: pedestrians
    double pp[3] = {0.0, 0.0, 0.0};
    double pvel[2] = {0.0, 0.0};
    const size_t pq = (size_t)e * N + (tid < n ? tid : 0);
    const bool is_ped = PEDS && tid < n;
    if (is_ped) {
        pp[0] = st.ped_pose[pq * 3]; pp[1] = st.ped_pose[pq * 3 + 1]; pp[2] = st.ped_pose[pq * 3 + 2];
        pvel[0] = st.ped_vel[pq * 2]; pvel[1] = st.ped_vel[pq * 2 + 1];
    }
    if (!reset_only && !PEDS) {
        if (tid == 0) {                                         // env.py:664
            double p[3] = {sh.old_rp[0], sh.old_rp[1], sh.old_rp[2]};
            nv::set_vel(p, sh.act[0], sh.act[1], dt, c.axle_offset, nullptr);
            sh.rp[0] = p[0]; sh.rp[1] = p[1]; sh.rp[2] = p[2];
        }
    } else if (!reset_only) {
        if (!peds_done)
            ped_phase<BLOCK, Field>(c, st, field, e, n, tid, is_ped, pq, dt, genv, (uint64_t)st.steps[e], sh.old_rp, pa_g[0],
                                    ps, dyn_lds, dyn_lds_bytes, pp, pvel);
        // ---- robot (env.py:664)
        if (tid == 0) {
            double p[3] = {sh.old_rp[0], sh.old_rp[1], sh.old_rp[2]};
            nv::set_vel(p, sh.act[0], sh.act[1], dt, c.axle_offset, nullptr);
            sh.rp[0] = p[0]; sh.rp[1] = p[1]; sh.rp[2] = p[2];
        }
    } else {
        if (tid == 0) { sh.rp[0] = sh.old_rp[0]; sh.rp[1] = sh.old_rp[1]; sh.rp[2] = sh.old_rp[2]; }
        if (is_ped) {                                           // env.py:809, 812-820
            st.ped_dist[pq * 3] = 0.0; st.ped_dist[pq * 3 + 1] = 0.0; st.ped_dist[pq * 3 + 2] = 0.0;
            st.ped_prev_yaw[pq] = nv::wrap_pi(pp[2]);
        }
    }

    // ---------------------------------------------------------------- phase 2: what the lidar sees
    if (is_ped) {                                               // env.py:392-414
        float dist3[3];
        if (reset_only) { dist3[0] = dist3[1] = dist3[2] = 0.0f; }
        else { dist3[0] = (float)st.ped_dist[pq * 3]; dist3[1] = (float)st.ped_dist[pq * 3 + 1];
               dist3[2] = (float)st.ped_dist[pq * 3 + 2]; }
        if (st.ped_has_legs[pq] && c.lidar_legs) {
            float cc[4];
            nv::leg_centres((float)pp[0], (float)pp[1], (float)pp[2], dist3[0], dist3[1], dist3[2], cc);
            int q = atomicAdd(&sh.ndisc, 2);
            ps.disc[q][0] = cc[0]; ps.disc[q][1] = cc[1];
            ps.disc[q + 1][0] = cc[2]; ps.disc[q + 1][1] = cc[3];
        } else {
            const double fpx[4] = {0.22, -0.22, -0.22, 0.22};   // human.py:5-10
            const double fpy[4] = {0.19, 0.19, -0.19, -0.19};
            double s, cs;
            nv::sincos(pp[2], s, cs);
            float vx[4], vy[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                vx[v] = (float)((cs * fpx[v] - s * fpy[v]) + pp[0]);
                vy[v] = (float)((s * fpx[v] + cs * fpy[v]) + pp[1]);
            }
            int q = atomicAdd(&sh.nseg, 4);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int w = (v + 1) & 3;
                ps.seg[q + v][0] = vx[v]; ps.seg[q + v][1] = vy[v];
                ps.seg[q + v][2] = vx[w]; ps.seg[q + v][3] = vy[w];
            }
        }
    }
    if (tid == 0) {
        sh.lx = (float)sh.rp[0]; sh.ly = (float)sh.rp[1]; sh.lth = (float)sh.rp[2];   // env.py:386
        nv::xy_to_ij_f32(sh.lx, sh.ly, c, sh.i0, sh.j0);                              // env.py:419
        nv::sincos((double)sh.lth, sh.sT, sh.cT);
        first_probe(field, sh.i0, sh.j0, (float)((long long)c.map_h * c.map_w), sh.t1, sh.r_all);
        sh.step_key = (unsigned long long)st.episode[e] * 0x100000000ULL +
                      (unsigned long long)(reset_only ? 0 : st.steps[e]) * 2ULL;
    }
    __syncthreads();
    if constexpr (MODE == kModePre) {                           // park the arena, the pool marches next
        for (int i = tid; i < (int)(sizeof(StepShared) / 4); i += BLOCK) ((int*)slot)[i] = ((const int*)&sh)[i];
        if constexpr (PEDS) {
            int* dst = (int*)(ws_prims + (size_t)e * kPrimWords * 4);
            for (int i = tid; i < 20 * c.max_peds; i += BLOCK) dst[i] = ((const int*)ps.seg)[i];
        }
        return;
    }
    }   // MODE fused / pre

    NAVSIM_STAMP(2);
    // ---------------------------------------------------------------- phase 3: scan A
    int n_hist = reset_only ? 0 : st.n_hist[e];
    if (MODE == kModeFinal && sh.respawn) n_hist = 0;
    int crash = 0, discomfort = 0;
    const uint64_t step_key = sh.step_key + (MODE == kModeFinal ? 1 : 0);
    float2* dir_lds = (float2*)dyn_lds;
    float* rng_lds = (float*)(dyn_lds + sizeof(float2) * (size_t)B);
    if constexpr (MODE != kModeFused)
        finish_beams<BLOCK>(c, sh, prims, st.beam_table, nullptr, ws_ranges + (size_t)e * B, nullptr, st.scan_threshold,
                            st.scan_discomfort, obs_row, n_hist, noise_std, step_key, genv, crash, discomfort);
    else if constexpr (R == 11)
        scan_beams_pred<BLOCK, Field>(c, sh, field, st.beam_table, prims, st.scan_threshold, st.scan_discomfort,
                                      obs_row, n_hist, noise_std, step_key, genv, crash, discomfort);
    else if constexpr (R == 0)
        scan_beams_dyn<BLOCK, Field>(c, sh, field, st.beam_table, prims, dir_lds, rng_lds, st.scan_threshold,
                                     st.scan_discomfort, obs_row, n_hist, noise_std, step_key, genv, crash, discomfort);
    else
        scan_beams<BLOCK, R, Field, PEDS>(c, sh, field, st.beam_table, prims, dir_lds, rng_lds, tiles_lds, st.scan_threshold,
                                          st.scan_discomfort, obs_row, n_hist, noise_std, step_key, genv, crash, discomfort);

    NAVSIM_STAMP(3);
    if (!reset_only && MODE != kModeFinal) {
        crash = __syncthreads_or(crash);
        discomfort = __syncthreads_or(discomfort);
        double rmin = 1.0e300;
        if (discomfort && !crash) {                             // env.py:563-569
            for (int k = tid; k < B; k += BLOCK)
            {
                double ratio = nv::discomfort_ratio((double)obs_row[(size_t)(S - 1) * B + k],
                                                    st.scan_threshold[k], st.scan_discomfort[k]);
                rmin = ratio < rmin ? ratio : rmin;
            }
            rmin = wave_min_f64(rmin);
            if ((tid & 63) == 0) sh.wave_ratio[tid >> 6] = rmin;
            __syncthreads();
        }
        NAVSIM_STAMP(4);
        // ------------------------------------------------------------ phase 4: reward / done / info
        if (tid == 0) {
            if (discomfort && !crash)
                for (int w = 1; w < (BLOCK + 63) / 64; ++w) rmin = sh.wave_ratio[w] < rmin ? sh.wave_ratio[w] : rmin;
            double prev_xy[2] = {pv_g[0], pv_g[1]};
            double pose[2] = {sh.rp[0], sh.rp[1]};
            double vel[2] = {pa_g[0], pa_g[1]};                 // env.py:453: the PREVIOUS action
            double goal[2] = {goal_g[0], goal_g[1]};
            nv::RewardOut o = nv::reward_scalar(c, prev_xy, pose, vel, goal, crash != 0, discomfort != 0, rmin);
            io.reward[e] = o.reward;
            io.done[e] = (uint8_t)o.done;
            io.is_success[e] = o.success;
            io.is_crash[e] = o.crash;
            io.distance[e] = o.distance;
            if (o.done && c.auto_reset && c.n_spawn > 0) {      // build-defined respawn
                uint64_t h = nv::hash4(c.seed, genv, (uint64_t)st.episode[e], 0x5eedULL);
                int idx = (int)(h % (uint64_t)c.n_spawn);
                const double* sp = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
                const double* sg = st.spawn_goal + ((size_t)e * c.n_spawn + idx) * 2;
                sh.rp[0] = sp[0]; sh.rp[1] = sp[1]; sh.rp[2] = sp[2];
                goal_g[0] = sg[0]; goal_g[1] = sg[1];
                st.episode[e] += 1;
                st.steps[e] = 0;
                sh.respawn = 1; sh.rescan = 1;
            } else if (o.crash != 0.0f) {                       // env.py:707-717
                sh.rp[0] = pv_g[0]; sh.rp[1] = pv_g[1]; sh.rp[2] = pv_g[2];
                sh.rescan = 1;
            }
            if (sh.rescan) {
                sh.lx = (float)sh.rp[0]; sh.ly = (float)sh.rp[1]; sh.lth = (float)sh.rp[2];
                nv::xy_to_ij_f32(sh.lx, sh.ly, c, sh.i0, sh.j0);
                nv::sincos((double)sh.lth, sh.sT, sh.cT);
                first_probe(field, sh.i0, sh.j0, (float)((long long)c.map_h * c.map_w), sh.t1, sh.r_all);
            }
        }
        __syncthreads();
        if constexpr (MODE == kModePost) {
            if (sh.rescan) {                                    // hand the arena back to the pool
                for (int i = tid; i < (int)(sizeof(StepShared) / 4); i += BLOCK) ((int*)slot)[i] = ((const int*)&sh)[i];
                return;
            }
        }
        // ------------------------------------------------------------ phase 5: scan B (env.py:718-723)
        if (MODE == kModeFused && sh.rescan) {
            if (sh.respawn) n_hist = 0;
            int c2, d2;
            if constexpr (R == 11)
                scan_beams_pred<BLOCK, Field>(c, sh, field, st.beam_table, prims, st.scan_threshold, st.scan_discomfort,
                                              obs_row, n_hist, noise_std, step_key + 1, genv, c2, d2);
            else if constexpr (R == 0)
                scan_beams_dyn<BLOCK, Field>(c, sh, field, st.beam_table, prims, dir_lds, rng_lds, st.scan_threshold,
                                             st.scan_discomfort, obs_row, n_hist, noise_std, step_key + 1, genv, c2, d2);
            else
                scan_beams<BLOCK, R, Field, PEDS>(c, sh, field, st.beam_table, prims, dir_lds, rng_lds, tiles_lds, st.scan_threshold,
                                                  st.scan_discomfort, obs_row, n_hist, noise_std, step_key + 1, genv, c2, d2);
        }
    }

    NAVSIM_STAMP(5);
    // ---------------------------------------------------------------- phase 6: pack the observation
    const bool fresh = reset_only || sh.respawn;                // first obs of an episode
    if (!fresh && obs_prev) {                                   // env.py:267-274: shift the stack
        for (int j = 0; j < S - 1; ++j)
            if (S - 1 - j <= n_hist)
                for (int k = tid; k < B; k += BLOCK) obs_row[(size_t)j * B + k] = obs_prev[(size_t)(j + 1) * B + k];
    }
    if (tid == 0) {
        float* tail = obs_row + (size_t)S * B;
        double yaw = nv::wrap_pi(sh.rp[2]);                     // env.py:454
        double pxy0 = fresh ? sh.rp[0] : pv_g[0];               // env.py:449-452
        double pxy1 = fresh ? sh.rp[1] : pv_g[1];
        double v0 = fresh ? 0.0 : pa_g[0], v1 = fresh ? 0.0 : pa_g[1];
        tail[0] = (float)pxy0; tail[1] = (float)pxy1;
        tail[2] = (float)sh.rp[0]; tail[3] = (float)sh.rp[1];
        tail[4] = (float)v0; tail[5] = (float)v1;
        tail[6] = (float)yaw;
        if (io.achieved_goal) { io.achieved_goal[2 * e] = (float)sh.rp[0]; io.achieved_goal[2 * e + 1] = (float)sh.rp[1]; }
        if (io.desired_goal) { io.desired_goal[2 * e] = (float)goal_g[0]; io.desired_goal[2 * e + 1] = (float)goal_g[1]; }
        // state for the next step (env.py:725-727)
        rp_g[0] = sh.rp[0]; rp_g[1] = sh.rp[1]; rp_g[2] = sh.rp[2];
        if (fresh) { pa_g[0] = 0.0; pa_g[1] = 0.0; st.n_hist[e] = (S - 1 < 1) ? S - 1 : 1; }
        else       { pa_g[0] = sh.act[0]; pa_g[1] = sh.act[1]; st.n_hist[e] = (n_hist + 1 < S - 1) ? n_hist + 1 : S - 1; }
        if (reset_only) st.steps[e] = 0;
        pv_g[0] = sh.rp[0]; pv_g[1] = sh.rp[1]; pv_g[2] = yaw;
        if (MODE == kModeFused && st.arena_cost && !reset_only)
            st.arena_cost[e] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_begin);
    }
    NAVSIM_STAMP(6);
}

// navsim_launch_order: arenas by descending cost.  One workgroup: maximum, 1024-bucket histogram on the cost
// scaled to the maximum, exclusive scan from the expensive end, scatter.  Order inside a bucket is free.
__global__ __launch_bounds__(1024) void launch_order_kernel(const uint32_t* __restrict__ cost, int32_t* __restrict__ order,
                                                            int n) {
    __shared__ unsigned hist[1024], base[1024];
    __shared__ unsigned max_s;
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) max_s = 1;
    __syncthreads();
    unsigned mx = 0;
    for (int e = tid; e < n; e += 1024) mx = cost[e] > mx ? cost[e] : mx;
    atomicMax(&max_s, mx);
    __syncthreads();
    const unsigned long long m = max_s;
    auto bucket = [&](unsigned cst) { return 1023 - (int)(((unsigned long long)cst * 1023ull) / m); };   // 0 = costliest
    for (int e = tid; e < n; e += 1024) atomicAdd(&hist[bucket(cost[e])], 1u);
    __syncthreads();
    base[tid] = hist[tid];
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                 // inclusive scan
        unsigned v = (tid >= off) ? base[tid - off] : 0;
        __syncthreads();
        base[tid] += v;
        __syncthreads();
    }
    base[tid] -= hist[tid];                                    // exclusive
    __syncthreads();
    for (int e = tid; e < n; e += 1024) order[atomicAdd(&base[bucket(cost[e])], 1u)] = e;
}

// ============================================================================================
// navsim_regen: reset() of finished arenas on the device with a new random map.
// Specification: oracle/navsim_ref.c navsim_regen_cpu (same hash-keyed uniforms, same tries).
// ============================================================================================
__device__ __forceinline__ double rg_u(uint64_t key, uint64_t i) {
    return (double)(nv::mix64(key + i * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
}

// ordered compaction of the arenas that finished in this step: list[0..count), mask[e]
__global__ __launch_bounds__(1024) void regen_select_kernel(const uint8_t* __restrict__ done, int E, int cap,
                                                            int* __restrict__ count, int* __restrict__ list,
                                                            uint8_t* __restrict__ mask) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (E + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < E) ? lo + per : E;
    int n = 0;
    for (int e = lo; e < hi; ++e) n += done[e] != 0;
    part[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                 // inclusive scan
        int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int pos = part[tid] - n;
    for (int e = lo; e < hi; ++e) {
        bool take = done[e] != 0 && pos < cap;
        mask[e] = take ? 1 : 0;
        if (take) list[pos] = e;
        pos += done[e] != 0;
    }
    if (tid == 1023) *count = part[1023] < cap ? part[1023] : cap;
}

// create_indoor_map (map_generator.py:97-123; oracle regen_map_indoor): the corridor tree on the coarse grid,
// one workgroup per regenerated arena.  The tree grows one node per iteration (nearest node by a workgroup
// min-reduction on (L1 distance, node index), then the two corridor rectangles carved by all threads);
// the grid lives in LDS and is written to grid_all[b] (G*G bytes, stride 100*100).  kind[b] = G for a
// corridor map, 0 for an outdoor one (regen_maps_kernel then draws the outdoor map as before).
__global__ __launch_bounds__(256) void regen_indoor_kernel(navsim_config c, navsim_state st,
                                                           const int* __restrict__ count, const int* __restrict__ list,
                                                           uint8_t* __restrict__ grid_all, int* __restrict__ kind) {
    __shared__ uint8_t g[100 * 100];
    __shared__ int tx[152], ty[152];
    __shared__ unsigned best_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= *count) return;
    const int e = list[b], size = c.map_w;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    const bool indoor = c.regen_indoor_ratio > 0.0 && rg_u(nv::hash4(c.seed, genv, ep, 0x4B494E44ULL), 0) < c.regen_indoor_ratio;
    if (!indoor) { if (tid == 0) kind[b] = 0; return; }                  // block-uniform
    const uint64_t key = nv::hash4(c.seed, genv, ep, 0x494E44ULL);
    uint64_t n = 0;
    const int r = 3 + (int)(rg_u(key, n++) * 2.0);
    const int it = 80 + (int)(rg_u(key, n++) * 71.0);
    int G = size / 10;
    G = G < 2 * r + 8 ? 2 * r + 8 : G;
    G = G > 100 ? 100 : G;
    int n_it = (it * G * G + 5000) / 10000;
    n_it = n_it < 4 ? 4 : (n_it > 150 ? 150 : n_it);
    for (int k = tid; k < G * G; k += 256) g[k] = 1;
    if (tid == 0) { tx[0] = G / 2; ty[0] = G / 2; }
    __syncthreads();
    if (tid == 0) g[(G / 2) * G + G / 2] = 0;
    const int span = G - 2 * r - 3;
    for (int k = 0; k < n_it; ++k) {
        const int px = r + 2 + (int)(rg_u(key, n) * span), py = r + 2 + (int)(rg_u(key, n + 1) * span);
        const bool coin = rg_u(key, n + 2) >= 0.5;
        n += 3;
        const int nt = k + 1;
        if (tid == 0) best_s = 0xFFFFFFFFu;
        __syncthreads();
        if (tid < nt) atomicMin(&best_s, ((unsigned)(abs(px - tx[tid]) + abs(py - ty[tid])) << 8) | (unsigned)tid);
        __syncthreads();
        const int best = (int)(best_s & 0xFFu);
        const int qx = tx[best], qy = ty[best];
        const int x1 = px < qx ? px : qx, x2 = px < qx ? qx : px;
        const int y1 = py < qy ? py : qy, y2 = py < qy ? qy : py;
        const bool constellation1 = (px > qx && py < qy) || (px < qx && py > qy);
        const int hx = coin ? x1 : x2;
        const int cy = coin ? (constellation1 ? y1 : y2) : (constellation1 ? y2 : y1);
        const int wh = y2 - y1 + 2 * r + 1, hv = x2 - x1 + 2 * r + 1, side = 2 * r + 1;
        for (int idx = tid; idx < side * wh; idx += 256) {
            int a = hx - r + idx / wh, bq = y1 - r + idx % wh;
            if (a >= 0 && a < G && bq >= 0 && bq < G) g[a * G + bq] = 0;
        }
        for (int idx = tid; idx < hv * side; idx += 256) {
            int a = x1 - r + idx / side, bq = cy - r + idx % side;
            if (a >= 0 && a < G && bq >= 0 && bq < G) g[a * G + bq] = 0;
        }
        if (tid == 0) { tx[nt] = px; ty[nt] = py; g[px * G + py] = 0; }
        __syncthreads();
    }
    uint8_t* out = grid_all + (size_t)b * 10000;
    for (int k = tid; k < G * G; k += 256) out[k] = g[k];
    if (tid == 0) kind[b] = G;
}

// create_outdoor_map (map_generator.py:126-143) at size x size, hash-keyed: kRegenSlices workgroups per map,
// each filling its own band of rows (border wall, four cells per store) and then the parts of the obstacle
// squares that fall into the band.
constexpr int kRegenSlices = 8;
__global__ __launch_bounds__(256) void regen_maps_kernel(navsim_config c, navsim_state st,
                                                         const int* __restrict__ count, const int* __restrict__ list,
                                                         uint8_t* __restrict__ occ_all,
                                                         const uint8_t* __restrict__ grid_all, const int* __restrict__ kind) {
    __shared__ int ocx[64], ocy[64];
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], size = c.map_w, tid = threadIdx.x;
    uint8_t* occ = occ_all + (size_t)b * size * size;
    if (const int G = kind[b]) {                                       // corridor map: nearest upscaling + flip
        const uint8_t* gsrc = grid_all + (size_t)b * 10000;
        const int rows_i = (size + kRegenSlices - 1) / kRegenSlices;
        const int ra = blockIdx.y * rows_i, rb = (ra + rows_i < size) ? ra + rows_i : size;
        for (int idx = ra * size + tid; idx < rb * size; idx += 256) {
            int yy = idx / size, xx = idx - yy * size;
            occ[(size_t)(size - 1 - yy) * size + xx] = gsrc[(int)(((long long)yy * G) / size) * G + (int)(((long long)xx * G) / size)];
        }
        return;
    }
    const uint64_t key = nv::hash4(c.seed, (uint64_t)(c.env_index_base + e), (uint64_t)st.episode[e], 0x4D4150ULL);
    double w = c.obstacle_width_lo + (c.obstacle_width_hi - c.obstacle_width_lo) * rg_u(key, 0);
    const int hw = (int)(10.0 * w);
    int span = size - 2 * hw - 3;
    span = span < 1 ? 1 : span;
    const int n_obs = c.obstacle_number < 64 ? c.obstacle_number : 64;
    if (tid < n_obs) {
        ocx[tid] = hw + 2 + (int)(rg_u(key, 1 + 2 * (uint64_t)tid) * span);
        ocy[tid] = hw + 2 + (int)(rg_u(key, 2 + 2 * (uint64_t)tid) * span);
    }
    __syncthreads();
    // this workgroup owns rows [r0, r1): background first, then the parts of the obstacle squares inside them
    const int rows = (size + kRegenSlices - 1) / kRegenSlices;
    const int r0 = blockIdx.y * rows, r1 = (r0 + rows < size) ? r0 + rows : size;
    if ((size & 3) == 0) {                                       // 4 cells per store
        const int wpr = size >> 2;
        uint32_t* occ32 = (uint32_t*)occ;
        for (int idx = r0 * wpr + tid; idx < r1 * wpr; idx += 256) {
            int r = idx / wpr, q4 = (idx - r * wpr) * 4;
            uint32_t wv = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int q = q4 + j;
                wv |= (uint32_t)(!(r >= 5 && r < size - 5 && q >= 5 && q < size - 5)) << (8 * j);
            }
            occ32[(size_t)(size - 1 - r) * wpr + (q4 >> 2)] = wv;
        }
    } else {
        for (int idx = r0 * size + tid; idx < r1 * size; idx += 256) {
            int r = idx / size, q = idx - r * size;
            occ[(size_t)(size - 1 - r) * size + q] = !(r >= 5 && r < size - 5 && q >= 5 && q < size - 5);
        }
    }
    __syncthreads();
    const int side = 2 * hw + 1;
    for (int o = 0; o < n_obs; ++o) {
        const int cx = ocx[o], cy = ocy[o];
        const int ra = (cx - hw > r0) ? cx - hw : r0, rb = (cx + hw < r1 - 1) ? cx + hw : r1 - 1;
        for (int idx = tid; idx < (rb - ra + 1) * side; idx += 256) {
            int r = ra + idx / side, q = cy - hw + idx % side;
            if (q >= 0 && q < size) occ[(size_t)(size - 1 - r) * size + q] = 1;
        }
    }
}

// install the new distance field of every regenerated arena (kRegenSlices workgroups per map, 16-byte copies)
__global__ __launch_bounds__(256) void regen_field_kernel(navsim_state st, const int* __restrict__ count,
                                                          const int* __restrict__ list,
                                                          const char* __restrict__ field_scratch, size_t field_bytes) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const char* src_b = field_scratch + (size_t)b * field_bytes;
    char* dst_b = (char*)st.field + (size_t)e * field_bytes;
    if (((field_bytes | (size_t)(uintptr_t)src_b | (size_t)(uintptr_t)dst_b) & 15) == 0) {
        const size_t n16 = field_bytes / 16;
        const size_t per = (n16 + kRegenSlices - 1) / kRegenSlices;
        const size_t lo = blockIdx.y * per, hi = (lo + per < n16) ? lo + per : n16;
        const uint4* src = (const uint4*)src_b;
        uint4* dst = (uint4*)dst_b;
        for (size_t i = lo + tid; i < hi; i += 256) dst[i] = src[i];
    } else {                                              // odd map sizes: every field format is 2-byte granular
        const size_t n2 = field_bytes / 2;
        const size_t per = (n2 + kRegenSlices - 1) / kRegenSlices;
        const size_t lo = blockIdx.y * per, hi = (lo + per < n2) ? lo + per : n2;
        for (size_t i = lo + tid; i < hi; i += 256) ((uint16_t*)dst_b)[i] = ((const uint16_t*)src_b)[i];
    }
}

template <typename Field>
__device__ __forceinline__ void rg_sample(const navsim_config& c, const Field& f, uint64_t key, uint64_t& n,
                                          double clr, bool use_ref, double rx, double ry, double dmin, double dmax,
                                          double& x, double& y) {
    const int W = c.map_w, H = c.map_h;
    int bi = 0, bj = 0;
    float bd = -1.0f;
    for (int t = 0; t < 64; ++t) {
        int i = (int)(rg_u(key, n++) * W), j = (int)(rg_u(key, n++) * H);
        float d = f.at(i, j);
        double px = ((double)i + 0.5) * c.resolution + c.origin_x;
        double py = ((double)j + 0.5) * c.resolution + c.origin_y;
        bool ok = (double)d >= clr;
        if (ok && use_ref) {
            double ddx = px - rx, ddy = py - ry;
            double dist = sqrt(ddx * ddx + ddy * ddy);
            ok = dist > dmin && dist < dmax;
        }
        if (ok) { x = px; y = py; return; }
        if (d > bd) { bd = d; bi = i; bj = j; }
    }
    x = ((double)bi + 0.5) * c.resolution + c.origin_x;
    y = ((double)bj + 0.5) * c.resolution + c.origin_y;
}

// install the new field, draw the start / goal table, the robot and the pedestrians
template <typename Field>
__global__ __launch_bounds__(256) void regen_commit_kernel(navsim_config c, navsim_state st,
                                                           const int* __restrict__ count, const int* __restrict__ list,
                                                           const char* __restrict__ field_scratch, size_t field_bytes) {
    __shared__ double robot_xy[2];
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, K = c.n_spawn, P = NAVSIM_MAX_WAYPOINTS;
    const Field f(st.field, st.field_overflow, e, c.map_h, c.map_w);
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    double* sp = (double*)st.spawn_pose + (size_t)e * K * 3;
    double* sg = (double*)st.spawn_goal + (size_t)e * K * 2;
    const double clr = c.spawn_clearance / c.resolution;
    for (int k = tid; k < K; k += 256) {
        uint64_t key = nv::hash4(c.seed, genv, ep, 0x53504157ULL + (uint64_t)k), n = 0;
        double x, y, gx, gy;
        rg_sample(c, f, key, n, clr, false, 0, 0, 0, 0, x, y);
        double th = nv::kTwoPi * rg_u(key, n++);
        rg_sample(c, f, key, n, clr, true, x, y, c.min_goal_dist, c.max_goal_dist, gx, gy);
        sp[3 * k] = x; sp[3 * k + 1] = y; sp[3 * k + 2] = th;
        sg[2 * k] = gx; sg[2 * k + 1] = gy;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int idx = (int)(nv::hash4(c.seed, genv, ep, 0x5eedULL) % (uint64_t)K);
        double* rp = st.robot_pose + 3 * (size_t)e;
        rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
        st.robot_goal[2 * e] = sg[2 * idx]; st.robot_goal[2 * e + 1] = sg[2 * idx + 1];
        robot_xy[0] = rp[0]; robot_xy[1] = rp[1];
    }
    __syncthreads();
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    const double pclr = c.ped_clearance / c.resolution;
    for (int i = tid; i < n; i += 256) {
        size_t q = (size_t)e * N + i;
        uint64_t key = nv::hash4(c.seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
        double x, y, gx, gy;
        rg_sample(c, f, key, m, pclr, true, robot_xy[0], robot_xy[1], c.ped_min_robot_dist, 1.0e300, x, y);
        double th = nv::kTwoPi * rg_u(key, m++);
        rg_sample(c, f, key, m, pclr, true, x, y, c.ped_min_goal_dist, 1.0e300, gx, gy);
        st.ped_pose[q * 3] = x; st.ped_pose[q * 3 + 1] = y; st.ped_pose[q * 3 + 2] = th;
        st.ped_vel[q * 2] = 0.0; st.ped_vel[q * 2 + 1] = 0.0;
        ((double*)st.ped_v_pref)[q] = c.v_pref_lo + (c.v_pref_hi - c.v_pref_lo) * rg_u(key, m++);
        ((uint8_t*)st.ped_has_legs)[q] = rg_u(key, m++) < c.has_legs_ratio;
        double* wp = st.ped_waypoints + (q * P) * 2;
        wp[0] = gx; wp[1] = gy;
        st.ped_n_waypoints[q] = 1;
    }
}

// ============================================================================================
// reset path: costmap (env.py:312-332), shortest path (pyastar2d at env.py:343-354), waypoints
// (env.py:1261-1277).  Specification incl. the tie-break: oracle/navsim_ref.c.
// ============================================================================================
__device__ __forceinline__ int reflect101(int k, int n) {
    if (n == 1) return 0;
    while (k < 0 || k >= n) { if (k < 0) k = -k; if (k >= n) k = 2 * (n - 1) - k; }
    return k;
}

__global__ __launch_bounds__(256) void costmap_kernel(const uint8_t* __restrict__ occ, int H, int W,
                                                      uint8_t* __restrict__ cost, const int* __restrict__ n_live,
                                                      const int* __restrict__ out_index) {
    const int Hc = H / 5, Wc = W / 5;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;
    if (idx >= Hc * Wc) return;
    int J = idx / Wc, I = idx - J * Wc;
    const uint8_t* o = occ + m * (size_t)H * W;
    int any = 0;
    for (int dj = -4; dj <= 4; ++dj)
        for (int di = -4; di <= 4; ++di) {
            int jj = reflect101(J + dj, Hc), ii = reflect101(I + di, Wc);
            any |= o[(size_t)(jj * 5) * W + ii * 5];
        }
    cost[(out_index ? (size_t)out_index[m] : m) * (size_t)Hc * Wc + idx] = any ? 1 : 0;
}

// one workgroup per query: level-synchronous breadth-first distances from the goal in LDS (int16),
// stopped at the start's level; thread 0 then walks the path (+i, -i, +j, -j order) and cuts it
// into waypoints exactly like path_to_waypoints.
// One query, executed by the whole 256-thread workgroup.  c = this query's costmap, w = its waypoint
// row (max_wp x 2); n_wp / path_cells / path_len point at its slots.
//
// LDS: dist[n_cells] int16 (-2 blocked, -1 free and unreached, else hops from the goal) followed by
// queue[n_cells] uint16, the breadth-first queue (every cell enters once; a level is the slice [lo, hi)).
// A level costs O(frontier) and ONE barrier: each frontier lane claims its free unreached neighbours with
// a 32-bit LDS atomic AND on the word holding the int16 (see `claim`), and the winner appends the cell.
// Queue order is arbitrary; only `dist` feeds the path, so results are deterministic.
// Thread 0 then walks the path and cuts the waypoints on the fly (nothing is stored per path cell).
constexpr size_t kPlanLdsMax = 160 * 1024 - 256;       // LDS per CU minus the static variables
inline size_t plan_lds(int Hc, int Wc) { return (size_t)((Hc * Wc + 1) & ~1) * 4; }
inline bool plan_fits(int Hc, int Wc) { return (size_t)Hc * Wc <= 65535 && plan_lds(Hc, Wc) <= kPlanLdsMax; }

__device__ __forceinline__ void plan_query(const uint8_t* __restrict__ c, int Hc, int Wc, double res_c, double ox,
                                           double oy, double sx_, double sy_, double gx_, double gy_, double interval,
                                           int max_wp, double* __restrict__ w, int32_t* __restrict__ n_wp,
                                           int32_t* __restrict__ path_cells, double* __restrict__ path_len) {
    extern __shared__ int16_t dist[];
    __shared__ int cnt[3], reached;
    const int tid = threadIdx.x;
    const int n_cells = Hc * Wc;
    uint16_t* queue = (uint16_t*)(dist + ((n_cells + 1) & ~1));
    navsim_config cc = {};
    cc.origin_x = ox; cc.origin_y = oy; cc.resolution = res_c; cc.map_h = Hc; cc.map_w = Wc;
    int si, sj, gi, gj;
    nv::xy_to_ij(sx_, sy_, cc, si, sj);
    nv::xy_to_ij(gx_, gy_, cc, gi, gj);
    bool ok = si < Wc && sj < Hc && gi < Wc && gj < Hc;
    if (ok) ok = !c[(size_t)sj * Wc + si] && !c[(size_t)gj * Wc + gi];
    if (tid == 0) {
        *n_wp = 0;
        if (path_cells) *path_cells = 0;
        if (path_len) *path_len = 0.0;
    }
    if (!ok) return;                                     // uniform: depends on the query only
    const int s_cell = sj * Wc + si, g_cell = gj * Wc + gi;
    if ((((uintptr_t)c) & 3) == 0) {                                       // four cells per load
        const uint32_t* c4 = (const uint32_t*)c;
        uint2* d4 = (uint2*)dist;
        for (int k = tid; k < n_cells / 4; k += 256) {
            const uint32_t v = c4[k];
            uint2 o;
            o.x = ((v & 0xFFu) ? 0xFFFEu : 0xFFFFu) | (((v >> 8) & 0xFFu) ? 0xFFFE0000u : 0xFFFF0000u);
            o.y = (((v >> 16) & 0xFFu) ? 0xFFFEu : 0xFFFFu) | ((v >> 24) ? 0xFFFE0000u : 0xFFFF0000u);
            d4[k] = o;
        }
        for (int k = (n_cells & ~3) + tid; k < n_cells; k += 256) dist[k] = c[k] ? (int16_t)-2 : (int16_t)-1;
    } else {
        for (int k = tid; k < n_cells; k += 256) dist[k] = c[k] ? (int16_t)-2 : (int16_t)-1;
    }
    if ((n_cells & 1) && tid == 0) dist[n_cells] = -2;                    // pad half of the last 32-bit word
    __syncthreads();
    if (tid == 0) {
        dist[g_cell] = 0; queue[0] = (uint16_t)g_cell;
        cnt[0] = 0; cnt[1] = 0; cnt[2] = 0;
        reached = (s_cell == g_cell);
    }
    __syncthreads();
    uint32_t* words = (uint32_t*)dist;
    const uint16_t* half = (const uint16_t*)dist;
    int lo = 0, hi = 1;
    for (int level = 1; level < 32767; ++level) {
        if (reached || lo == hi) break;
        const int slot = level % 3;
        if (tid == 0) cnt[(level + 1) % 3] = 0;          // last read two barriers ago
        // claim a neighbour for this level: one atomic AND turns an unreached half-word (0xFFFF) into
        // `level` and leaves a half-word some other lane claimed in this level unchanged; the lane that
        // saw 0xFFFF come back owns the cell.  Reached and blocked cells are filtered by the plain read.
        // The four reads, then the four atomics, are issued together (independent LDS round trips).
        for (int f = lo + tid; f < hi; f += 256) {
            const int k = queue[f], j = k / Wc, i = k - j * Wc;
            const int m[4] = {k + 1, k - 1, k + Wc, k - Wc};
            const bool in[4] = {i + 1 < Wc, i > 0, j + 1 < Hc, j > 0};
            bool want[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) want[d] = in[d] && half[in[d] ? m[d] : k] == 0xFFFFu;
            uint32_t old[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {               // branch-free: a lane with nothing to claim ANDs all ones
                const uint32_t sh = (uint32_t)(m[d] & 1) * 16u;
                const uint32_t mask = want[d] ? (((uint32_t)level << sh) | (0xFFFFu << (16u - sh))) : 0xFFFFFFFFu;
                old[d] = atomicAnd(&words[(want[d] ? m[d] : k) >> 1], mask);
            }
            // queue slots: one atomic per wavefront (ballot prefix), not one per lane or per cell
            uint64_t won[4];
            int total = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                want[d] = want[d] && ((old[d] >> ((uint32_t)(m[d] & 1) * 16u)) & 0xFFFFu) == 0xFFFFu;
                won[d] = __ballot(want[d]);
                total += __popcll(won[d]);
            }
            if (total) {                                     // wave-uniform
                const uint64_t below = (1ull << (tid & 63)) - 1ull;
                int base = 0;
                if ((__ballot(1) & below) == 0) base = atomicAdd(&cnt[slot], total);     // first active lane
                base = hi + __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (want[d]) {
                        queue[base + __popcll(won[d] & below)] = (uint16_t)m[d];
                        if (m[d] == s_cell) reached = 1;
                    }
                    base += __popcll(won[d]);
                }
            }
        }
        __syncthreads();
        lo = hi;
        hi += cnt[slot];
    }
#ifdef NAVSIM_DIAG_NO_WALK
    return;
#endif
    if (tid != 0 || dist[s_cell] < 0) return;
    int n = 0, count = 0, ci = si, cj = sj;
    const double fx0 = ((double)si + 0.5) * res_c + ox, fy0 = ((double)sj + 0.5) * res_c + oy;
    double fx = fx0, fy = fy0;                           // env.py:1261-1277, cut while walking
    const double i2_hi = interval * interval * (1.0 + 1.0e-12), i2_lo = interval * interval * (1.0 - 1.0e-12);
    for (;;) {
        ++n;
        const double cx = ((double)ci + 0.5) * res_c + ox, cy = ((double)cj + 0.5) * res_c + oy;
        const double dx = fx - cx, dy = fy - cy;
        const int k = cj * Wc + ci;
        const int dcur = dist[k];
        // sqrt(d2) > interval, decided on d2 unless it sits within 1e-12 of interval^2 (sqrt is monotone
        // and correctly rounded, so the two tests agree outside that band)
        const double d2 = dx * dx + dy * dy;
        const bool far = (d2 > i2_hi) || (!(d2 < i2_lo) && sqrt(d2) > interval);
        if (far) {
            if (count < max_wp) { w[2 * count] = cx; w[2 * count + 1] = cy; }
            ++count; fx = cx; fy = cy;
        }
        if (dcur == 0) {                                 // the goal cell closes the list
            if (count < max_wp) { w[2 * count] = cx; w[2 * count + 1] = cy; }
            ++count;
            break;
        }
        const int want = dcur - 1;                       // first neighbour one hop closer, (+i, -i, +j, -j)
        const bool e0 = ci + 1 < Wc && dist[k + 1] == want, e1 = ci > 0 && dist[k - 1] == want;
        const bool e2 = cj + 1 < Hc && dist[k + Wc] == want;
        if (e0) ++ci; else if (e1) --ci; else if (e2) ++cj; else --cj;
    }
    int nw = count < max_wp ? count : max_wp;
    *n_wp = nw;
    if (path_cells) *path_cells = n;
    if (path_len) {
        double sx = sx_ - w[0], sy = sy_ - w[1];
        double L = sqrt(sx * sx + sy * sy);
        for (int k = 0; k + 1 < nw; ++k) {
            double ax = w[2 * k + 2] - w[2 * k], ay = w[2 * k + 3] - w[2 * k + 1];
            L += sqrt(ax * ax + ay * ay);
        }
        *path_len = L;
    }
}

__global__ __launch_bounds__(256) void plan_kernel(const uint8_t* __restrict__ cost, const int32_t* __restrict__ map_index,
                                                   int Hc, int Wc, double res_c, double ox, double oy,
                                                   const double* __restrict__ start, const double* __restrict__ goal,
                                                   double interval, int max_wp, double* __restrict__ wp,
                                                   int32_t* __restrict__ n_wp, int32_t* __restrict__ path_cells,
                                                   double* __restrict__ path_len) {
    const int q = blockIdx.x;
    plan_query(cost + (size_t)(map_index ? map_index[q] : q) * Hc * Wc, Hc, Wc, res_c, ox, oy, start[2 * q],
               start[2 * q + 1], goal[2 * q], goal[2 * q + 1], interval, max_wp, wp + (size_t)q * max_wp * 2, n_wp + q,
               path_cells ? path_cells + q : nullptr, path_len ? path_len + q : nullptr);
}

// --------------------------------------------------------------------------------------------
// navsim_regen with cfg.regen_plan = 1 (oracle/navsim_ref.c regen_planned): candidates on the costmap,
// a path must join start and goal.  Rounds of {sample, plan, accept} kernels; no host round trip.
// --------------------------------------------------------------------------------------------
struct RegenPlanWs {
    uint8_t* cost;        // [M, Hc, Wc] scratch, or the resident st.costmap (then indexed by arena)
    int cost_by_arena;
    double* qstart;       // [M, Q, 2]
    double* qgoal;        // [M, Q, 2]
    double* qwp;          // [M, Q, P, 2]   robot stage only (pedestrian paths go straight into the state)
    int32_t* qnwp;        // [M, Q]
    double* qlen;         // [M, Q]
    uint8_t* active;      // [M, Q]
    uint8_t* res_robot;   // [M, K]
    uint8_t* res_ped;     // [M, N]
    int Q;
};

__device__ __forceinline__ void rgp_cell(const navsim_config& c, const uint8_t* __restrict__ cost, int Hc, int Wc,
                                         double res_c, uint64_t key, uint64_t& n, bool use_ref, double rx, double ry,
                                         double dmin, double dmax, double& x, double& y) {
    for (int t = 0; t < 16; ++t) {
        int I = (int)(rg_u(key, n++) * Wc), J = (int)(rg_u(key, n++) * Hc);
        x = ((double)I + 0.5) * res_c + c.origin_x;
        y = ((double)J + 0.5) * res_c + c.origin_y;
        if (cost[(size_t)J * Wc + I]) continue;
        if (use_ref) {
            double ddx = x - rx, ddy = y - ry;
            double dist = sqrt(ddx * ddx + ddy * ddy);
            if (!(dist > dmin && dist < dmax)) continue;
        }
        return;
    }
}

// install the new field (same copy as regen_commit_kernel) and clear the per-slot flags
__global__ __launch_bounds__(256) void regen_install_kernel(navsim_config c, navsim_state st,
                                                            const int* __restrict__ count, const int* __restrict__ list,
                                                            const char* __restrict__ field_scratch, size_t field_bytes,
                                                            RegenPlanWs ws) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int tid = threadIdx.x;
    for (int k = tid; k < c.n_spawn; k += 256) ws.res_robot[(size_t)b * c.n_spawn + k] = 0;
    for (int i = tid; i < c.max_peds; i += 256) ws.res_ped[(size_t)b * c.max_peds + i] = 0;
}

// robot stage, one round: accept what the previous round planned, then draw a new candidate for every
// slot that is still open (round == 4: accept only, pick the robot, initialise the pedestrians)
__global__ __launch_bounds__(256) void regen_robot_round_kernel(navsim_config c, navsim_state st,
                                                                const int* __restrict__ count,
                                                                const int* __restrict__ list, RegenPlanWs ws, int round) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, K = c.n_spawn, Q = ws.Q;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = ws.cost + (size_t)(ws.cost_by_arena ? e : b) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    double* sp = (double*)st.spawn_pose + (size_t)e * K * 3;
    double* sg = (double*)st.spawn_goal + (size_t)e * K * 2;
    for (int k = tid; k < Q; k += 256) {
        const size_t q = (size_t)b * Q + k;
        if (k >= K) { ws.active[q] = 0; continue; }
        uint8_t& res = ws.res_robot[(size_t)b * K + k];
        if (round > 0 && !res) {
            double ddx = sg[2 * k] - sp[3 * k], ddy = sg[2 * k + 1] - sp[3 * k + 1];
            res = ws.qnwp[q] > 0 && ws.qlen[q] <= 2.0 * sqrt(ddx * ddx + ddy * ddy);      // env.py:761
        }
        ws.active[q] = 0;
        if (res || round >= 4) continue;
        uint64_t key = nv::hash4(c.seed, genv, ep, 0x52504C00ULL + (uint64_t)round * 256 + (uint64_t)k), n = 0;
        double x, y, gx, gy;
        rgp_cell(c, cost, Hc, Wc, res_c, key, n, false, 0, 0, 0, 0, x, y);
        rgp_cell(c, cost, Hc, Wc, res_c, key, n, true, x, y, c.min_goal_dist, c.max_goal_dist, gx, gy);
        sp[3 * k] = x; sp[3 * k + 1] = y; sp[3 * k + 2] = nv::kTwoPi * rg_u(key, n++);
        sg[2 * k] = gx; sg[2 * k + 1] = gy;
        ws.qstart[2 * q] = x; ws.qstart[2 * q + 1] = y;
        ws.qgoal[2 * q] = gx; ws.qgoal[2 * q + 1] = gy;
        ws.active[q] = 1;
    }
    if (round < 4) return;
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int idx = (int)(nv::hash4(c.seed, genv, ep, 0x5eedULL) % (uint64_t)K);
        const uint8_t* res = ws.res_robot + (size_t)b * K;
        if (!res[idx])
            for (int s_ = 1; s_ < K; ++s_) { int j = (idx + s_) % K; if (res[j]) { idx = j; break; } }
        double* rp = st.robot_pose + 3 * (size_t)e;
        rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
        st.robot_goal[2 * e] = sg[2 * idx]; st.robot_goal[2 * e + 1] = sg[2 * idx + 1];
    }
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    for (int i = tid; i < n; i += 256) {
        size_t q = (size_t)e * N + i;
        uint64_t k0 = nv::hash4(c.seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
        st.ped_pose[q * 3 + 2] = nv::kTwoPi * rg_u(k0, m++);
        ((double*)st.ped_v_pref)[q] = c.v_pref_lo + (c.v_pref_hi - c.v_pref_lo) * rg_u(k0, m++);
        ((uint8_t*)st.ped_has_legs)[q] = rg_u(k0, m++) < c.has_legs_ratio;
        st.ped_vel[q * 2] = 0.0; st.ped_vel[q * 2 + 1] = 0.0;
    }
}

// pedestrian stage, one round (round == 4: accept only)
__global__ __launch_bounds__(256) void regen_ped_round_kernel(navsim_config c, navsim_state st,
                                                              const int* __restrict__ count,
                                                              const int* __restrict__ list, RegenPlanWs ws, int round) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, Q = ws.Q, P = NAVSIM_MAX_WAYPOINTS;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = ws.cost + (size_t)(ws.cost_by_arena ? e : b) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    const double rx = st.robot_pose[3 * (size_t)e], ry = st.robot_pose[3 * (size_t)e + 1];
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    for (int i = tid; i < Q; i += 256) {
        const size_t q = (size_t)b * Q + i;
        if (i >= n) { ws.active[q] = 0; continue; }
        const size_t pq = (size_t)e * N + i;
        uint8_t& res = ws.res_ped[(size_t)b * N + i];
        if (round > 0 && !res && ws.qnwp[q] > 0) { st.ped_n_waypoints[pq] = ws.qnwp[q]; res = 1; }
        ws.active[q] = 0;
        if (res || round >= 4) continue;
        uint64_t key = nv::hash4(c.seed, genv, ep, 0x50504C00ULL + (uint64_t)round * 256 + (uint64_t)i), nn = 0;
        double x, y, gx, gy;
        rgp_cell(c, cost, Hc, Wc, res_c, key, nn, true, rx, ry, c.ped_min_robot_dist, 1.0e300, x, y);
        rgp_cell(c, cost, Hc, Wc, res_c, key, nn, true, x, y, c.ped_min_goal_dist, 1.0e300, gx, gy);
        st.ped_pose[pq * 3] = x; st.ped_pose[pq * 3 + 1] = y;
        double* w = st.ped_waypoints + (pq * P) * 2;
        w[0] = gx; w[1] = gy;
        st.ped_n_waypoints[pq] = 1;
        ws.qstart[2 * q] = x; ws.qstart[2 * q + 1] = y;
        ws.qgoal[2 * q] = gx; ws.qgoal[2 * q + 1] = gy;
        ws.active[q] = 1;
    }
}

// plan every active query of the round; ped_stage: waypoints go straight into st.ped_waypoints
__global__ __launch_bounds__(256) void regen_plan_kernel(navsim_config c, navsim_state st, const int* __restrict__ count,
                                                         const int* __restrict__ list, RegenPlanWs ws, int ped_stage) {
    const int q = blockIdx.x, b = q / ws.Q, k = q - b * ws.Q;
    if (b >= *count || !ws.active[q]) return;            // uniform per workgroup
    const int Hc = c.map_h / 5, Wc = c.map_w / 5, P = NAVSIM_MAX_WAYPOINTS;
    double* w = ped_stage ? st.ped_waypoints + (((size_t)list[b] * c.max_peds + k) * P) * 2
                          : ws.qwp + (size_t)q * P * 2;
    plan_query(ws.cost + (size_t)(ws.cost_by_arena ? list[b] : b) * Hc * Wc, Hc, Wc, c.resolution * 5.0, c.origin_x, c.origin_y, ws.qstart[2 * q],
               ws.qstart[2 * q + 1], ws.qgoal[2 * q], ws.qgoal[2 * q + 1], ped_stage ? 2.0 : 5.0, P, w, ws.qnwp + q,
               nullptr, ped_stage ? nullptr : ws.qlen + q);
}

// --------------------------------------------------------------------------------------------
// navsim_replan (env.py:667-680; oracle navsim_replan_cpu): ordered list of the pedestrians standing on
// their final waypoint, then one workgroup per listed pedestrian: draw a goal, plan, up to 4 rounds.
// --------------------------------------------------------------------------------------------
// one wavefront per arena: bit i of due[e] = pedestrian i stands within 0.5 m of its final waypoint
__global__ __launch_bounds__(64) void replan_flag_kernel(navsim_config c, navsim_state st, uint64_t* __restrict__ due) {
    const int e = blockIdx.x, i = threadIdx.x, N = c.max_peds, P = NAVSIM_MAX_WAYPOINTS;
    bool flag = false;
    if (i < N && i < st.n_peds[e]) {
        const size_t q = (size_t)e * N + i;
        const double* pp = st.ped_pose + q * 3;
        const double* w = st.ped_waypoints + (q * P) * 2;
        int nw = st.ped_n_waypoints[q];
        double ddx = pp[0] - w[2 * (nw - 1)], ddy = pp[1] - w[2 * (nw - 1) + 1];
        flag = sqrt(ddx * ddx + ddy * ddy) < 0.5;
    }
    uint64_t m = __ballot(flag);
    if (i == 0) due[e] = m;
}

// ordered compaction of the set bits, (arena, pedestrian) order, at most cap entries
__global__ __launch_bounds__(1024) void replan_select_kernel(const uint64_t* __restrict__ due, int E, int N, int cap,
                                                             int* __restrict__ count, int* __restrict__ list) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (E + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < E) ? lo + per : E;
    int n = 0;
    for (int e = lo; e < hi; ++e) n += __popcll(due[e]);
    part[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int pos = part[tid] - n;
    if (n)
        for (int e = lo; e < hi && pos < cap; ++e)
            for (uint64_t m = due[e]; m && pos < cap; m &= m - 1) list[pos++] = e * N + (__ffsll((unsigned long long)m) - 1);
    if (tid == 1023) *count = part[1023] < cap ? part[1023] : cap;
}

__global__ __launch_bounds__(256) void replan_kernel(navsim_config c, navsim_state st, const int* __restrict__ count,
                                                     const int* __restrict__ list) {
    __shared__ double goal_s[2];
    __shared__ int32_t nwp_s;
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int q = list[b], N = c.max_peds, P = NAVSIM_MAX_WAYPOINTS, tid = threadIdx.x;
    const int e = q / N, i = q - e * N;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = st.costmap + (size_t)(c.shared_field ? 0 : e) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e);
    const uint64_t when = (uint64_t)st.steps[e] + ((uint64_t)st.episode[e] << 40);
    const double px = st.ped_pose[(size_t)q * 3], py = st.ped_pose[(size_t)q * 3 + 1];
    double* w = st.ped_waypoints + ((size_t)q * P) * 2;
    for (int round = 0; round < 4; ++round) {
        if (tid == 0) {
            uint64_t key = nv::hash4(c.seed, genv, when, 0x52504E00ULL + (uint64_t)round * 256 + (uint64_t)i), m = 0;
            double gx, gy;
            rgp_cell(c, cost, Hc, Wc, res_c, key, m, true, px, py, c.ped_min_goal_dist, 1.0e300, gx, gy);
            goal_s[0] = gx; goal_s[1] = gy;
        }
        __syncthreads();
        plan_query(cost, Hc, Wc, res_c, c.origin_x, c.origin_y, px, py, goal_s[0], goal_s[1], 2.0, P, w, &nwp_s,
                   nullptr, nullptr);
        __syncthreads();
        if (nwp_s > 0) {
            if (tid == 0) st.ped_n_waypoints[q] = nwp_s;
            break;
        }
        __syncthreads();                                 // nwp_s is rewritten by the next round
    }
}

// ============================================================================================
// Pedestrian control block with the HumanPolicy actor (env.py:617-662, human_policy.py:19-52).
// Specification: oracle/navsim_ref.c policy_actor -- every dot product is a float32 fused-multiply-add
// chain in index order from 0, bias added last.  That is exactly what v_mfma_f32_32x32x2_f32 computes
// along k, so the 4096 -> 256 layer runs on the matrix cores and still equals the oracle bit for bit.
//   policy_features_kernel   one workgroup per pedestrian: clip / scale, conv1 + ReLU, conv2 + ReLU
//   policy_fc1_kernel        [P,4096] x [4096,256] on MFMA (128 x 128 tiles, LDS double buffer)
//   policy_head_kernel       waypoint pop, local goal, 260 -> 128, the two heads, clip, * v_pref
// ============================================================================================
constexpr int kPolFeat = 4096, kPolH1 = 256, kPolH2 = 128, kPolIn2 = 260;

__global__ __launch_bounds__(256) void policy_features_kernel(const float* __restrict__ scans, int p0, int n_ped,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ feat) {
    __shared__ float x[520];                 // x[1 + i] = input i, x[0] = left padding
    __shared__ float o1[32][258];            // o1[c][1 + t], zero padding at both ends
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    if (p >= n_ped) return;
    const float* scan = scans + (size_t)(p0 + p) * 512;
    for (int k = tid; k < 512; k += 256) {                      // env.py:629-630
        double v = (double)scan[k];
        v = v < 0.0 ? 0.0 : (v > 6.0 ? 6.0 : v);
        x[1 + k] = (float)(v / 6.0 - 0.5);
    }
    if (tid == 0) x[0] = 0.0f;
    if (tid < 32) { o1[tid][0] = 0.0f; o1[tid][256] = 0.0f; o1[tid][257] = 0.0f; }
    __syncthreads();
    if (tid < 255) {                                            // conv1: thread = output position
        float xv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) xv[k] = x[2 * tid + k];     // input index 2t + k - 1
        for (int o = 0; o < 32; ++o) {
            float acc = 0.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int k = 0; k < 5; ++k) acc = __builtin_fmaf(w1[(o * 3 + ch) * 5 + k], xv[k], acc);
            acc = acc + b1[o];
            o1[o][1 + tid] = acc > 0.0f ? acc : 0.0f;
        }
    }
    __syncthreads();
    {                                                           // conv2: thread = (position, half of the channels)
        const int t = tid & 127;
        const int og = __builtin_amdgcn_readfirstlane((tid >> 7) * 16);   // wave-uniform: weights come by s_load
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
        for (int c = 0; c < 32; ++c) {
            const float i0 = o1[c][2 * t], i1 = o1[c][2 * t + 1], i2 = o1[c][2 * t + 2];   // index 2t + k - 1
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float* ww = w2 + ((og + j) * 32 + c) * 3;         // wave-uniform: scalar loads
                acc[j] = __builtin_fmaf(ww[0], i0, acc[j]);
                acc[j] = __builtin_fmaf(ww[1], i1, acc[j]);
                acc[j] = __builtin_fmaf(ww[2], i2, acc[j]);
            }
        }
        float* f = feat + (size_t)p * kPolFeat;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = acc[j] + b2[og + j];
            f[(og + j) * 128 + t] = v > 0.0f ? v : 0.0f;
        }
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// h1[p][n] = relu(bias[n] + sum_k feat[p][k] * W[n][k]); one accumulator per output, k ascending
__global__ __launch_bounds__(256) void policy_fc1_kernel(const float* __restrict__ feat, int n_ped,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ h1) {
    constexpr int MT = 128, NT = 128, KT = 32, LD = KT + 1;     // +1: rows land in different banks
    extern __shared__ float lds_f[];                            // [2][MT*LD] A, then [2][NT*LD] B
    float* As = lds_f;
    float* Bs = lds_f + 2 * MT * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * MT, n0 = blockIdx.y * NT;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    // global -> register staging: 128 rows x 32 floats per operand = 1024 float4, 4 per thread
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + i * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            int m = m0 + row; m = m < n_ped ? m : n_ped - 1;
            ra[i] = *(const float4*)(feat + (size_t)m * kPolFeat + k0 + c4);
            rb[i] = *(const float4*)(W + (size_t)(n0 + row) * kPolFeat + k0 + c4);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + i * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            float* a = As + buf * MT * LD + row * LD + c4;
            float* b = Bs + buf * NT * LD + row * LD + c4;
            a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
            b[0] = rb[i].x; b[1] = rb[i].y; b[2] = rb[i].z; b[3] = rb[i].w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    fetch(0);
    stash(0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int k0 = 0, buf = 0; k0 < kPolFeat; k0 += KT, buf ^= 1) {
        const bool more = k0 + KT < kPolFeat;
        if (more) fetch(k0 + KT);
        const float* a_ = As + buf * MT * LD + (wm + lr) * LD + lk;
        const float* b_ = Bs + buf * NT * LD + (wn + lr) * LD + lk;
#pragma unroll
        for (int kk = 0; kk < KT; kk += 2) {
            const float a0 = a_[kk], a1 = a_[32 * LD + kk];
            const float b0 = b_[kk], b1 = b_[32 * LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + lr;
            const float bn = bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float v = acc[i][j][r] + bn;
                if (m < n_ped) h1[(size_t)m * kPolH1 + n] = v > 0.0f ? v : 0.0f;
            }
        }
}

// W2t[k][j] = W2[j][k]: coalesced rows for the head kernel
__global__ __launch_bounds__(256) void policy_transpose_kernel(const float* __restrict__ w2, float* __restrict__ w2t) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kPolH2 * kPolIn2) return;
    int j = idx / kPolIn2, k = idx - j * kPolIn2;
    w2t[k * kPolH2 + j] = w2[idx];
}

__global__ __launch_bounds__(128) void policy_head_kernel(navsim_config c, navsim_state st, int p0, int n_ped,
                                                          const float* __restrict__ h1, const float* __restrict__ w2t,
                                                          navsim_policy_weights w, float* __restrict__ prev_actions,
                                                          double* __restrict__ ped_cmd) {
    __shared__ float z[kPolIn2];
    __shared__ float h2[kPolH2];
    __shared__ float head[2];
    const int tid = threadIdx.x, N = c.max_peds, P = NAVSIM_MAX_WAYPOINTS;
    const int p = blockIdx.x;
    if (p >= n_ped) return;
    const size_t q = (size_t)(p0 + p);
    const int e = (int)(q / N), i = (int)(q - (size_t)e * N);
    int n = st.n_peds[e];
    n = n > N ? N : n;
    if (i >= n) {                                              // block-uniform
        if (tid == 0) { ped_cmd[2 * q] = 0.0; ped_cmd[2 * q + 1] = 0.0; }
        return;
    }
    z[tid] = h1[(size_t)p * kPolH1 + tid];
    z[tid + 128] = h1[(size_t)p * kPolH1 + tid + 128];
    if (tid == 0) {
        const double* pp = st.ped_pose + q * 3;
        double* wp = st.ped_waypoints + (q * P) * 2;
        int nw = st.ped_n_waypoints[q];
        while (nw > 1) {                                       // env.py:633-640
            double ddx = pp[0] - wp[0], ddy = pp[1] - wp[1];
            if (sqrt(ddx * ddx + ddy * ddy) < 1.0) {
                for (int k = 0; k + 1 < nw; ++k) { wp[2 * k] = wp[2 * k + 2]; wp[2 * k + 1] = wp[2 * k + 3]; }
                nw -= 1;
            } else break;
        }
        st.ped_n_waypoints[q] = nw;
        double s, cs;
        nv::sincos(pp[2], s, cs);                              // env.py:644-645
        double gx = wp[0] - pp[0], gy = wp[1] - pp[1];
        z[256] = (float)(gx * cs + gy * s);
        z[257] = (float)(-gx * s + gy * cs);
        z[258] = prev_actions[2 * q];
        z[259] = prev_actions[2 * q + 1];
    }
    __syncthreads();
    {
        float acc = 0.0f;
        for (int k = 0; k < kPolIn2; ++k) acc = __builtin_fmaf(w2t[k * kPolH2 + tid], z[k], acc);
        acc = acc + w.fc2_b[tid];
        h2[tid] = acc > 0.0f ? acc : 0.0f;
    }
    __syncthreads();
    if (tid == 0 || tid == 64) {                               // one head per wavefront
        const float* aw = tid == 0 ? w.a1_w : w.a2_w;
        float acc = 0.0f;
        for (int k = 0; k < kPolH2; ++k) acc = __builtin_fmaf(aw[k], h2[k], acc);
        head[tid >> 6] = acc + (tid == 0 ? w.a1_b[0] : w.a2_b[0]);
    }
    __syncthreads();
    if (tid == 0) {
        double x1 = (double)head[0], m1;                       // sigmoid, tanh in float64 on the shared exp
        if (x1 >= 0.0) m1 = 1.0 / (1.0 + nv::exp_neg(-x1));
        else { double ex = nv::exp_neg(x1); m1 = ex / (1.0 + ex); }
        double a2 = fabs((double)head[1]);
        double ex2 = nv::exp_neg(-2.0 * a2);
        double t2 = (1.0 - ex2) / (1.0 + ex2);
        float mean0 = (float)m1, mean1 = (float)(head[1] < 0.0f ? -t2 : t2);
        mean0 = mean0 < 0.0f ? 0.0f : (mean0 > 1.0f ? 1.0f : mean0);          // env.py:656-657
        mean1 = mean1 < -1.0f ? -1.0f : (mean1 > 1.0f ? 1.0f : mean1);
        prev_actions[2 * q] = mean0; prev_actions[2 * q + 1] = mean1;
        const double vp = st.ped_v_pref[q];
        ped_cmd[2 * q] = (double)mean0 * vp;                   // env.py:659-662
        ped_cmd[2 * q + 1] = (double)mean1 * vp;
    }
}

// ============================================================================================
// env.py:685-693: the 512-beam half-plane scan of every pedestrian (what the reference feeds to
// HumanPolicy).  One workgroup per (pedestrian, arena): rectangles of the other agents in LDS,
// march from the pedestrian's integer cell, bearing-culled polygon merge, clip to 6 m.
// ============================================================================================
template <typename Field>
__global__ __launch_bounds__(256) void ped_scan_kernel(navsim_config c, navsim_state st, float* __restrict__ out) {
    constexpr int BLOCK = 256;
    __shared__ float seg[4 * (NAVSIM_MAX_PEDS + 1)][4];
    __shared__ float info_s[4 * (NAVSIM_MAX_PEDS + 1)];
    __shared__ int nseg_s, i0_s, j0_s;
    __shared__ float lx_s, ly_s, lth_s;
    extern __shared__ __attribute__((aligned(16))) char dyn[];       // float2 dir[PB], float rng[PB]
    const int e = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
    const int N = c.max_peds, PB = c.ped_n_beams, H = c.map_h, W = c.map_w;
    int n = st.n_peds[e];
    n = n > N ? N : n;
    if (i >= n) return;
    float2* dir = (float2*)dyn;
    float* rng = (float*)(dyn + sizeof(float2) * (size_t)PB);
    if (tid == 0) {
        nseg_s = 0;
        const double* pp = st.ped_pose + ((size_t)e * N + i) * 3;
        lx_s = (float)pp[0]; ly_s = (float)pp[1]; lth_s = (float)pp[2];              // env.py:386
        nv::xy_to_ij_f32(lx_s, ly_s, c, i0_s, j0_s);                                 // env.py:419
    }
    __syncthreads();
    if (tid <= n && tid != i) {                                                      // env.py:404-414
        const int a = tid;
        const double* pose = (a < n) ? st.ped_pose + ((size_t)e * N + a) * 3 : st.robot_pose + 3 * (size_t)e;
        const double hfx[4] = {0.22, -0.22, -0.22, 0.22}, hfy[4] = {0.19, 0.19, -0.19, -0.19};   // human.py:5-10
        double s, cs;
        nv::sincos(pose[2], s, cs);
        float vx[4], vy[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double x = (a < n) ? hfx[v] : c.robot_seen_footprint[2 * v];
            double y = (a < n) ? hfy[v] : c.robot_seen_footprint[2 * v + 1];
            vx[v] = (float)((cs * x - s * y) + pose[0]);
            vy[v] = (float)((s * x + cs * y) + pose[1]);
        }
        int q = atomicAdd(&nseg_s, 4);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            int w = (v + 1) & 3;
            seg[q + v][0] = vx[v]; seg[q + v][1] = vy[v]; seg[q + v][2] = vx[w]; seg[q + v][3] = vy[w];
        }
    }
    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, H, W);
    const float max_range = march_limit(H, W, c.ped_range_max, c.resolution);
    const float res = (float)c.resolution, rmax = (float)c.ped_range_max;
    const double step = (PB > 1) ? (c.ped_angle_last - c.ped_angle_min) / (double)(PB - 1) : 0.0;
    const float x0 = (float)i0_s, y0 = (float)j0_s;
    const double lth = (double)lth_s;
    for (int k = tid; k < PB; k += BLOCK) {
        double lin = (PB == 1) ? c.ped_angle_min : ((k == PB - 1) ? c.ped_angle_last : (double)k * step + c.ped_angle_min);
        float dx, dy;
        nv::beam_dir((float)(lin + lth), dx, dy);
        dir[k] = make_float2(dx, dy);
        rng[k] = march_ray(field, x0, y0, dx, dy, 0.0f, max_range, (unsigned)W, (unsigned)H) * res;
    }
    __syncthreads();
    const Prims pr = {seg, nullptr, info_s};
    merge_prims_culled_core<BLOCK>(PB, lx_s, ly_s, (float)step, (float)(c.ped_angle_min + lth), nseg_s, 0, pr, dir, rng,
                                   rmax * 1.0001f + 0.01f);
    __syncthreads();
    float* row = out + ((size_t)e * N + i) * PB;
    for (int k = tid; k < PB; k += BLOCK) {
        float r = rng[k];
        r = r < 0.0f ? 0.0f : r;
        r = r > rmax ? rmax : r;
        row[k] = r;
    }
}

__global__ __launch_bounds__(256) void beam_table_kernel(navsim_config c, double* __restrict__ tab) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= c.n_beams) return;
    double s, cs;
    nv::sincos(nv::linspace_k(c, k, nv::linspace_step(c)), s, cs);
    tab[2 * k] = cs;
    tab[2 * k + 1] = s;
}

// test hook: the deterministic math on device
__global__ void math_kernel(int fn, const double* x, const double* x2, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    switch (fn) {
        case 0: nv::sincos(x[i], s, c); out[i] = s; break;
        case 1: nv::sincos(x[i], s, c); out[i] = c; break;
        case 2: out[i] = nv::atan2_(x[i], x2 ? x2[i] : 1.0); break;
        case 3: out[i] = nv::exp_neg(x[i]); break;
        case 4: out[i] = nv::wrap_pi(x[i]); break;
        case 5: out[i] = nv::mod_2pi(x[i]); break;
        case 6: out[i] = (double)nv::sqrt_small_int((float)x[i]); break;
        default: out[i] = 0.0;
    }
}

// microbenchmark (profiles/gather_granularity.py): random 4-byte gathers over a large buffer.
// mode 0: one load per thread; 1: + the neighbour in the same 64-B sector; 2: + the word 64 B away
// in the same 128-B line; 3: + a second independent random word.
__global__ __launch_bounds__(256) void gather_probe_kernel(const float* __restrict__ x, uint64_t n_words,
                                                           int mode, int iters, uint64_t seed,
                                                           float* __restrict__ out) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        uint64_t h = nv::mix64(seed + gid * 0x9E3779B97F4A7C15ULL + (uint64_t)it);
        uint64_t i = h % n_words;
        acc += x[i];
        if (mode == 1) acc += x[i ^ 1];
        if (mode == 2) acc += x[i ^ 16];
        if (mode == 3) acc += x[nv::mix64(h) % n_words];
    }
    out[gid] = acc;
}

thread_local hipError_t g_last_hip_error = hipSuccess;
inline int launch_status() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return NAVSIM_OK;
    g_last_hip_error = e;
    return NAVSIM_E_LAUNCH;
}

// Launch geometry: BLOCK threads per arena, R rays per thread (R = 0: wave-dynamic scan).
// NAVSIM_STEP_VARIANT="<block>x<rays>" overrides the default of the one-launch schedule (tuning).
template <int BLOCK, int R, int MODE>
void launch_step(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                 int reset_only, const uint8_t* mask, char* ws_env, char* ws_prims, float* ws_ranges,
                 hipStream_t s) {
    const bool peds = c->ped_model != NAVSIM_PED_NONE;
    const bool peds_ = c->ped_model != NAVSIM_PED_NONE;
    size_t lds = ((R == 0 || (peds_ && R != 11)) && MODE == kModeFused)
                     ? (size_t)c->n_beams * (sizeof(float2) + sizeof(float)) : 0;      // dir + rng
    // analytic tile table in LDS: one-launch schedule, 1 ray per thread, packed field, table <= 40 KiB and
    // 16-byte granular per arena (so arena e's table starts at e * tile_bytes)
    size_t tile_bytes = 0;
    if (MODE == kModeFused && R == 1 && st->tile_table && c->field_format == NAVSIM_FIELD_U16T &&
        !getenv("NAVSIM_NO_TILES")) {
        size_t tb = navsim_tile_table_bytes(1, c->map_h, c->map_w);
        if (tb <= 40960) tile_bytes = tb;
    }
    const size_t lds_scan = lds;
    lds += tile_bytes;
    if (peds) lds = ((lds + 15) & ~(size_t)15) + ped_lds_bytes(c->max_peds);       // PedShared behind dir / rng
    if (const char* pad = getenv("NAVSIM_LDS_PAD")) lds += (size_t)atoi(pad);   // occupancy experiments only
    // pedestrians ahead of the step, one wavefront per arena (NAVSIM_PED_SPLIT=0: inside the step as before)
    // (pays when the chip runs several generations of arenas: c3 13.2 -> 14.0 M env-steps/s; a 512-arena
    // launch is latency-bound and loses 2 % to the extra kernel, so small batches keep the fused form)
    if (MODE == kModeFused && peds && !reset_only && c->max_peds <= 56) {                        // <= 64 KB of LDS
        const char* v = getenv("NAVSIM_PED_SPLIT");             // "0" never, "1" always, unset: large batches
        const int split = v ? (v[0] != '0') : (c->n_envs >= 3072);
        if (split) {
            const size_t pl = (((size_t)c->max_peds * (c->max_peds + 1) * sizeof(double2) + 15) & ~(size_t)15) +
                              ped_lds_bytes(c->max_peds);
            if (c->field_format == NAVSIM_FIELD_U16T)      ped_update_kernel<FieldU16T><<<c->n_envs, 64, pl, s>>>(*c, *st);
            else if (c->field_format == NAVSIM_FIELD_F32S) ped_update_kernel<FieldF32S><<<c->n_envs, 64, pl, s>>>(*c, *st);
            else                                           ped_update_kernel<FieldF32><<<c->n_envs, 64, pl, s>>>(*c, *st);
            reset_only |= 2;
        }
    }
    if (c->field_format == NAVSIM_FIELD_U16T) {
        if (peds) navsim_step_kernel<BLOCK, R, true, FieldU16T, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
        else      navsim_step_kernel<BLOCK, R, false, FieldU16T, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
    } else if (c->field_format == NAVSIM_FIELD_F32S) {
        if (peds) navsim_step_kernel<BLOCK, R, true, FieldF32S, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
        else      navsim_step_kernel<BLOCK, R, false, FieldF32S, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
    } else {
        if (peds) navsim_step_kernel<BLOCK, R, true, FieldF32, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
        else      navsim_step_kernel<BLOCK, R, false, FieldF32, MODE><<<c->n_envs, BLOCK, lds, s>>>(*c, *st, *io, reset_only, mask, ws_env, ws_prims, ws_ranges, (unsigned)lds_scan, (unsigned)tile_bytes);
    }
}

constexpr size_t kPrimBytes = sizeof(float) * 4 * 4 * NAVSIM_MAX_PEDS + sizeof(float) * 2 * 2 * NAVSIM_MAX_PEDS;

size_t workspace_bytes(const navsim_config* c) {
    size_t E = (size_t)c->n_envs;
    size_t b = E * kPoolEnvBytes + E * (size_t)c->n_beams * sizeof(float);
    if (c->ped_model != NAVSIM_PED_NONE) b += E * kPrimBytes;
    return b;
}

// pooled schedule: prologue per arena, one flat pool of march tasks, epilogue per arena, and the
// same pair again for the (few) arenas that were reverted / respawned
int run_pooled(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
               const uint8_t* mask, hipStream_t s) {
    const size_t E = (size_t)c->n_envs;
    char* ws_env = (char*)st->workspace;
    float* ws_ranges = (float*)(ws_env + E * kPoolEnvBytes);
    char* ws_prims = (char*)(ws_ranges + E * (size_t)c->n_beams);
    const unsigned G = (unsigned)((c->n_beams + 63) / 64);
    const unsigned n_logical = (unsigned)((E * G + 3) / 4);
    const unsigned grid = (n_logical + 7u) & ~7u;
    launch_step<64, 1, kModePre>(c, st, io, reset_only, mask, ws_env, ws_prims, ws_ranges, s);
    for (int pass = 0; pass < (reset_only ? 1 : 2); ++pass) {
        if (c->field_format == NAVSIM_FIELD_U16T)
            pool_scan_kernel<FieldU16T><<<grid, 256, 0, s>>>(*c, *st, ws_env, ws_ranges, pass, n_logical);
        else if (c->field_format == NAVSIM_FIELD_F32S)
            pool_scan_kernel<FieldF32S><<<grid, 256, 0, s>>>(*c, *st, ws_env, ws_ranges, pass, n_logical);
        else
            pool_scan_kernel<FieldF32><<<grid, 256, 0, s>>>(*c, *st, ws_env, ws_ranges, pass, n_logical);
        if (pass == 0) launch_step<256, 1, kModePost>(c, st, io, reset_only, mask, ws_env, ws_prims, ws_ranges, s);
        else           launch_step<256, 1, kModeFinal>(c, st, io, reset_only, mask, ws_env, ws_prims, ws_ranges, s);
    }
    return launch_status();
}

int dispatch_step(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                  int reset_only, const uint8_t* mask, hipStream_t s) {
    const char* mode = getenv("NAVSIM_STEP_MODE");
    if (st->workspace && mode && strcmp(mode, "pool") == 0)      // opt-in: measured slower than one launch
        return run_pooled(c, st, io, reset_only, mask, s);
    int block = 0, rays = 0;
    const char* v = getenv("NAVSIM_STEP_VARIANT");
    if (v && sscanf(v, "%dx%d", &block, &rays) != 2) { block = 0; rays = 0; }
    if (!block) {
        // Threads per arena.  With >= 12 arenas per CU the chip is kept full by 256-thread workgroups (8 per
        // CU, several generations).  With fewer arenas a launch is one generation whose length is a
        // workgroup's own march, i.e. beams per thread: wider workgroups shorten it (measured, c2 world:
        // 2048 arenas 14.4 / 16.4 / 13.7 M env-steps/s for 256 / 512 / 1024 threads; 1024 arenas
        // 8.8 / 11.3 / 11.6; 512 arenas 5.2 / 6.7 / 7.3; 4096 arenas 19.5 / 17.4 / -).
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0, v2 = 0;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&v2, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v2 > 0)
                n_cu = v2;
            else
                n_cu = 256;
        }
        const int B = c->n_beams;
        const long per_cu_x2 = 2L * c->n_envs / n_cu;            // arenas per CU, doubled
        rays = 1;
        if (B <= 64) block = 64;
        else if (per_cu_x2 >= 24 || B <= 256) block = 256;
        else if (per_cu_x2 >= 12 || B <= 512) block = 512;
        else block = 1024;
    }
#define NAVSIM_VARIANT(BK, RR) if (block == BK && rays == RR) { launch_step<BK, RR, kModeFused>(c, st, io, reset_only, mask, nullptr, nullptr, nullptr, s); return launch_status(); }
    NAVSIM_VARIANT(64, 1)
    NAVSIM_VARIANT(256, 0)
    NAVSIM_VARIANT(256, 1)
    NAVSIM_VARIANT(320, 1)
    NAVSIM_VARIANT(384, 1)
    NAVSIM_VARIANT(512, 1)
    NAVSIM_VARIANT(192, 1)
    NAVSIM_VARIANT(256, 11)
    NAVSIM_VARIANT(256, 2)
    NAVSIM_VARIANT(256, 5)
    NAVSIM_VARIANT(512, 0)
    NAVSIM_VARIANT(768, 1)
    NAVSIM_VARIANT(1024, 1)
#undef NAVSIM_VARIANT
    return NAVSIM_E_UNSUPPORTED;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
// kernels that want more than 64 KB of dynamic LDS must say so once
static int allow_lds(const void* kernel, size_t lds) {
    if (lds <= 64 * 1024) return NAVSIM_OK;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess
               ? NAVSIM_OK : NAVSIM_E_UNSUPPORTED;
}

extern "C" {

int navsim_abi_version(void) { return NAVSIM_ABI_VERSION; }

const char* navsim_error_string(int code) {
    switch (code) {
        case NAVSIM_OK: return "ok";
        case NAVSIM_E_ARG: return "invalid argument";
        case NAVSIM_E_LAUNCH: return "kernel launch failed";
        case NAVSIM_E_UNSUPPORTED: return "configuration outside compiled limits";
        case NAVSIM_E_NODEVICE: return "no HIP device";
        default: return "unknown error";
    }
}

int navsim_default_config(navsim_config* c) {
    if (!c) return NAVSIM_E_ARG;
    memset(c, 0, sizeof(*c));
    c->n_envs = 1;
    c->n_beams = 512;                          // keti_robot.py:48
    c->map_h = 400; c->map_w = 400;            // map_generator.py:133
    c->max_peds = 16;
    c->n_scan_stack = 1;                       // __init__.py:11
    c->ped_model = NAVSIM_PED_NONE;
    c->lidar_legs = 1;                         // env.py:697
    c->resolution = 0.05;                      // map_generator.py:139
    c->time_step = 0.2;                        // __init__.py:8
    c->angle_min = -3.141592;                  // keti_robot.py:45
    c->angle_last = 3.141592 - 0.0122718463;   // keti_robot.py:44,46 / env.py:389
    c->range_max = 25.0;                       // keti_robot.py:47
    c->axle_offset = 0.14474;                  // keti_robot.py:73
    c->min_turning_radius = 0.0;               // __init__.py:9
    c->distance_threshold = 0.5;               // __init__.py:10
    c->reward_scale = 15.0;                    // __init__.py:19-25
    c->reward_success_factor = 1.0;
    c->reward_crash_factor = 1.0;
    c->reward_progress_factor = 0.001;
    c->reward_forward_factor = 0.0;
    c->reward_rotation_factor = 0.005;
    c->reward_discomfort_factor = 0.01;
    c->sfm_tau = 0.5;
    c->sfm_k_desired = 1.0;
    c->sfm_k_social = 2.1;
    c->sfm_k_obstacle = 10.0;
    c->sfm_lambda = 2.0;
    c->sfm_gamma = 0.35;
    c->sfm_n = 2.0;
    c->sfm_n_prime = 3.0;
    c->sfm_sigma_obstacle = 0.8;
    c->sfm_agent_radius = 0.35;
    c->ped_angle_min = -1.57079632679;      // human.py:13 
    c->ped_angle_last = 1.57079632679 - 0.00613592315;   // human.py:12,14; env.py:389 
    c->ped_range_max = 6.0;                 // human.py:15 
    c->ped_n_beams = 512;                   // human.py:16 
    {   // keti_robot.py:18-23 threshold_footprint 
        const double fp[8] = {0.6, 0.6, -0.7, 0.6, -0.7, -0.6, 0.6, -0.6};
        for (int i = 0; i < 8; ++i) c->robot_seen_footprint[i] = fp[i];
    }
    c->regen_cap = 64;
    c->obstacle_number = 10;                // __init__.py:34
    c->obstacle_width_lo = 0.3;             // __init__.py:35
    c->obstacle_width_hi = 1.0;
    c->spawn_clearance = 1.2;
    c->ped_clearance = 0.5;
    c->min_goal_dist = 10.0;                // __init__.py:17-18
    c->max_goal_dist = 20.0;
    c->ped_min_robot_dist = 4.0;            // env.py:372
    c->ped_min_goal_dist = 10.0;            // env.py:788-791
    c->v_pref_lo = 0.0;                     // __init__.py:14
    c->v_pref_hi = 0.6;
    c->has_legs_ratio = 0.5;                // __init__.py:15
    c->regen_indoor_ratio = 0.0;
    c->seed = 1234;
    return NAVSIM_OK;
}

size_t navsim_sizeof_config(void) { return sizeof(navsim_config); }
size_t navsim_sizeof_state(void) { return sizeof(navsim_state); }
size_t navsim_sizeof_step_io(void) { return sizeof(navsim_step_io); }

size_t navsim_build_dt_workspace_bytes(int32_t n_maps, int32_t H, int32_t W) {
    if (n_maps <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)n_maps * H * W * sizeof(uint16_t);
}

size_t navsim_field_bytes(int32_t n_maps, int32_t H, int32_t W, int32_t format) {
    if (n_maps <= 0 || H <= 0 || W <= 0) return 0;
    if (format == NAVSIM_FIELD_F32) return (size_t)n_maps * H * W * sizeof(float);
    if (format == NAVSIM_FIELD_U16T) return (size_t)n_maps * ((H + 7) / 8) * ((W + 7) / 8) * 64 * sizeof(uint16_t);
    if (format == NAVSIM_FIELD_F32S) return (size_t)n_maps * ((H + 3) / 4) * ((W + 7) / 8) * 32 * sizeof(float);
    return 0;
}

int navsim_build_field(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, int32_t format, void* field,
                       float* overflow, int32_t* n_saturated, void* workspace, size_t workspace_bytes,
                       void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!occ || !field || !workspace || n_maps < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    if (format != NAVSIM_FIELD_F32 && format != NAVSIM_FIELD_U16T && format != NAVSIM_FIELD_F32S) return NAVSIM_E_UNSUPPORTED;
    if (format == NAVSIM_FIELD_F32S && !overflow) return NAVSIM_E_ARG;
    if (H >= kDtInf || W >= kDtInf || (size_t)W * 4 > 64 * 1024) return NAVSIM_E_UNSUPPORTED;
    size_t per_map = (size_t)H * W * sizeof(uint16_t);
    size_t chunk = workspace_bytes / per_map;
    if (chunk == 0) return NAVSIM_E_ARG;
    if (chunk > 65535) chunk = 65535;
    hipStream_t s = (hipStream_t)stream;
    const size_t field_per_map = navsim_field_bytes(1, H, W, format);
    if (format != NAVSIM_FIELD_F32)        // padding cells of edge tiles are never read; keep them defined
        (void)hipMemsetAsync(field, format == NAVSIM_FIELD_U16T ? 0xFF : 0, field_per_map * (size_t)n_maps, s);
    for (int32_t m0 = 0; m0 < n_maps; m0 += (int32_t)chunk) {
        int32_t m = (n_maps - m0 < (int32_t)chunk) ? n_maps - m0 : (int32_t)chunk;
        dt_columns_kernel<<<dim3((W + 63) / 64, m), 64 * kColSeg, 0, s>>>(occ + (size_t)m0 * H * W,
                                                                   (uint16_t*)workspace, H, W, nullptr);
        void* f = (char*)field + field_per_map * (size_t)m0;
        float* o = overflow ? overflow + (size_t)m0 * H * W : nullptr;
        if (format == NAVSIM_FIELD_F32)
            dt_rows_kernel<0><<<dim3(H, m), 256, (size_t)W * 4, s>>>((const uint16_t*)workspace, f, nullptr, nullptr, H, W, nullptr);
        else if (format == NAVSIM_FIELD_U16T)
            dt_rows_kernel<1><<<dim3(H, m), 256, (size_t)W * 4, s>>>((const uint16_t*)workspace, f, o, n_saturated, H, W, nullptr);
        else
            dt_rows_kernel<2><<<dim3(H, m), 256, (size_t)W * 4, s>>>((const uint16_t*)workspace, f, o, n_saturated, H, W, nullptr);
    }
    return launch_status();
}

size_t navsim_tile_table_bytes(int32_t n_maps, int32_t H, int32_t W) {
    if (n_maps <= 0 || H <= 0 || W <= 0) return 0;
    size_t n_tiles = (size_t)((H + 7) / 8) * ((W + 7) / 8);
    return (size_t)n_maps * ((n_tiles + 3) & ~(size_t)3) * sizeof(uint32_t);     // 16-byte granular per arena
}

size_t navsim_build_tiles_workspace_bytes(int32_t n_maps, int32_t H, int32_t W) {
    if (n_maps <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)n_maps * H * W * 10;               // nearest row int16 + d2 int32 + (ox, oy) int16 x 2
}

int navsim_build_tiles(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, uint32_t* tiles,
                       void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!occ || !tiles || !workspace || n_maps < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    if (H > 16383 || W > 16383 || (size_t)W * 4 > 64 * 1024) return NAVSIM_E_UNSUPPORTED;
    const size_t cells = (size_t)H * W;
    size_t chunk = workspace_bytes / (cells * 10);
    if (chunk == 0) return NAVSIM_E_ARG;
    if (chunk > 65535) chunk = 65535;
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = ((H + 7) / 8) * ((W + 7) / 8);
    for (int32_t m0 = 0; m0 < n_maps; m0 += (int32_t)chunk) {
        int32_t m = (n_maps - m0 < (int32_t)chunk) ? n_maps - m0 : (int32_t)chunk;
        int32_t* d2 = (int32_t*)workspace;
        int16_t* oxy = (int16_t*)(d2 + (size_t)m * cells);
        int16_t* nr = oxy + 2 * (size_t)m * cells;
        ft_columns_kernel<<<dim3((W + 255) / 256, m), 256, 0, s>>>(occ + (size_t)m0 * cells, nr, H, W);
        ft_rows_kernel<<<dim3(H, m), 256, (size_t)W * 4, s>>>(nr, d2, oxy, H, W);
        tile_table_kernel<<<dim3(n_tiles, m), 64, 0, s>>>(d2, oxy, tiles + (size_t)m0 * ((n_tiles + 3) & ~3), H, W);
    }
    return launch_status();
}

int navsim_build_dt(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, float* field,
                    void* workspace, size_t workspace_bytes, void* stream) {
    return navsim_build_field(occ, n_maps, H, W, NAVSIM_FIELD_F32, field, nullptr, nullptr, workspace,
                              workspace_bytes, stream);
}

int navsim_cast_static(const float* field, int32_t E, int32_t H, int32_t W, const float* q,
                       int32_t n_per_env, float max_range, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!field || E < 0 || n_per_env < 0 || H <= 0 || W <= 0) return NAVSIM_E_ARG;
    long long total = (long long)E * n_per_env;
    if (total == 0) return NAVSIM_OK;
    if (!q || !out) return NAVSIM_E_ARG;
    cast_static_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        field, H, W, q, n_per_env, total, max_range, out);
    return launch_status();
}

int navsim_render_polys(float* ranges, const double* angles, int32_t E, int32_t B, const float* verts,
                        const int32_t* n_verts, int32_t V, const float* origin, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!ranges || !angles || !verts || !n_verts || !origin || E < 0 || B < 0 || V < 0) return NAVSIM_E_ARG;
    if (E == 0 || B == 0) return NAVSIM_OK;
    render_polys_kernel<<<dim3((B + 255) / 256, E), 256, 0, (hipStream_t)stream>>>(ranges, angles, B, verts,
                                                                                  n_verts, V, origin);
    return launch_status();
}

int navsim_render_legs(float* ranges, const double* angles, int32_t E, int32_t B, const float* agents,
                       const int32_t* n_agents, int32_t A, const float* origin, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!ranges || !angles || !agents || !n_agents || !origin || E < 0 || B < 0 || A < 0) return NAVSIM_E_ARG;
    if (E == 0 || B == 0) return NAVSIM_OK;
    render_legs_kernel<<<dim3((B + 255) / 256, E), 256, 0, (hipStream_t)stream>>>(ranges, angles, B, agents,
                                                                                 n_agents, A, origin);
    return launch_status();
}

int navsim_integrate(double* pose, const double* cmd, double* vel_out, int32_t n, double dt, double off,
                     void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!pose || !cmd || n < 0) return NAVSIM_E_ARG;
    if (n == 0) return NAVSIM_OK;
    integrate_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(pose, cmd, vel_out, n, dt, off);
    return launch_status();
}

int navsim_reward_done(const navsim_config* c, const void* obs, const void* goals, int32_t is64, int32_t n,
                       const float* thr, const float* dthr, double* reward, uint8_t* done,
                       float* is_success, float* is_crash, double* distance, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!c || !obs || !goals || !thr || !dthr || n < 0) return NAVSIM_E_ARG;
    if (n == 0) return NAVSIM_OK;
    if (is64)
        reward_done_kernel<double><<<n, 256, 0, (hipStream_t)stream>>>(
            *c, (const double*)obs, (const double*)goals, thr, dthr, reward, done, is_success, is_crash, distance);
    else
        reward_done_kernel<float><<<n, 256, 0, (hipStream_t)stream>>>(
            *c, (const float*)obs, (const float*)goals, thr, dthr, reward, done, is_success, is_crash, distance);
    return launch_status();
}

int navsim_scan_threshold(const navsim_config* c, const float* fp, int32_t nvert, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!c || !fp || !out || nvert < 2 || nvert > 16) return NAVSIM_E_ARG;
    scan_threshold_kernel<<<(c->n_beams + 255) / 256, 256, 0, (hipStream_t)stream>>>(*c, fp, nvert, out);
    return launch_status();
}

int navsim_beam_table(const navsim_config* c, double* table, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!c || !table || c->n_beams < 1) return NAVSIM_E_ARG;
    beam_table_kernel<<<(c->n_beams + 255) / 256, 256, 0, (hipStream_t)stream>>>(*c, table);
    return launch_status();
}

static int check_step_args(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                           int reset_only) {
    if (!c || !st || !io) return NAVSIM_E_ARG;
    if (c->n_envs < 0 || c->n_beams < 1 || c->n_scan_stack < 1 || c->map_h < 1 || c->map_w < 1) return NAVSIM_E_ARG;
    if (c->max_peds > NAVSIM_MAX_PEDS) return NAVSIM_E_UNSUPPORTED;
    if (c->field_format != NAVSIM_FIELD_F32 && c->field_format != NAVSIM_FIELD_U16T &&
        c->field_format != NAVSIM_FIELD_F32S) return NAVSIM_E_UNSUPPORTED;
    if (c->field_format == NAVSIM_FIELD_F32S && !st->field_overflow) return NAVSIM_E_ARG;
    if (!st->field || !st->scan_threshold || !st->scan_discomfort || !st->robot_pose || !st->robot_goal ||
        !st->prev_action || !st->prev_pose || !st->n_hist || !st->episode || !st->steps || !io->obs)
        return NAVSIM_E_ARG;
    if (!reset_only && (!io->action || !io->reward || !io->done || !io->is_success || !io->is_crash ||
                        !io->distance))
        return NAVSIM_E_ARG;
    if (!reset_only && c->n_scan_stack > 1 && !io->obs_prev) return NAVSIM_E_ARG;
    if (c->ped_model != NAVSIM_PED_NONE) {
        if (!st->n_peds || !st->ped_pose || !st->ped_vel || !st->ped_prev_yaw || !st->ped_dist ||
            !st->ped_has_legs || !st->ped_waypoints || !st->ped_n_waypoints)
            return NAVSIM_E_ARG;
        if (c->ped_model == NAVSIM_PED_EXTERNAL && !st->ped_cmd && !reset_only) return NAVSIM_E_ARG;
        if (c->ped_model == NAVSIM_PED_SFM && !st->ped_v_pref) return NAVSIM_E_ARG;
    }
    if (c->auto_reset && c->n_spawn > 0 && (!st->spawn_pose || !st->spawn_goal)) return NAVSIM_E_ARG;
    return NAVSIM_OK;
}

int navsim_ped_scans(const navsim_config* c, const navsim_state* st, float* out, void* stream) {
    (void)hipGetLastError();
    if (!c || !st || !out || c->ped_model == NAVSIM_PED_NONE || !st->n_peds || !st->ped_pose || !st->robot_pose ||
        !st->field || c->ped_n_beams < 1 || c->max_peds < 1) return NAVSIM_E_ARG;
    if (c->max_peds > NAVSIM_MAX_PEDS || c->ped_n_beams > 4096) return NAVSIM_E_UNSUPPORTED;
    if (c->field_format == NAVSIM_FIELD_F32S && !st->field_overflow) return NAVSIM_E_ARG;
    if (c->n_envs == 0) return NAVSIM_OK;
    dim3 grid(c->max_peds, c->n_envs);
    size_t lds = (size_t)c->ped_n_beams * (sizeof(float2) + sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    if (c->field_format == NAVSIM_FIELD_U16T)      ped_scan_kernel<FieldU16T><<<grid, 256, lds, s>>>(*c, *st, out);
    else if (c->field_format == NAVSIM_FIELD_F32S) ped_scan_kernel<FieldF32S><<<grid, 256, lds, s>>>(*c, *st, out);
    else if (c->field_format == NAVSIM_FIELD_F32)  ped_scan_kernel<FieldF32><<<grid, 256, lds, s>>>(*c, *st, out);
    else return NAVSIM_E_UNSUPPORTED;
    return launch_status();
}

int navsim_costmap(const uint8_t* occ, int32_t n_maps, int32_t H, int32_t W, uint8_t* cost, void* stream) {
    (void)hipGetLastError();
    if (!occ || !cost || n_maps < 0 || H < 5 || W < 5) return NAVSIM_E_ARG;
    if (n_maps == 0) return NAVSIM_OK;
    if (n_maps > 65535) return NAVSIM_E_UNSUPPORTED;
    int cells = (H / 5) * (W / 5);
    costmap_kernel<<<dim3((cells + 255) / 256, n_maps), 256, 0, (hipStream_t)stream>>>(occ, H, W, cost, nullptr, nullptr);
    return launch_status();
}

size_t navsim_plan_workspace_bytes(int32_t n_queries, int32_t Hc, int32_t Wc) {
    (void)n_queries; (void)Hc; (void)Wc;
    return 0;                                              // the search lives in LDS; kept for ABI stability
}

int navsim_plan(const uint8_t* cost, const int32_t* map_index, int32_t n, int32_t Hc, int32_t Wc, double res_c,
                double ox, double oy, const double* start, const double* goal, double interval, int32_t max_wp,
                double* wp, int32_t* n_wp, int32_t* path_cells, double* path_len, void* workspace,
                size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    (void)workspace; (void)workspace_bytes;
    if (!cost || !start || !goal || !wp || !n_wp || n < 0 || Hc <= 0 || Wc <= 0 || max_wp < 1) return NAVSIM_E_ARG;
    if (!plan_fits(Hc, Wc)) return NAVSIM_E_UNSUPPORTED;                // LDS-resident search
    if (n == 0) return NAVSIM_OK;
    if (allow_lds((const void*)plan_kernel, plan_lds(Hc, Wc)) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
    plan_kernel<<<n, 256, plan_lds(Hc, Wc), (hipStream_t)stream>>>(
        cost, map_index, Hc, Wc, res_c, ox, oy, start, goal, interval, max_wp, wp, n_wp, path_cells, path_len);
    return launch_status();
}

size_t navsim_regen_workspace_bytes(const navsim_config* c) {
    if (!c || c->regen_cap < 1) return 0;
    const size_t M = (size_t)c->regen_cap, cells = (size_t)c->map_h * c->map_w;
    size_t b = 16 + M * 4;                                  // count, list
    b = (b + 255) & ~(size_t)255;
    b += ((size_t)c->n_envs + 255) & ~(size_t)255;          // mask
    b += M * cells;                                         // occupancy scratch
    b += M * cells * sizeof(uint16_t);                      // column pass
    b += M * navsim_field_bytes(1, c->map_h, c->map_w, c->field_format);
    b += M * (10000 + sizeof(int)) + 512;                   // corridor grids, map kinds
    if (c->regen_plan) {
        const size_t cc = (size_t)(c->map_h / 5) * (c->map_w / 5), P = NAVSIM_MAX_WAYPOINTS;
        const size_t Q = (size_t)(c->n_spawn > c->max_peds ? c->n_spawn : c->max_peds);
        b += M * cc + 256;                                            // costmaps
        b += M * Q * (2 + 2 + 2 * P + 1) * sizeof(double) + 256;      // start, goal, waypoints, length
        b += M * Q * sizeof(int32_t) + 256;                           // waypoint counts
        b += M * (Q + (size_t)c->n_spawn + (size_t)c->max_peds) + 256;   // active, resolved flags
        b += 8 * 256;                                                 // alignment of the ten sub-buffers
    }
    return b + 1024;
}

int navsim_regen(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, void* workspace,
                 size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!c || !st || !io || !io->done || !io->obs || !workspace) return NAVSIM_E_ARG;
    if (c->map_h != c->map_w || c->n_spawn < 1 || c->regen_cap < 1 || c->obstacle_number > 64 || st->tile_table || st->field_overflow ||
        (c->field_format != NAVSIM_FIELD_F32 && c->field_format != NAVSIM_FIELD_U16T) || c->shared_field)
        return NAVSIM_E_UNSUPPORTED;
    if (workspace_bytes < navsim_regen_workspace_bytes(c) || !st->spawn_pose || !st->spawn_goal) return NAVSIM_E_ARG;
    if (c->regen_plan && (c->n_spawn > 256 || c->map_h < 5 || !plan_fits(c->map_h / 5, c->map_w / 5) ||
                          allow_lds((const void*)regen_plan_kernel, plan_lds(c->map_h / 5, c->map_w / 5)) != NAVSIM_OK))
        return NAVSIM_E_UNSUPPORTED;
    int rc = check_step_args(c, st, io, 1);
    if (rc != NAVSIM_OK) return rc;
    if (c->n_envs == 0) return NAVSIM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int M = c->regen_cap, H = c->map_h, W = c->map_w;
    const size_t cells = (size_t)H * W;
    char* w = (char*)workspace;
    int* count = (int*)w;
    int* list = count + 4;
    size_t off = (16 + (size_t)M * 4 + 255) & ~(size_t)255;
    uint8_t* mask = (uint8_t*)(w + off);
    off += ((size_t)c->n_envs + 255) & ~(size_t)255;
    uint8_t* occ = (uint8_t*)(w + off);
    off += (size_t)M * cells;
    off = (off + 255) & ~(size_t)255;
    uint16_t* cols = (uint16_t*)(w + off);
    off += (size_t)M * cells * sizeof(uint16_t);
    off = (off + 255) & ~(size_t)255;
    char* fscratch = w + off;
    const size_t fbytes = navsim_field_bytes(1, H, W, c->field_format);
    off += fbytes * (size_t)M;
    off = (off + 255) & ~(size_t)255;
    uint8_t* grids = (uint8_t*)(w + off);
    off += (size_t)M * 10000;
    off = (off + 255) & ~(size_t)255;
    int* kind = (int*)(w + off);
    off += (size_t)M * sizeof(int);
    regen_select_kernel<<<1, 1024, 0, s>>>(io->done, c->n_envs, M, count, list, mask);
    regen_indoor_kernel<<<M, 256, 0, s>>>(*c, *st, count, list, grids, kind);
    regen_maps_kernel<<<dim3(M, kRegenSlices), 256, 0, s>>>(*c, *st, count, list, occ, grids, kind);
    if (c->field_format == NAVSIM_FIELD_U16T) (void)hipMemsetAsync(fscratch, 0xFF, fbytes * (size_t)M, s);
    dt_columns_kernel<<<dim3((W + 63) / 64, M), 64 * kColSeg, 0, s>>>(occ, cols, H, W, count);
    if (c->field_format == NAVSIM_FIELD_F32)
        dt_rows_kernel<0><<<dim3(H, M), 256, (size_t)W * 4, s>>>(cols, fscratch, nullptr, nullptr, H, W, count);
    else
        dt_rows_kernel<1><<<dim3(H, M), 256, (size_t)W * 4, s>>>(cols, fscratch, nullptr, nullptr, H, W, count);
    regen_field_kernel<<<dim3(M, kRegenSlices), 256, 0, s>>>(*st, count, list, fscratch, fbytes);
    if (c->regen_plan) {
        const int Hc = H / 5, Wc = W / 5, P = NAVSIM_MAX_WAYPOINTS;
        const size_t cc = (size_t)Hc * Wc;
        const int Q = c->n_spawn > c->max_peds ? c->n_spawn : c->max_peds;
        auto take = [&](size_t bytes) { off = (off + 255) & ~(size_t)255; char* p = w + off; off += bytes; return p; };
        RegenPlanWs ws;
        ws.Q = Q;
        ws.cost = (uint8_t*)take((size_t)M * cc);
        ws.cost_by_arena = st->costmap != nullptr;
        if (st->costmap) ws.cost = st->costmap;
        ws.qstart = (double*)take((size_t)M * Q * 2 * sizeof(double));
        ws.qgoal = (double*)take((size_t)M * Q * 2 * sizeof(double));
        ws.qwp = (double*)take((size_t)M * Q * P * 2 * sizeof(double));
        ws.qlen = (double*)take((size_t)M * Q * sizeof(double));
        ws.qnwp = (int32_t*)take((size_t)M * Q * sizeof(int32_t));
        ws.active = (uint8_t*)take((size_t)M * Q);
        ws.res_robot = (uint8_t*)take((size_t)M * c->n_spawn);
        ws.res_ped = (uint8_t*)take((size_t)M * (c->max_peds > 0 ? c->max_peds : 1));
        const size_t lds = plan_lds(Hc, Wc);
        costmap_kernel<<<dim3(((int)cc + 255) / 256, M), 256, 0, s>>>(occ, H, W, ws.cost, count,
                                                                      st->costmap ? list : nullptr);
        regen_install_kernel<<<M, 256, 0, s>>>(*c, *st, count, list, fscratch, fbytes, ws);
        for (int round = 0; round <= 4; ++round) {
            regen_robot_round_kernel<<<M, 256, 0, s>>>(*c, *st, count, list, ws, round);
            if (round < 4) regen_plan_kernel<<<M * Q, 256, lds, s>>>(*c, *st, count, list, ws, 0);
        }
        if (c->ped_model != NAVSIM_PED_NONE && c->max_peds > 0)
            for (int round = 0; round <= 4; ++round) {
                regen_ped_round_kernel<<<M, 256, 0, s>>>(*c, *st, count, list, ws, round);
                if (round < 4) regen_plan_kernel<<<M * Q, 256, lds, s>>>(*c, *st, count, list, ws, 1);
            }
    } else if (st->costmap) {
        costmap_kernel<<<dim3(((H / 5) * (W / 5) + 255) / 256, M), 256, 0, s>>>(occ, H, W, st->costmap, count, list);
    }
    if (!c->regen_plan) {
        if (c->field_format == NAVSIM_FIELD_F32)
            regen_commit_kernel<FieldF32><<<M, 256, 0, s>>>(*c, *st, count, list, fscratch, fbytes);
        else
            regen_commit_kernel<FieldU16T><<<M, 256, 0, s>>>(*c, *st, count, list, fscratch, fbytes);
    }
    if (launch_status() != NAVSIM_OK) return NAVSIM_E_LAUNCH;
    // first observation of the new episodes; the other arenas keep the row the step just wrote
    navsim_step_io io2 = *io;
    io2.obs_prev = io->obs;
    return dispatch_step(c, st, &io2, 1, mask, s);
}

size_t navsim_replan_workspace_bytes(const navsim_config* c, int32_t max_queries) {
    if (!c || max_queries < 0) return 0;
    return 256 + (((size_t)max_queries * sizeof(int32_t) + 255) & ~(size_t)255) + (size_t)c->n_envs * sizeof(uint64_t);
}

int navsim_replan(const navsim_config* c, const navsim_state* st, int32_t max_queries, void* workspace,
                  size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!c || !st || !workspace || max_queries < 0 || !st->costmap || !st->ped_pose || !st->ped_waypoints ||
        !st->ped_n_waypoints || !st->n_peds || !st->steps || !st->episode)
        return NAVSIM_E_ARG;
    if (workspace_bytes < navsim_replan_workspace_bytes(c, max_queries)) return NAVSIM_E_ARG;
    if (c->ped_model == NAVSIM_PED_NONE || c->n_envs == 0 || max_queries == 0) return NAVSIM_OK;
    const int Hc = c->map_h / 5, Wc = c->map_w / 5;
    if (Hc < 1 || Wc < 1 || !plan_fits(Hc, Wc) || allow_lds((const void*)replan_kernel, plan_lds(Hc, Wc)) != NAVSIM_OK)
        return NAVSIM_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int* count = (int*)workspace;
    int* list = (int*)((char*)workspace + 256);
    uint64_t* due = (uint64_t*)((char*)workspace + 256 + (((size_t)max_queries * sizeof(int32_t) + 255) & ~(size_t)255));
    replan_flag_kernel<<<c->n_envs, 64, 0, s>>>(*c, *st, due);
    replan_select_kernel<<<1, 1024, 0, s>>>(due, c->n_envs, c->max_peds, max_queries, count, list);
    replan_kernel<<<max_queries, 256, plan_lds(Hc, Wc), s>>>(*c, *st, count, list);
    return launch_status();
}

constexpr int kPolicyChunk = 32768;          // pedestrians per pass: bounds the feature scratch (512 MiB)

size_t navsim_ped_policy_workspace_bytes(const navsim_config* c) {
    if (!c) return 0;
    size_t P = (size_t)c->n_envs * (size_t)c->max_peds;
    size_t chunk = P < (size_t)kPolicyChunk ? P : (size_t)kPolicyChunk;
    return 1024 + (size_t)kPolH2 * kPolIn2 * sizeof(float) + chunk * (size_t)(kPolFeat + kPolH1) * sizeof(float);
}

int navsim_ped_policy(const navsim_config* c, const navsim_state* st, const navsim_policy_weights* w,
                      const float* ped_scans, float* prev_actions, double* ped_cmd, void* workspace,
                      size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!c || !st || !w || !ped_scans || !prev_actions || !ped_cmd || !workspace || !st->ped_pose ||
        !st->ped_waypoints || !st->ped_n_waypoints || !st->ped_v_pref || !st->n_peds)
        return NAVSIM_E_ARG;
    if (!w->cv1_w || !w->cv1_b || !w->cv2_w || !w->cv2_b || !w->fc1_w || !w->fc1_b || !w->fc2_w || !w->fc2_b ||
        !w->a1_w || !w->a1_b || !w->a2_w || !w->a2_b)
        return NAVSIM_E_ARG;
    if (c->ped_n_beams != 512 || c->max_peds < 1) return NAVSIM_E_UNSUPPORTED;
    if (workspace_bytes < navsim_ped_policy_workspace_bytes(c)) return NAVSIM_E_ARG;
    const size_t P = (size_t)c->n_envs * (size_t)c->max_peds;
    if (P == 0) return NAVSIM_OK;
    hipStream_t s = (hipStream_t)stream;
    float* w2t = (float*)workspace;
    float* feat = (float*)((char*)workspace + ((kPolH2 * kPolIn2 * sizeof(float) + 1023) & ~(size_t)1023));
    const size_t chunk = P < (size_t)kPolicyChunk ? P : (size_t)kPolicyChunk;
    float* h1 = feat + chunk * kPolFeat;
    constexpr size_t fc1_lds = (size_t)2 * (128 + 128) * 33 * sizeof(float);       // 67,584 B
    if (allow_lds((const void*)policy_fc1_kernel, fc1_lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
    policy_transpose_kernel<<<(kPolH2 * kPolIn2 + 255) / 256, 256, 0, s>>>(w->fc2_w, w2t);
    for (size_t p0 = 0; p0 < P; p0 += chunk) {
        const int n = (int)(P - p0 < chunk ? P - p0 : chunk);
        policy_features_kernel<<<n, 256, 0, s>>>(ped_scans, (int)p0, n, w->cv1_w, w->cv1_b, w->cv2_w, w->cv2_b, feat);
        policy_fc1_kernel<<<dim3((n + 127) / 128, 2), 256, fc1_lds, s>>>(feat, n, w->fc1_w, w->fc1_b, h1);
        policy_head_kernel<<<n, 128, 0, s>>>(*c, *st, (int)p0, n, h1, w2t, *w, prev_actions, ped_cmd);
    }
    return launch_status();
}

int navsim_launch_order(const uint32_t* cost, int32_t* order, int32_t n, void* stream) {
    (void)hipGetLastError();
    if (!cost || !order || n < 0) return NAVSIM_E_ARG;
    if (n == 0) return NAVSIM_OK;
    launch_order_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(cost, order, n);
    return launch_status();
}

size_t navsim_step_workspace_bytes(const navsim_config* c) { return c ? workspace_bytes(c) : 0; }

int navsim_step(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    int rc = check_step_args(c, st, io, 0);
    if (rc != NAVSIM_OK) return rc;
    if (c->n_envs == 0) return NAVSIM_OK;
    return dispatch_step(c, st, io, 0, nullptr, (hipStream_t)stream);
}

int navsim_reset_obs(const navsim_config* c, const navsim_state* st, const navsim_step_io* io,
                     const uint8_t* mask, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    int rc = check_step_args(c, st, io, 1);
    if (rc != NAVSIM_OK) return rc;
    if (c->n_envs == 0) return NAVSIM_OK;
    return dispatch_step(c, st, io, 1, mask, (hipStream_t)stream);
}

const char* navsim_step_kernel_name(void) { return "navsim_step_kernel"; }

// test hook (declared in include/navsim.h under "test hooks")
int navsim_debug_math(int32_t fn, const double* x, const double* x2, double* out, int32_t n, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!x || !out || n < 0) return NAVSIM_E_ARG;
    if (n == 0) return NAVSIM_OK;
    math_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(fn, x, x2, out, n);
    return launch_status();
}

// text of the HIP error behind the last NAVSIM_E_LAUNCH on this thread
const char* navsim_last_hip_error(void) { return hipGetErrorString(g_last_hip_error); }

// microbenchmark hook, see gather_probe_kernel
int navsim_debug_gather(const float* x, uint64_t n_words, int32_t mode, int32_t iters, int32_t n_threads,
                        float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated earlier runtime calls
    if (!x || !out || n_threads <= 0) return NAVSIM_E_ARG;
    gather_probe_kernel<<<(n_threads + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, n_words, mode, iters, 12345, out);
    return launch_status();
}

// diagnostic build only: where the per-arena stamps go (NULL disables)
int navsim_debug_set_stamps(unsigned long long* buf) {
#ifdef NAVSIM_STAMPS
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? NAVSIM_OK : NAVSIM_E_LAUNCH;
#else
    (void)buf;
    return NAVSIM_E_UNSUPPORTED;
#endif
}

}  // extern "C"
