// kernels_step.hpp -- the fused step: scan, pedestrian merge, pedestrian phase, step kernel (rows a1, a2, a7, a10 SFM, a11, a16 pop).
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).

// ============================================================================================
// a1: the fused step.  One workgroup = one arena.
// ============================================================================================
struct StepShared {
    double rp[3];                 // robot pose being scanned
    double old_rp[3];             // robot pose at the start of the step (social force input)
    double act[2];                // action after the turning-radius clamp
    float lx, ly, lth;            // float32 lidar pose (env.py:386)
    int steps_now;                // steps[e] after this step's increment (done_steps: read in the reward block without a global
                                  // load; sits in what was padding -- the struct's size decides how many arenas share a CU)
    double cT, sT;                // cos / sin of (double)lth for the beam-table fast path
    int i0, j0;                   // integer ray origin (env.py:419)
    int nseg, ndisc;
    int rescan;
    int respawn;
    float t1;                     // march parameter after the shared first probe (origin cell)
    float r_all;                  // >= 0: every beam has this raw range (origin occupied / no march)
    unsigned long long step_key;  // scan-noise counter of this step
    unsigned long long rescan_key;  // ... of the second scan (crash revert: step_key + 1; restart: the new episode's reset key)
    int next_chunk;               // scan: next 64-beam chunk to hand to a wavefront (reset before every scan)
    int park_count, park_next;    // scan: parked rays (written / handed out), reset with next_chunk
    int pair_done;                // pedestrian phase: wavefronts that have written their share of the pair table (ped_pair_share)
    int term;                     // io.final_obs of an arena that restarts in this step: 1 = its rows are scan A's, 2 = a scan of their own
                                  // (crash: the re-scan at the reverted pose, env.py:707-723); sits in what was padding
    double wave_ratio[kMaxWaves];
};
// Pedestrian scratch of the pedestrian variants of the kernel, carved out of dynamic LDS behind the scan's
// dir / rng area and sized by cfg.max_peds (N), not by the compiled maximum: 144 N + 32 bytes, so that a
// 20-pedestrian world still fits 8 arenas per CU (the static 64-pedestrian layout allowed 7).
struct PedShared {
    double *ax, *ay, *avx, *avy;             // [N + 1] agent positions / velocities at time t (robot last)
    float (*seg)[4];                         // [4 N] rectangle edges seen by the lidar ...
    float (*disc)[2];                        // [2 N] ... leg discs (stored right behind seg)
    float* info;                             // [8 N] merge_prims_culled_core: beam-index interval of every primitive (prim_in_range)
};
__host__ __device__ inline size_t ped_lds_bytes(int N) { return (size_t)144 * N + 32; }
// the arena's table of social-force pair terms, N (N - 1) / 2 + N double2 (ped_pair_term); in the fused step it sits behind PedShared
__host__ __device__ inline size_t ped_pair_bytes(int N) { return ((size_t)(N * (N - 1) / 2 + N) * sizeof(double2) + 15) & ~(size_t)15; }
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

// direction of a beam at robot-frame angle lin seen from heading lth (env.py:388-390, 424) through its table entry
// (ct, st) = cos / sin(lin) and (cT, sT) = cos / sin(lth); false: the float32 rounding could not be proven, the caller
// evaluates beam_dir(heading).  heading = fl32(lin + lth) is returned either way.
__device__ __forceinline__ bool beam_dir_fast(double lin, double lth, double ct, double st, double cT, double sT,
                                              float& heading, float& dx, float& dy) {
    const double ang = lin + lth;
    heading = (float)ang;
    // heading = lin + lth + delta EXACTLY: (heading - ang) is exact (Sterbenz), and the rounding error of the
    // float64 sum is recovered by TwoSum
    const double bb = ang - lin;
    const double eps = (lin - (ang - bb)) + (lth - bb);
    const double delta = ((double)heading - ang) + eps;
    return nv::beam_dir_from_table(ct, st, cT, sT, delta, dx, dy);
}
// direction of beam k: table fast path with proven rounding, else full sincos
__device__ __forceinline__ void beam_dir_k(const navsim_config& c, const double* __restrict__ tab, int k,
                                           double step, double lth, double cT, double sT,
                                           float& dx, float& dy) {
#ifdef NAVSIM_DIAG_CHEAP_DIR     // diagnostic build only (wrong directions): what does the exact direction cost?
    { float h = (float)c.angle_min + (float)k * (float)step + (float)lth; dx = __cosf(h); dy = __sinf(h); return; }
#endif
    const double lin = nv::linspace_k(c, k, step);
    float heading = (float)(lin + lth);
    bool fast = false;
    if (tab) {
        const double2 cs = ((const double2*)tab)[k];
        fast = beam_dir_fast(lin, lth, cs.x, cs.y, cT, sT, heading, dx, dy);
    }
    if (!fast) nv::beam_dir(heading, dx, dy);
}

// the t = 0 sample of calc_range is the origin cell for every beam: take it once per scan
template <int RULE, typename Field>
__device__ __forceinline__ void first_probe(const Field& field, int i0, int j0, float max_range,
                                            float& t1, float& r_all) {
    t1 = 0.0f; r_all = -1.0f;
    typename Field::raw_t raw0 = field.load(i0, j0);               // origin is clipped into the map
    if (field.occupied(raw0)) { r_all = 0.0f; return; }            // sqrtf(0): starts inside an obstacle
    t1 = march_step<RULE>(field.decode(raw0, i0, j0));
    if (!(t1 < max_range)) r_all = max_range;
}

// Pedestrians into the scan (env.py:428-432), culled by bearing.  A rectangle side or a leg disc is
// seen under a small angle, so instead of testing every beam against every primitive (B x P ray
// tests: 3x the cost of the whole map march at 20 pedestrians) each wave takes one primitive, derives
// the beam-index interval that can possibly hit it (bearing +- half-width, two beams of margin, all
// three 2*pi aliases) and runs the SAME float32 seg_merge / circle_merge on those beams only;
// results land with an LDS atomicMin on the (non-negative) float bits, so they do not depend on the
// order of primitives.  rng[] holds metres, dir[] the beam directions.
// One lane per primitive, once per scan: can it change the clipped scan at all, and which beams can hit it?  Every
// point of a segment is at least |u| - |v - u| from the lidar (a disc: |u| - r); beyond the clip range a hit cannot
// matter, because clip(min(r, t)) = clip(r) for t >= range_max.  info[2 p], info[2 p + 1] = the beam-index interval
// [klo, khi] of the primitive's bearing (before the 2 pi aliases), empty (klo > khi) for a primitive out of range, all
// beams (klo = -inf) where a bearing is meaningless.  (Round 3: the eight lanes that share a primitive in the merge
// used to derive the interval themselves, two atan2f per primitive and ROUND of the merge instead of per primitive:
// 6 % of c3's vector instructions.)
constexpr float kPrimAllBeams = 1.0e30f;
// threads of the workgroup: the template argument, or (BLOCK = 0) read from the launch
template <int BLOCK> __device__ __forceinline__ int block_threads() { return BLOCK ? BLOCK : (int)blockDim.x; }
template <int BLOCK>
__device__ __forceinline__ void prim_in_range(int nprim, int nseg, float lx, float ly, float rcull, const Prims pr,
                                              float stepf, float beta0) {
    const float kTwoPiF = 6.2831853f;
    for (int p = (int)threadIdx.x; p < nprim; p += block_threads<BLOCK>()) {
        bool skip;
        float ac = 0.0f, w = 0.0f;
        bool full = (stepf <= 0.0f);
        if (p < nseg) {
            float ux = pr.seg[p][0] - lx, uy = pr.seg[p][1] - ly, vx = pr.seg[p][2] - lx, vy = pr.seg[p][3] - ly;
            skip = sqrtf(ux * ux + uy * uy) - sqrtf((vx - ux) * (vx - ux) + (vy - uy) * (vy - uy)) > rcull;
            float b1 = atan2f(uy, ux), b2 = atan2f(vy, vx);
            float d = b2 - b1;
            d -= kTwoPiF * floorf(d / kTwoPiF + 0.5f);              // (-pi, pi]
            ac = b1 + 0.5f * d;
            w = 0.5f * fabsf(d);
            if (fabsf(d) > 3.0f || ux * ux + uy * uy < 1e-6f || vx * vx + vy * vy < 1e-6f) full = true;
        } else {
            float ux = pr.disc[p - nseg][0] - lx, uy = pr.disc[p - nseg][1] - ly;
            float dist = sqrtf(ux * ux + uy * uy);
            skip = dist - nv::kLegRadius > rcull;
            if (dist <= nv::kLegRadius * 1.05f) full = true;
            else { ac = atan2f(uy, ux); w = asinf(fminf(1.0f, nv::kLegRadius / dist)); }
        }
        float rel = ac - beta0;
        rel -= kTwoPiF * floorf(rel / kTwoPiF + 0.5f);              // [-pi, pi)
        float klo = (rel - w) / stepf - 2.0f, khi = (rel + w) / stepf + 2.0f;   // two beams of margin
        if (full) { klo = -kPrimAllBeams; khi = kPrimAllBeams; }
        if (skip) { klo = kPrimAllBeams; khi = -kPrimAllBeams; }
        pr.info[2 * p] = klo;
        pr.info[2 * p + 1] = khi;
    }
}

template <int BLOCK>
__device__ __forceinline__ void merge_prims_culled_core(int B, float lx, float ly, float stepf,
                                                        int nseg, int ndisc, const Prims pr,
                                                        const float2* __restrict__ dir, float* __restrict__ rng) {
    const int lane = (int)threadIdx.x & 63;
    const float kTwoPiF = 6.2831853f;
    const float Kf = (stepf > 0.0f) ? kTwoPiF / stepf : 0.0f;
    const int nprim = nseg + ndisc;
    // eight lanes per primitive, eight primitives per wavefront at a time (a pedestrian a few metres away
    // spans 10-50 beams; measured 4 / 8 / 16 / 32 / 64 lanes: c3 11.81 / 11.80 / 11.68 / 11.13 / 10.26 M env-steps/s)
#ifndef NAVSIM_MERGE_G
#define NAVSIM_MERGE_G 8
#endif
    constexpr int G = NAVSIM_MERGE_G;
    const int sub = lane & (G - 1);
    // aliases of the bearing: rel + w <= pi + 1.6 < 2 pi - 2 beams, so rel - 2 pi (m = -1) lies below beam 0 unless the
    // beams are very few
    const int m_first = (kTwoPiF - 2.0f * stepf > 4.8f) ? 0 : -1;
    for (int p = ((int)threadIdx.x) / G; p < nprim; p += block_threads<BLOCK>() / G) {
        const float klo = pr.info[2 * p], khi = pr.info[2 * p + 1];  // prim_in_range ran before the scan (prims_prepare)
        if (klo > khi) continue;                                    // beyond the clip range
        const bool full = klo < -0.5f * kPrimAllBeams;
        const bool is_seg = p < nseg;
        float a0, a1, a2 = 0.f, a3 = 0.f;
        if (is_seg) { a0 = pr.seg[p][0]; a1 = pr.seg[p][1]; a2 = pr.seg[p][2]; a3 = pr.seg[p][3]; }
        else { a0 = pr.disc[p - nseg][0]; a1 = pr.disc[p - nseg][1]; }
        for (int m = full ? 0 : m_first; m <= (full ? 0 : 1); ++m) {
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

// which primitives can matter, and for which beams (info[]): needs only the primitives and the lidar pose, so it runs
// BEFORE the scan and is covered by the barrier that ends the scan
__host__ __device__ inline float prim_cull_range(double range_max) { return (float)range_max * 1.0001f + 0.01f; }
template <int BLOCK>
__device__ __forceinline__ void prims_prepare(const navsim_config& c, const StepShared& sh, const Prims pr) {
    prim_in_range<BLOCK>(sh.nseg + sh.ndisc, sh.nseg, sh.lx, sh.ly, prim_cull_range(c.range_max), pr,
                         (float)nv::linspace_step(c), (float)(c.angle_min + (double)sh.lth));
}
template <int BLOCK>
__device__ __forceinline__ void merge_prims_culled(const navsim_config& c, const StepShared& sh, const Prims pr,
                                                   const float2* __restrict__ dir, float* __restrict__ rng) {
    merge_prims_culled_core<BLOCK>(c.n_beams, sh.lx, sh.ly, (float)nv::linspace_step(c), sh.nseg, sh.ndisc, pr, dir, rng);
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
    const uint64_t nkey = (noise_std > 0.0f) ? nv::noise_stream(c.seed, genv, noise_key) : 0;
    // rng[] holds METRES (the scan stored range * resolution as each ray finished): round 3 dropped the conversion pass
    // and its barrier; prim_in_range has run before the scan
    const bool culled = dir && rng_rw && (nseg | ndisc);           // LDS-resident: bearing-culled merge
    (void)r_all; (void)res;
    if (culled) {
#ifndef NAVSIM_DIAG_NO_MERGE          // diagnostic build only (pedestrians invisible): what does the merge cost?
        merge_prims_culled<BLOCK>(c, sh, pr, dir, rng_rw);
#endif
        __syncthreads();
    }
    for (int k = (int)threadIdx.x; k < B; k += BLOCK) {
        float rr = rng[k];
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
            rr = rr + noise_std * nv::gauss_noise(nkey, (uint32_t)k);
        cr |= (rr < thr[k]);
        dc |= (rr < dthr[k]);
        obs_row[(size_t)(S - 1) * B + k] = rr;
        for (int j = 0; j < S - 1; ++j)
            if (S - 1 - j > n_hist) obs_row[(size_t)j * B + k] = rr;
    }
    crash = cr;
    discomfort = dc;
}

// ---- wave masks as scalars.  The march keeps its per-lane flags (still marching / hit) as 64-bit lane masks in
// scalar registers and combines them with scalar instructions: a comparison writes its mask straight to an SGPR
// pair, the loop ends on s_cmp_lg_u64, and a value is picked per lane by v_cndmask on the mask.  (Written with
// `bool`s the compiler kept the flags as masks too, but rebuilt them through v_cndmask 0/1 + v_cmp at every
// ballot: 4 of the ~50 vector instructions of a probe, and twice the scalar bookkeeping.)
typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t mask_of(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ lanemask_t mask_ult(unsigned a, unsigned b) { return __builtin_amdgcn_uicmp(a, b, 36); }   // ICMP_ULT
__device__ __forceinline__ lanemask_t mask_eq(unsigned a, unsigned b) { return __builtin_amdgcn_uicmp(a, b, 32); }    // ICMP_EQ
__device__ __forceinline__ lanemask_t mask_flt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }         // FCMP_OLT
// lane bit of m set ? a : b.  Pass masks that a SCALAR instruction produced (an and / or / andn2 of comparison
// masks, as everywhere below): the compiler's hazard recognizer does not look into the asm, and a v_cndmask that
// reads an SGPR pair written by the vector instruction right before it needs wait states on gfx950.
__device__ __forceinline__ float mask_sel(lanemask_t m, float a, float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ int mask_sel(lanemask_t m, int a, int b) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ bool mask_lane(lanemask_t m) { return mask_sel(m, 1, 0) != 0; }

// One probe of calc_range (env.py:425) for every lane of `active`: sample position, distance there, hit test, step.
// Lanes that are finished or outside the map run along with their updates masked off.  A lane that hits keeps the
// t of its hit probe (the hit cell is recomputed from it after the march).
// RECT: 0 = the field alone; 1 = the tile's two-rectangle record from global memory; 2 = from the INDEX form of the arena's
// table staged in LDS (kernels_rect.hpp: `rects` then points at the LDS copy of the arena's row -- list[256], then two
// list indices per tile; round 4.  Round 3 staged the 16-byte records themselves, 63.5 KB per arena: two workgroups per CU).
template <typename Field, int RULE, int RECT>
__device__ __forceinline__ void probe_round(const Field& field, const char* __restrict__ rects, unsigned tpr,
                                            float x0, float y0, float dx, float dy, unsigned uW, unsigned uH,
                                            float max_range, float& t, lanemask_t& active, lanemask_t& hit) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));            // v_pk_mul_f32 + v_pk_add_f32 (never fused:
    const f32x2 org = {x0, y0}, dir = {dx, dy};                          // the translation unit is -ffp-contract=off)
    f32x2 pos;
    if constexpr (RULE == NAVSIM_MARCH_F32_FMA) pos = __builtin_elementwise_fma(dir, (f32x2){t, t}, org);   // v_pk_fma_f32
    else pos = org + dir * t;
    int px = (int)pos.x, py = (int)pos.y;
    // RECT == 2 runs on CLOSED maps only (cfg.closed_maps: a ring of at least 3 occupied cells around every map, verified by
    // navsim_build_rect_index / true of every map navsim_regen draws): a marching ray then never leaves the map -- from a
    // free cell c at distance d of the nearest obstacle the next sample lies within d (1 + 1e-6) + 1 cells of c's corner, and
    // the ring's inner layer is an obstacle at distance >= d, so the sample is at worst one cell inside the ring -- and a
    // finished lane stays where it stopped.  No bounds test, no select of the tile offset: 3 of a round's 41 vector
    // instructions (c2 +4 %, profiles/r04_idx/ab_closed.txt).
    const lanemask_t live = (RECT == 2) ? active : (active & mask_ult((unsigned)px, uW) & mask_ult((unsigned)py, uH));
    lanemask_t occ;
    float d;
    if constexpr (RECT == 0) {
        px = mask_sel(live, px, 0);
        py = mask_sel(live, py, 0);
    }
    if constexpr (RECT != 0) {
        // the tile's two-rectangle record (kernels_rect.hpp): exact integer d2 without touching the field; the
        // rare probe in a tile without a valid record reads the field.  32-bit lane offset on a uniform base.
        // tile rows and tiles per row stay far below 2^24: the 24-bit multiply-add is a full-rate instruction
        // a lane that is not live keeps its out-of-map px, py (every result of it is masked) and reads record 0
        const unsigned cell = rect_cell(px, py);
        // (the tile index from the packed cell by v_pk_lshrrev_b16 + v_dot2_u32_u16, one instruction fewer: measured -0.7 %)
        unsigned tile = __umul24((unsigned)py >> kRectShift, tpr) + ((unsigned)px >> kRectShift);
        if constexpr (RECT != 2) tile = (unsigned)mask_sel(live, (int)tile, 0);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 rec;
        lanemask_t inval;
        if constexpr (RECT == 2) {
            // the arena's row in LDS: ds_read_u16 of the tile's index pair, then the two rectangles (ds_read_b64 each; most
            // lanes of a wavefront name the same few list entries: broadcast reads)
            // The row starts the dynamic LDS, so every address below is a register plus an immediate.  The two indices are
            // read as bytes (ds_read_u8 x2: an index then needs one shift to become its rectangle's address; unpacking a
            // ds_read_u16 took five vector instructions -- the LDS port is idle, the vector unit is what a round waits for).
            // (the compiler fuses the two byte loads into one ds_read_u16 and unpacks it with and / shift / two shift-adds: four
            // instructions; written out as two ds_read_u8 it is three, measured +1 % on c2 and -0.6 % on c4: left to the compiler)
            typedef __attribute__((address_space(3))) const unsigned char lds_u8;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) const u32x2 lds_u2;
            lds_u8* row = (lds_u8*)rects;
            const unsigned ia = row[kRectListLen * 8 + tile * 2u], ib = row[kRectListLen * 8 + 1 + tile * 2u];
            const u32x2 ra = *(lds_u2*)(row + ia * 8u);
            const u32x2 rb = *(lds_u2*)(row + ib * 8u);
            rec.x = ra.x; rec.y = ra.y; rec.z = rb.x; rec.w = rb.y;
            inval = live & mask_eq(ia, 0xFFu);                       // no index: both bytes are 0xFF
        } else {
            // The load is written out: after a compiler-generated global_load_dwordx4 the register allocator moved three
            // of the four loaded dwords to other registers before using them (3 of 42 vector instructions per probe).
            // The base goes through an s_mov inside the statement: when the register allocator has spilled `rects` to
            // VGPR lanes it reloads it with v_readlane RIGHT in front of this statement, and a VMEM instruction may not read
            // an SGPR within 5 wait states of the VALU instruction that wrote it -- a hazard the compiler pads for in its
            // own code and cannot see inside inline assembly (round 4: the 64-thread pedestrian variant loaded from a stale
            // base and faulted after an unrelated edit moved its spills).  SALU reads of such an SGPR are interlocked.
            const unsigned off = tile * (unsigned)sizeof(uint4);
            unsigned long long base;
            asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(rec), "=&s"(base) : "v"(off), "s"(rects) : "memory");
            inval = live & mask_eq(rec.x & 0xFFFFu, (unsigned)kRectInvalid);
        }
        const int da = rect_dist2(rec.x, rec.y, cell), db = rect_dist2(rec.z, rec.w, cell);
        const int d2 = da < db ? da : db;
        occ = live & mask_eq((unsigned)d2, 0u);
        d = Field::sqrt_d2(d2);
        if (inval != 0) {                                               // wave-uniform branch
            // Only the load is per-lane control flow: a lane mask assigned under a divergent branch would stop
            // being one value per wave.  Lanes with a valid record decode a dummy distance of one cell.
            typename Field::raw_t raw = (typename Field::raw_t)1;
            if (mask_lane(inval)) raw = field.load(px, py);
            occ = (occ & ~inval) | (inval & mask_of(field.occupied(raw)));
            d = mask_sel(inval, field.decode_nz(raw, px, py), d);
        }
    } else {
        typename Field::raw_t raw = field.load(px, py);
        occ = live & mask_of(field.occupied(raw));
        d = field.decode_nz(raw, px, py);
    }
    hit |= occ;
    const float tn = t + march_step<RULE>(d);
    const lanemask_t go = live & ~occ;
    t = mask_sel(go, tn, t);
    active = go & mask_flt(tn, max_range);
}

// raw range (cells) of a finished ray: the hit cell recomputed from the t of the hit probe
template <int RULE>
__device__ __forceinline__ float ray_result(lanemask_t hit, float x0, float y0, float dx, float dy, float t, float miss) {
    const float xd = (float)(int)march_pos<RULE>(x0, dx, t) - x0;
    const float yd = (float)(int)march_pos<RULE>(y0, dy, t) - y0;
    return mask_sel(hit, sqrtf(xd * xd + yd * yd), miss);
}

// Parking.  A wavefront's 64 adjacent beams need 7.9 probes on average and 13.5 for the slowest (c2): the last
// rounds of a chunk run for a handful of lanes.  So a wavefront leaves a chunk when at most kParkLanes rays are still
// marching, parks those (beam, t, direction: 16 bytes in LDS) and goes on; after the chunks the parked rays -- about
// 70 per scan -- are marched 64 at a time.  Which wavefront finishes a ray, and when, changes no result.  On the c2
// bench state (profiles/_diag/park_model.py) 0.84 of the probe rounds remain at 8 lanes and 0.82 at 16; a group of
// parked rays points everywhere, so its record loads do not coalesce and the kernel gains less than that: c2 +4.6 % at
// 16 lanes (+1 % at 8, 24 = 16, 32 +1 %).  A launch that is a single generation of workgroups lasts as long as its slowest
// workgroup and loses to the second pass (c4 -8 %, c5 -4 %): parking is compiled into the 256-thread kernels only.  The
// pedestrian variants park since the end of round 3 (park_lds_bytes: the parked ray lives in its own rng / dir slots):
// c3 23.2 -> 24.0 M at 16 lanes (8: 23.6; 24 and 32, which cost the eighth arena of a CU its LDS: 23.2, 23.0).
#ifndef NAVSIM_PARK_LANES
#define NAVSIM_PARK_LANES 16
#endif
#ifndef NAVSIM_PAIR_SHARE_MIN_BLOCK       // threads per arena from which the social force's pair terms are shared by four wavefronts
#define NAVSIM_PAIR_SHARE_MIN_BLOCK 512
#endif
constexpr int kParkLanesMax = NAVSIM_PARK_LANES;
#ifndef NAVSIM_PARK_LANES_PEDS
#define NAVSIM_PARK_LANES_PEDS 16
#endif
#ifdef NAVSIM_PARK_512           // experiment, not kept (profiles/r04_blocks/ab_park512.txt: c2 at 768-2048 arenas and c4 -3..-5 %): parked rays in the 512-thread kernels too
__host__ __device__ constexpr bool step_parks(int block, bool peds) { return (block == 256 && (!peds || NAVSIM_PARK_LANES_PEDS > 0)) || (block == 512 && !peds); }
#else
__host__ __device__ constexpr bool step_parks(int block, bool peds) { return block == 256 && (!peds || NAVSIM_PARK_LANES_PEDS > 0); }
#endif
// A parked ray is (beam, t, direction).  Without pedestrians: 16 bytes in the park area.  In the pedestrian variants the
// ray's own slots of the LDS copy -- rng[k], dir[k], which it fills only when it finishes -- hold t and the direction
// meanwhile, and the park area keeps the beam index alone (2 bytes): 0.5 KB instead of 4.3, which is what keeps eight
// arenas on a CU (round 2 measured parking in these variants with the 16-byte records: -3 %, for the lost residency).
__host__ __device__ inline size_t park_lds_bytes(int B, int park_lanes, bool peds) {
    return (((size_t)((B + 63) / 64) * park_lanes * (peds ? 2 : 16)) + 15) & ~(size_t)15;
}
__device__ __forceinline__ int lanes_below(lanemask_t m) {          // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Predicated one-ray-per-lane scan: the march loop has ONE wave-level branch
// (any lane still marching?) instead of a divergent if-ladder per probe; finished or out-of-map
// lanes keep executing with their updates masked off.
//
// Beams are handed out in chunks of 64 adjacent beams, one chunk per wavefront at a time, from a counter in LDS:
// 1081 beams are 16.9 chunks, so a static `k += BLOCK` walk gives wavefront 0 of every 256-thread workgroup five
// chunks and the others four, and the workgroup lives as long as its slowest wavefront.  Which wavefront
// marches a beam changes no result.  The caller zeroes sh.next_chunk behind a barrier before every scan.
template <int BLOCK, typename Field, bool TO_LDS, int RULE, int RECT>
__device__ __forceinline__ void scan_beams_pred(const navsim_config& c, StepShared& sh,
                                                const Field& field, const uint4* __restrict__ rects,
                                                const double* __restrict__ tab,
                                                const Prims pr, float2* __restrict__ dir_lds, float* __restrict__ rng_lds,
                                                float4* __restrict__ park, const int park_lanes,
                                                const float* __restrict__ thr, const float* __restrict__ dthr,
                                                float* __restrict__ obs_row, int n_hist, float noise_std,
                                                uint64_t noise_key, uint64_t genv,
                                                int& crash, int& discomfort, bool prepare_prims = true) {
    const int B = c.n_beams, S = c.n_scan_stack, H = c.map_h, W = c.map_w;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const float res = (float)c.resolution;
    const float rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    const float x0 = (float)sh.i0, y0 = (float)sh.j0;
    const float lx = sh.lx, ly = sh.ly;
    const int nseg = sh.nseg, ndisc = sh.ndisc;
    const unsigned uW = (unsigned)W, uH = (unsigned)H;
    const unsigned tpr = (unsigned)((W + 7) >> kRectShift);
    const float t1 = sh.t1, r_all = sh.r_all;
    const uint64_t nkey = (noise_std > 0.0f) ? nv::noise_stream(c.seed, genv, noise_key) : 0;
    int cr = 0, dc = 0;
    // what happens to a finished ray (identical for every schedule)
    auto finish = [&](int k, float dx, float dy, float rr) {
        if (TO_LDS) {                                           // pedestrians: culled merge on the LDS copy, in metres
            rng_lds[k] = rr * res;                              // env.py:426 (rr is r_all itself where the origin is occupied)
            dir_lds[k] = make_float2(dx, dy);
            return;
        }
        rr = rr * res;                                          // env.py:426
        for (int p = 0; p < nseg; ++p)
            nv::seg_merge(rr, lx, ly, dx, dy, pr.seg[p][0], pr.seg[p][1], pr.seg[p][2], pr.seg[p][3]);
        for (int p = 0; p < ndisc; ++p)
            nv::circle_merge(rr, lx, ly, dx, dy, pr.disc[p][0], pr.disc[p][1], nv::kLegRadius);
        rr = rr < 0.0f ? 0.0f : rr;                             // env.py:435
        rr = rr > rmax ? rmax : rr;
        if (noise_std > 0.0f && rr != rmax)                     // env.py:437-440
            rr = rr + noise_std * nv::gauss_noise(nkey, (uint32_t)k);
        cr |= (rr < thr[k]);
        dc |= (rr < dthr[k]);
        obs_row[(size_t)(S - 1) * B + k] = rr;
        for (int j = 0; j < S - 1; ++j)
            if (S - 1 - j > n_hist) obs_row[(size_t)j * B + k] = rr;
    };
    const float miss = (r_all >= 0.0f) ? r_all : max_range;
    // info[] of the culled merge: visible after the scan's barrier (the step's first scan: wavefront 0 has written it)
    if constexpr (TO_LDS) { if (prepare_prims) prims_prepare<BLOCK>(c, sh, pr); }
    constexpr bool kPark = step_parks(BLOCK, TO_LDS);           // compiled in only where the host ever asks for it
    int own_chunk = 0;                                    // one wavefront per arena: no counter needed
    const int lane = (int)threadIdx.x & 63;
    for (;;) {
        int chunk = 0;
        if (BLOCK > 64) {
            if (lane == 0) chunk = atomicAdd(&sh.next_chunk, 1);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
        } else {
            chunk = own_chunk++;
        }
        if (chunk * 64 >= B) break;
        const int k = chunk * 64 + lane;
        const bool valid = k < B;
        float dx, dy;
        beam_dir_k(c, tab, valid ? k : B - 1, step, (double)sh.lth, sh.cT, sh.sT, dx, dy);
        float t = t1;
        lanemask_t active = mask_of(valid & (r_all < 0.0f));
        lanemask_t hit = 0;
        if constexpr (!kPark) {
            while (active != 0)
                probe_round<Field, RULE, RECT>(field, (const char*)rects, tpr, x0, y0, dx, dy, uW, uH, max_range, t, active, hit);
            if (valid) finish(k, dx, dy, ray_result<RULE>(hit, x0, y0, dx, dy, t, miss));
        } else {
            while ((int)__builtin_popcountll(active) > park_lanes)
                probe_round<Field, RULE, RECT>(field, (const char*)rects, tpr, x0, y0, dx, dy, uW, uH, max_range, t, active, hit);
            const bool marching = mask_lane(active);
            if (active != 0) {                                   // park what is still marching
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh.park_count, (int)__builtin_popcountll(active));
                base = __builtin_amdgcn_readfirstlane(base);
                if (marching) {
                    if constexpr (TO_LDS) {
                        ((unsigned short*)park)[base + lanes_below(active)] = (unsigned short)k;
                        rng_lds[k] = t;
                        dir_lds[k] = make_float2(dx, dy);
                    } else {
                        park[base + lanes_below(active)] = make_float4(__int_as_float(k), t, dx, dy);
                    }
                }
            }
            if (valid & !marching) finish(k, dx, dy, ray_result<RULE>(hit, x0, y0, dx, dy, t, miss));
        }
    }
    // the parked rays, 64 at a time
    if constexpr (kPark) __syncthreads();
    const int n_parked = kPark ? sh.park_count : 0;
    for (;;) {
        if (!kPark || n_parked == 0) break;
        int g = 0;
        if (lane == 0) g = atomicAdd(&sh.park_next, 64);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= n_parked) break;
        const bool valid = g + lane < n_parked;
        int k;
        float t, dx, dy;
        if constexpr (TO_LDS) {
            k = (int)((const unsigned short*)park)[valid ? g + lane : g];
            t = rng_lds[k];
            const float2 d = dir_lds[k];
            dx = d.x; dy = d.y;
        } else {
            const float4 r = park[valid ? g + lane : g];
            k = __float_as_int(r.x);
            t = r.y; dx = r.z; dy = r.w;
        }
        lanemask_t active = mask_of(valid);
        lanemask_t hit = 0;
        while (active != 0)
            probe_round<Field, RULE, RECT>(field, (const char*)rects, tpr, x0, y0, dx, dy, uW, uH, max_range, t, active, hit);
        if (valid) finish(k, dx, dy, ray_result<RULE>(hit, x0, y0, dx, dy, t, miss));
    }
    if (TO_LDS) {
        __syncthreads();
        finish_beams<BLOCK>(c, sh, pr, tab, dir_lds, rng_lds, rng_lds, thr, dthr, obs_row, n_hist, noise_std,
                            noise_key, genv, cr, dc);
    }
    crash = cr;
    discomfort = dc;
}

// ---- pieces of phase 1 for ONE pedestrian (env.py:617-693 with the build-defined social force or external commands),
// shared by the fused step kernel (pedestrian i on thread i of the arena's workgroup) and by ped_update_kernel (a pack
// of arenas per workgroup).  Same functions, same operation order: same results.
// waypoint pop (env.py:633-642): the current waypoint is wp[head], a pop advances the head (ABI 5: the list is read-only between
// two plans; rounds 1-4 shifted the whole remaining list through global memory on this lane, up to 63 double2 moves per pop on
// the wavefront the c3 / c5 launches wait for).  wp = the pedestrian's row, nw = waypoints stored; returns the new head
__device__ __forceinline__ int ped_pop_waypoints(const double* wp, int head, int nw, const double (&pp)[3]) {
    while (head + 1 < nw) {
        double ddx = pp[0] - wp[2 * head], ddy = pp[1] - wp[2 * head + 1];
        if (sqrt(ddx * ddx + ddy * ddy) < 1.0) ++head;
        else break;
    }
    return head;
}
// the (i, j) of pair term t of an arena with n pedestrians: the strict upper triangle row by row (n (n - 1) / 2 terms
// between pedestrians, each evaluated once), then the robot (index n) acting on pedestrian t - n_pp
__device__ __forceinline__ void ped_pair_index(int t, int n, int& i, int& j) {
    const int n_pp = n * (n - 1) / 2;
    if (t < n_pp) {
        i = 0;
        int rem = t;
        while (rem >= n - 1 - i) { rem -= n - 1 - i; ++i; }
        j = i + 1 + rem;
    } else {
        i = t - n_pp; j = n;
    }
}
// pair term t of an arena into its table of n (n - 1) / 2 + n terms (slot t = term t): the term on pedestrian j from
// pedestrian i is EXACTLY minus the term on i from j (every operand of sfm_pair changes sign or stays -- differences, their
// squares, quotients, the odd atan2 of two sign-symmetric products -- and round-to-nearest is symmetric under negation), so
// each unordered pedestrian pair is evaluated and stored once and read twice (ped_sfm_step subtracts for the second
// reader); the robot (index n) only acts, it receives nothing.  (Round 3: the table was [n][n + 1] with both signs stored;
// half of it keeps eight 256-thread arenas on a CU once the fused kernel carries the table.)
__device__ __forceinline__ void ped_pair_term(const navsim_config& c, const PedShared& ps, double2* pair, int n, int t) {
    int i, j;
    ped_pair_index(t, n, i, j);
    double fx, fy;
    sfm_pair(c, ps.ax[i], ps.ay[i], ps.avx[i], ps.avy[i], ps.ax[j], ps.ay[j], ps.avx[j], ps.avy[j], fx, fy);
    pair[t] = make_double2(fx, fy);
}
// forces on pedestrian i and its semi-implicit Euler step: desired + social (its row of the pair
// table in partner order; a second, inlined sfm_pair in this loop cost the fused kernel 100 bytes of spills) + obstacle
// (distance-field gradient)
template <typename Field>
__device__ __forceinline__ void ped_sfm_step(const navsim_config& c, const Field& field, const PedShared& ps,
                                             const double2* pair, int n, int i, double vpref, const double* wp, double dt,
                                             double (&pp)[3], double (&pvel)[2]) {
    double ex = wp[0] - ps.ax[i], ey = wp[1] - ps.ay[i];
    double L = sqrt(ex * ex + ey * ey);
    if (L > 1e-9) { ex = ex / L; ey = ey / L; } else { ex = 0.0; ey = 0.0; }
    double fdx = (vpref * ex - ps.avx[i]) / c.sfm_tau;
    double fdy = (vpref * ey - ps.avy[i]) / c.sfm_tau;
    double fsx = 0.0, fsy = 0.0;
    // the terms on pedestrian i in partner order j = 0 .. n (ped_pair_term's slots): partners below i hold the pair in
    // THEIR row, with the opposite sign (x - y is x + (-y) exactly); then its own row; then the robot
    {
        int slot = i - 1;                                   // (0, i)
        for (int j = 0; j < i; ++j) {
            const double2 f = pair[slot];
            fsx -= f.x;
            fsy -= f.y;
            slot += n - 2 - j;                              // (j + 1, i)
        }
        slot = i * (n - 1) - i * (i - 1) / 2;               // (i, i + 1)
        for (int j = i + 1; j < n; ++j, ++slot) {
            const double2 f = pair[slot];
            fsx += f.x;
            fsy += f.y;
        }
        const double2 f = pair[n * (n - 1) / 2 + i];
        fsx += f.x;
        fsy += f.y;
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
// new goal at the final waypoint (env.py:667-680: table draw, or wait for navsim_replan), leg odometry and the
// pedestrian's obs yaw (env.py:683-693), state.  wp = the pedestrian's row of waypoints, head = its current one, nw = stored.
// Returns whether the pedestrian stands within 0.5 m of its final waypoint (what navsim_replan serves: st.ped_due).
__device__ __forceinline__ bool ped_finish(const navsim_config& c, const navsim_state& st, int e, int i, size_t pq, double* wp,
                                           int head, int nw, double dt, uint64_t genv, uint64_t steps_now, const double (&pp)[3],
                                           const double (&pvel)[2]) {
    double ddx = pp[0] - wp[2 * (nw - 1)], ddy = pp[1] - wp[2 * (nw - 1) + 1];
    const bool at_final = sqrt(ddx * ddx + ddy * ddy) < 0.5;
    if (at_final && c.n_spawn > 0 && st.spawn_pose && !st.costmap) {
        uint64_t h = nv::hash4(c.seed, genv, (uint64_t)i + 1000, steps_now);
        for (int tries = 0; tries < c.n_spawn; ++tries) {
            int idx = (int)((h + (uint64_t)tries) % (uint64_t)c.n_spawn);
            const double* cand = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
            double gx = cand[0] - pp[0], gy = cand[1] - pp[1];
            if (sqrt(gx * gx + gy * gy) > 10.0) {
                wp[0] = cand[0]; wp[1] = cand[1]; head = 0;
                st.ped_n_waypoints[pq] = 1;
                if (st.ped_goal) { st.ped_goal[pq * 2] = cand[0]; st.ped_goal[pq * 2 + 1] = cand[1]; }
                break;
            }
        }
    }
    st.ped_wp_head[pq] = head;
    double dist[3] = {st.ped_dist[pq * 3], st.ped_dist[pq * 3 + 1], st.ped_dist[pq * 3 + 2]};
    nv::leg_odometry(pp, pvel, st.ped_prev_yaw[pq], dt, dist);
    st.ped_dist[pq * 3] = dist[0]; st.ped_dist[pq * 3 + 1] = dist[1]; st.ped_dist[pq * 3 + 2] = dist[2];
    st.ped_prev_yaw[pq] = nv::wrap_pi(pp[2]);
    st.ped_pose[pq * 3] = pp[0]; st.ped_pose[pq * 3 + 1] = pp[1]; st.ped_pose[pq * 3 + 2] = pp[2];
    st.ped_vel[pq * 2] = pvel[0]; st.ped_vel[pq * 2 + 1] = pvel[1];
    return at_final;
}

// LDS written by some lanes of a wavefront and read by others of the SAME wavefront: no workgroup barrier (the other
// wavefronts are elsewhere), only the order of the wavefront's own LDS accesses, which the fences pin for the compiler
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Phase 1 for the pedestrians of one arena inside the fused step kernel: run by wavefront 0 ALONE (pedestrian i on its
// lane i, N <= 64), while the arena's other wavefronts already march the static map -- nothing of the scan before its
// merge reads a pedestrian, and nothing here reads the robot's new pose.  Round 3, c5: the phase used to hold the other
// 15 wavefronts of a 1024-thread workgroup at a barrier for 15 of the workgroup's 41 us.  The pair terms (n (n - 1) / 2
// + n, independent) are spread over the 64 lanes into the arena's pair table (LDS behind PedShared), then every
// pedestrian adds its row in partner order -- the same sums, in the same order, as a sequential loop.
// (a) before the workgroup's first barrier -- beside the robot's step on another wavefront: waypoint pop, every agent's
// position / velocity at time t staged (pedestrians, then the robot from its global state, which nobody writes before
// the end of the kernel); returns the index of the current waypoint after the pop, nw = the waypoints stored
__device__ __forceinline__ int ped_stage_wave(const navsim_config& c, const navsim_state& st, int n, int lane, bool is_ped,
                                              size_t pq, const double* rp_t, double prev_v, const PedShared& ps,
                                              const double (&pp)[3], const double (&pvel)[2], int& nw) {
    const int P = c.max_waypoints;
    const double* wp = st.ped_waypoints + (pq * P) * 2;
    int head = 0;
    nw = 1;
    if (is_ped) {
        nw = st.ped_n_waypoints[pq];
        head = ped_pop_waypoints(wp, st.ped_wp_head[pq], nw, pp);
    }
    if (c.ped_model == NAVSIM_PED_SFM) {
        if (is_ped) { ps.ax[lane] = pp[0]; ps.ay[lane] = pp[1]; ps.avx[lane] = pvel[0]; ps.avy[lane] = pvel[1]; }
        if (lane == 0) {
            double s, cs;
            nv::sincos(rp_t[2], s, cs);
            ps.ax[n] = rp_t[0]; ps.ay[n] = rp_t[1];
            ps.avx[n] = prev_v * cs; ps.avy[n] = prev_v * s;
        }
    }
    return head;
}
// The pair terms shared by the first `waves` wavefronts of the workgroup (1024-thread variants, round 6): twenty pedestrians are
// 210 independent terms of ~700 dependent float64 instructions each -- four rounds on wavefront 0's 64 lanes, 12.6 of the 21 us
// its chain takes while fifteen wavefronts that finish their scan in 8 us wait for it (profiles/r03_c5_chain/ped_chain.txt); on
// four wavefronts it is one round.  Each writes its share of the table, signals through sh.pair_done and leaves for the scan.
// (Round 3 measured this form at +-0: wavefront 0's remaining chain then ran slower beside the marching wavefronts -- since round
// 6 it runs at priority 3.)
__device__ __forceinline__ void ped_pair_share(const navsim_config& c, const PedShared& ps, double2* pair, int n, int tid, int waves,
                                               int* pair_done) {
    const int n_terms = n * (n - 1) / 2 + n;
    for (int t = tid; t < n_terms; t += waves * 64) ped_pair_term(c, ps, pair, n, t);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((tid & 63) == 0) __hip_atomic_fetch_add(pair_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (b) behind it: pair terms, forces and the Euler step, new goals, leg odometry, state
template <typename Field>
__device__ __forceinline__ void ped_advance_wave(const navsim_config& c, const navsim_state& st, const Field& field, int e,
                                                 int n, int lane, bool is_ped, size_t pq, double dt, uint64_t genv,
                                                 uint64_t steps_now, const PedShared& ps, double2* pair, int head, int nw,
                                                 double (&pp)[3], double (&pvel)[2], int* pair_done = nullptr, int pair_waves = 1) {
    const int P = c.max_waypoints;
    double* wp = st.ped_waypoints + (pq * P) * 2;
    if (c.ped_model == NAVSIM_PED_SFM) {
        if (pair_waves > 1) {
            // the pair terms were shared out (ped_pair_share): wait for the other wavefronts' parts -- every wavefront of a workgroup
            // is resident and writes its share right behind the workgroup's first barrier: a wait of one pair term's length
            while (__hip_atomic_load(pair_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < pair_waves) __builtin_amdgcn_s_sleep(1);
        } else {
            wave_lds_sync();                                // the staged agents (the workgroup barrier lies in between as well)
            const int n_terms = n * (n - 1) / 2 + n;
            for (int t = lane; t < n_terms; t += 64) ped_pair_term(c, ps, pair, n, t);
        }
        wave_lds_sync();
        if (is_ped) ped_sfm_step(c, field, ps, pair, n, lane, st.ped_v_pref[pq], wp + 2 * head, dt, pp, pvel);
    } else if (c.ped_model == NAVSIM_PED_EXTERNAL && is_ped) {
        const double* cmd = st.ped_cmd + pq * 2;
        nv::set_vel(pp, cmd[0], cmd[1], dt, 0.0, pvel);     // env.py:662
    }
    bool due = false;
    if (is_ped) due = ped_finish(c, st, e, lane, pq, wp, head, nw, dt, genv, steps_now, pp, pvel);
    // who waits for navsim_replan (navsim_state.ped_due): one word per arena, pedestrian i = bit i (this wavefront's lanes)
    const unsigned long long m = __ballot(due);
    if (st.ped_due && lane == 0) st.ped_due[e] = m;
}

// The pedestrians of every arena ahead of the fused step (navsim_config.ped_split = 2; rounds 1-2 took this form for large
// batches because the fused phase then held three of a workgroup's four wavefronts at a barrier -- since it runs on
// wavefront 0 beside the scan, ped_stage_wave / ped_advance_wave, the fused form is faster at every batch size and this
// kernel is kept for callers that ask for it).  A workgroup takes a PACK of arenas: arena s of the pack owns threads
// [s N, (s + 1) N) for its pedestrians, and the pair terms of the whole pack form ONE list walked by all threads (2 x 210
// terms over 128 threads).  Same functions per pedestrian and per pair as the fused form: same results.
// (measured on c3, same box, threads x arenas per workgroup: 64 x 1 19.56 M env-steps/s, 64 x 3 17.8, 128 x 1 19.55,
// 128 x 2 19.94, 256 x 1 19.35; by counters 2 290 vector instructions per wavefront at 0.40 of the issue cycles,
// profiles/_diag/ped_update_pmc.sh)
#ifndef NAVSIM_PED_UPDATE_BLOCK
#define NAVSIM_PED_UPDATE_BLOCK 128
#endif
#ifndef NAVSIM_PED_PACK_MAX
#define NAVSIM_PED_PACK_MAX 2
#endif
constexpr int kPedUpdateBlock = NAVSIM_PED_UPDATE_BLOCK;
constexpr int kPedPack = 8;
__host__ __device__ inline size_t ped_slot_bytes(int N) { return ped_pair_bytes(N) + ((ped_lds_bytes(N) + 15) & ~(size_t)15); }
// arenas per workgroup of ped_update_kernel: as many as its threads cover, at most NAVSIM_PED_PACK_MAX, and no more than fit the
// 64 KB of dynamic LDS a kernel gets without asking (64 pedestrians: 42 KB per arena -> one; round-3 advisor: two did not
// fit and cfg.ped_split = 2 silently ran the fused form)
__host__ __device__ inline int ped_pack(int N) {
    int g = N > 0 ? kPedUpdateBlock / N : 1;
    g = g > NAVSIM_PED_PACK_MAX ? NAVSIM_PED_PACK_MAX : g;
    g = g < 1 ? 1 : (g > kPedPack ? kPedPack : g);
    while (g > 1 && (size_t)g * ped_slot_bytes(N) > 64 * 1024) --g;
    return g;
}
template <typename Field>
__global__ __launch_bounds__(kPedUpdateBlock) void ped_update_kernel(navsim_config c, navsim_state st) {
    extern __shared__ __attribute__((aligned(16))) char ped_dyn[];
    __shared__ int slot_n[kPedPack], slot_off[kPedPack + 1];
    const int lane = threadIdx.x;                                      // thread of the workgroup
    const int N = c.max_peds, G = ped_pack(N), P = c.max_waypoints;
    const int s = lane / N, i = lane - s * N;                          // arena of the pack, pedestrian of the arena
    const int e = (int)blockIdx.x * G + s;
    const bool slot_ok = s < G && e < c.n_envs;
    int n = slot_ok ? st.n_peds[e] : 0;
    n = n > N ? N : (n < 0 ? 0 : n);
    const bool is_ped = slot_ok && i < n;
    const size_t slot_bytes = ped_slot_bytes(N);
    char* my = ped_dyn + (size_t)(slot_ok ? s : 0) * slot_bytes;
    const PedShared ps = ped_lds_carve(my + ped_pair_bytes(N), N);
    if (lane < kPedPack) slot_n[lane] = 0;
    __syncthreads();
    if (slot_ok && i == 0) slot_n[s] = n;
    const size_t pq = (size_t)(slot_ok ? e : 0) * N + (is_ped ? i : 0);
    const Field field(st.field, st.field_overflow, map_slot_of(c, st, slot_ok ? e : 0), c.map_h, c.map_w);
    double pp[3] = {0.0, 0.0, 0.0}, pvel[2] = {0.0, 0.0};
    double* wp = st.ped_waypoints + (pq * P) * 2;
    int nw = 1, head = 0;
    if (is_ped) {
        pp[0] = st.ped_pose[pq * 3]; pp[1] = st.ped_pose[pq * 3 + 1]; pp[2] = st.ped_pose[pq * 3 + 2];
        pvel[0] = st.ped_vel[pq * 2]; pvel[1] = st.ped_vel[pq * 2 + 1];
        nw = st.ped_n_waypoints[pq];
        head = ped_pop_waypoints(wp, st.ped_wp_head[pq], nw, pp);
    }
    if (st.ped_due && slot_ok && i == 0) st.ped_due[e] = 0ull;          // (ordered before the atomics below by the barriers in between)
    const double dt = c.time_step;
    if (c.ped_model == NAVSIM_PED_SFM) {
        if (is_ped) { ps.ax[i] = pp[0]; ps.ay[i] = pp[1]; ps.avx[i] = pvel[0]; ps.avy[i] = pvel[1]; }
        if (slot_ok && i == 0 && n > 0) {                              // the robot at time t: entry n of the arena's arrays
            const double* rp = st.robot_pose + 3 * (size_t)e;
            const double prev_v = st.prev_action[2 * (size_t)e];
            double sn, cs;
            nv::sincos(rp[2], sn, cs);
            ps.ax[n] = rp[0]; ps.ay[n] = rp[1];
            ps.avx[n] = prev_v * cs; ps.avy[n] = prev_v * sn;
        }
        __syncthreads();
        if (lane == 0) {                                               // the pack's pair terms as one list
            int off = 0;
            for (int q = 0; q < G; ++q) { slot_off[q] = off; const int m = slot_n[q]; off += m * (m - 1) / 2 + m; }
            slot_off[G] = off;
        }
        __syncthreads();
        const int total = slot_off[G];
        for (int t = lane; t < total; t += kPedUpdateBlock) {
            int q = 0;
            while (t >= slot_off[q + 1]) ++q;
            char* base = ped_dyn + (size_t)q * slot_bytes;
            ped_pair_term(c, ped_lds_carve(base + ped_pair_bytes(N), N), (double2*)base, slot_n[q], t - slot_off[q]);
        }
        __syncthreads();
        if (is_ped) ped_sfm_step(c, field, ps, (const double2*)my, n, i, st.ped_v_pref[pq], wp + 2 * head, dt, pp, pvel);
    } else if (c.ped_model == NAVSIM_PED_EXTERNAL && is_ped) {
        const double* cmd = st.ped_cmd + pq * 2;
        nv::set_vel(pp, cmd[0], cmd[1], dt, 0.0, pvel);               // env.py:662
    }
    __syncthreads();                                                   // every arena's ped_due word is zero by now
    // the step increments steps[e] before anything else (env.py:592); it has not run yet
    if (is_ped) {
        const bool due = ped_finish(c, st, e, i, pq, wp, head, nw, dt, (uint64_t)(c.env_index_base + e), (uint64_t)st.steps[e] + 1, pp, pvel);
        if (due && st.ped_due) atomicOr(&st.ped_due[e], 1ull << i);
    }
}

// The fused step.  BLOCK threads = one arena; PEDS: the pedestrian variants (primitives + culled merge in LDS);
// RULE: the march step rule (NAVSIM_MARCH_*), a compile-time copy of cfg.march_rule so that the probe loop
// carries no select.  PINL (pedestrian variants): the pedestrian phase is compiled into the kernel (wavefront 0 runs it
// beside the scan of the others; 36-44 bytes of private scratch per lane under the 64-register cap, none in a hot loop);
// false = reset-only launches and launches behind ped_update_kernel: no pedestrian phase, Scratch_Size 0.
// The arena with a waiting pedestrian that launch slot `slot` of a NAVSIM_STEP_DUE launch steps: the slot-th arena, in index
// order, whose word of st.ped_due_prev is not zero; -1 when fewer arenas wait.  Every workgroup of that launch finds its own
// arena (a prefix count over E words from L2, ~2 us): the launch is as small as the work -- a few dozen workgroups instead of
// one per arena, nearly all of which would only claim their LDS to find out that they have nothing to do
// (profiles/r05_replan/timeline_v2.txt: 55-85 us for ~80 arenas).
template <int BLOCK>
__device__ __forceinline__ int due_arena_pick(const unsigned long long* __restrict__ due, int E, int slot,
                                              int* peds_before = nullptr, int* n_peds = nullptr) {
    // peds_before: waiting pedestrians of the arenas in front of the returned one; n_peds: waiting pedestrians in all
    __shared__ int wave_a[kMaxWaves], wave_p[kMaxWaves], e_s, before_s;
    const int tid = threadIdx.x, lane = tid & 63, nthr = block_threads<BLOCK>();
    const int per = (E + nthr - 1) / nthr;
    const int lo = tid * per, hi = (lo + per < E) ? lo + per : E;
    int n = 0, m = 0;
    for (int e = lo; e < hi; ++e) { const unsigned long long w = due[e]; n += w != 0ull; m += __popcll(w); }
    int incl = n, incl_p = m;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64), vp = __shfl_up(incl_p, off, 64);
        if (lane >= off) { incl += v; incl_p += vp; }
    }
    if (lane == 63 || tid == nthr - 1) { wave_a[tid >> 6] = incl; wave_p[tid >> 6] = incl_p; }
    if (tid == 0) { e_s = -1; before_s = 0; }
    __syncthreads();
    int pos = incl - n, pos_p = incl_p - m, all_p = 0;
    for (int w = 0; w < (nthr + 63) / 64; ++w) {
        if (w < (tid >> 6)) { pos += wave_a[w]; pos_p += wave_p[w]; }
        all_p += wave_p[w];
    }
    if (slot >= pos && slot < pos + n)
        for (int e = lo; e < hi; ++e) {
            const unsigned long long w = due[e];
            if (w == 0ull) continue;
            if (pos++ == slot) { e_s = e; before_s = pos_p; break; }
            pos_p += __popcll(w);
        }
    __syncthreads();
    const int e = e_s;
    if (peds_before) *peds_before = before_s;
    if (n_peds) *n_peds = all_p;
    __syncthreads();                                     // (e_s is rewritten by the next pick of this workgroup)
    return e;
}

// rank of arena e among the arenas with a waiting pedestrian (those in front of it), and the waiting pedestrians in front
template <int BLOCK>
__device__ __forceinline__ int due_arena_rank(const unsigned long long* __restrict__ due, int e, int* peds_before) {
    __shared__ int wave_a[kMaxWaves], wave_p[kMaxWaves];
    const int tid = threadIdx.x, lane = tid & 63, nthr = block_threads<BLOCK>();
    int n = 0, m = 0;
    for (int k = tid; k < e; k += nthr) { const unsigned long long w = due[k]; n += w != 0ull; m += __popcll(w); }
    for (int off = 32; off > 0; off >>= 1) { n += __shfl_down(n, off, 64); m += __shfl_down(m, off, 64); }
    if (lane == 0) { wave_a[tid >> 6] = n; wave_p[tid >> 6] = m; }
    __syncthreads();
    int rank = 0, before = 0;
    for (int w = 0; w < (nthr + 63) / 64; ++w) { rank += wave_a[w]; before += wave_p[w]; }
    *peds_before = before;
    __syncthreads();
    return rank;
}

// navsim_step_install: what the step needs to install a finished arena's STAGED world itself (navsim_regen_swap's work, done by
// the arena's own workgroup in place of the respawn's second scan).  big[]: field, overflow plane, rect records, costmap, rect
// index rows -- per-arena stride = bytes.
#ifndef NAVSIM_INSTALL_UNROLL
#define NAVSIM_INSTALL_UNROLL 8
#endif
struct StepInstallBig { char* dst; const char* src; size_t bytes; };
struct StepInstall {
    navsim_state stage;                  // the staged state (include/navsim.h navsim_regen_stage)
    const float* stage_obs;              // [E][D] first observations of the staged worlds
    uint8_t* mark;                       // [E rounded up to 4] "stage me again" flags (32-bit atomics)
    const long long* ready;              // [E] episode number the last finished staging pass generated for
    // NAVSIM_AUTORESET_NEXT_STEP only (navsim_step_install_next): lateness is decided when the episode ENDS -- late_next[e] = 1
    // for an arena that finishes in this call and finds no world staged for the episode it will start (0 for everyone else) --
    // and late_prev = the flags the previous call wrote: an arena flagged there is reset by the caller's navsim_regen, which runs
    // BESIDE this launch on another stream (this launch only zeroes the arena's outputs)
    uint8_t* late_next;
    const uint8_t* late_prev;
    int lone;                            // NEXT_STEP: an arena that is reset and finds no world staged regenerates its own
                                         // (regen_lone, kernels_regen_dev.hpp) -- no fallback launch at all; the host sets it for
                                         // worlds of outdoor maps without planning / costmap and >= 256 threads per arena
    uint8_t* late;                       // [E] or NULL: out, 1 = the arena finished, is due a new world and its staged one was not
                                         // ready -- the caller regenerates it now (navsim_regen with these flags as io->done)
    StepInstallBig big[5];
};
// "stage arena e again, for episode ep": the number first, then the flag (a staging pass that takes the flag -- possibly while
// this launch still runs -- must see the number: both go through the device-coherent path)
__device__ __forceinline__ void stage_request(int64_t* stage_episode, uint8_t* mark, int e, int64_t ep) {
    __hip_atomic_store(&stage_episode[e], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    atomicOr((unsigned*)(mark + (e & ~3)), 1u << (8 * (e & 3)));
}
// ready[e] as the last finished staging pass left it, with acquire semantics at device scope: the rows install_arena is about to
// read were written by that pass, possibly while this launch was already running, and a line of them may sit stale in this CU's
// vector cache from an earlier install of a neighbouring arena (round-5 advisor) -- the acquire invalidates it
__device__ __forceinline__ long long stage_ready(const long long* ready, int e) {
    return __hip_atomic_load(&ready[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
// behind install_arena, whole workgroup: "stage the world after this one".  Only now -- a staging pass that takes the flag writes
// into the slot install_arena has just handed over and over the staged rows it has just read (round-5 advisor: the request used
// to leave in the reward block, ahead of the copies, and held only because a pass needs longer to start than an install to end)
__device__ __forceinline__ void install_publish(const StepInstall& in, const navsim_state& live, int e) {
    __syncthreads();                                         // every thread's loads of the staged rows and stores of the live ones
    if (threadIdx.x == 0) {
        __threadfence();
        stage_request(in.stage.episode, in.mark, e, live.episode[e] + 1);
    }
}
// every array navsim_regen writes, arena e, staged -> live (the list of regen_swap_kernel); whole workgroup
template <int BLOCK>
__device__ __forceinline__ void install_arena(const navsim_config& c, const navsim_state& live, const navsim_step_io& io,
                                              const StepInstall& in, const int e, float* __restrict__ obs_row) {
    const int tid = threadIdx.x;
    // the map: with slot tables in both states (they then share the five per-map arrays) the two entries change places --
    // the staged map becomes the live one where it lies, the old live map is what the next staging pass overwrites;
    // otherwise the map is copied, by this one workgroup (c5, 500 x 500 cells: 600 KB, 12 us at the end of the launch)
    const bool by_slot = live.map_slot && in.stage.map_slot;
    if (by_slot && tid == 0) {
        const int32_t a = live.map_slot[e], b = in.stage.map_slot[e];
        live.map_slot[e] = b; in.stage.map_slot[e] = a;
    }
    for (int k = 0; k < 5 && !by_slot; ++k) {
        if (!in.big[k].dst) continue;
        const size_t bytes = in.big[k].bytes;
        const char* src = in.big[k].src + (size_t)e * bytes;
        char* dst = in.big[k].dst + (size_t)e * bytes;
        if ((((size_t)(uintptr_t)src | (size_t)(uintptr_t)dst | bytes) & 15) == 0) {
            const size_t n16 = bytes / 16;
            size_t i = tid;
#ifndef NAVSIM_DIAG_INSTALL_NO_BIG      // diagnostic build only (WRONG worlds): what do the map copies cost the launch?
            constexpr int U = NAVSIM_INSTALL_UNROLL;             // loads in flight per lane: ONE workgroup moves the map
            for (; i + (U - 1) * BLOCK < n16; i += U * BLOCK) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)src + i + u * BLOCK);   // read once
#pragma unroll
                for (int u = 0; u < U; ++u) ((u32x4*)dst)[i + u * BLOCK] = v[u];
            }
            for (; i < n16; i += BLOCK) ((uint4*)dst)[i] = ((const uint4*)src)[i];
#endif
        } else {
            for (size_t i = tid; i < bytes; i += BLOCK) dst[i] = src[i];
        }
    }
    const navsim_state& stage = in.stage;
    const int N = c.max_peds, K = c.n_spawn, P = c.max_waypoints, D = c.n_scan_stack * c.n_beams + NAVSIM_OBS_TAIL;
    const bool peds = c.ped_model != NAVSIM_PED_NONE;
    // The small arrays.  Every row is a load -> store round trip, and rows may alias as far as the compiler knows: walking through
    // the twenty of them one after the other is twenty memory latencies (~25 us of the arena's tail on c5).  So: every row's
    // first BLOCK elements are LOADED first, all rows, then stored; what a row has beyond BLOCK elements follows in a loop.
#define NAVSIM_ROW_LOAD(name, dstp, srcp, cnt) \
    const size_t n_##name = ((dstp) && (srcp)) ? (size_t)(cnt) : 0; \
    auto v_##name = (srcp)[tid < (int)n_##name ? (size_t)e * n_##name + tid : 0];
#define NAVSIM_ROW_STORE(name, dstp, srcp) \
    if (tid < (int)n_##name) (dstp)[(size_t)e * n_##name + tid] = v_##name; \
    for (size_t i = (size_t)tid + BLOCK; i < n_##name; i += BLOCK) (dstp)[(size_t)e * n_##name + i] = (srcp)[(size_t)e * n_##name + i];
    // (a row that is absent has n = 0: its load reads element 0 of a pointer that may be NULL -- guard the pointer itself)
#define NAVSIM_ROW_SRC(p) ((p) ? (p) : (decltype(p))live.steps)
    NAVSIM_ROW_LOAD(noise, live.scan_noise_std, NAVSIM_ROW_SRC(stage.scan_noise_std), 1)
    NAVSIM_ROW_LOAD(pose, live.robot_pose, NAVSIM_ROW_SRC(stage.robot_pose), 3)
    NAVSIM_ROW_LOAD(goal, live.robot_goal, NAVSIM_ROW_SRC(stage.robot_goal), 2)
    NAVSIM_ROW_LOAD(pact, live.prev_action, NAVSIM_ROW_SRC(stage.prev_action), 2)
    NAVSIM_ROW_LOAD(ppose, live.prev_pose, NAVSIM_ROW_SRC(stage.prev_pose), 3)
    NAVSIM_ROW_LOAD(nhist, live.n_hist, NAVSIM_ROW_SRC(stage.n_hist), 1)
    NAVSIM_ROW_LOAD(steps, live.steps, NAVSIM_ROW_SRC(stage.steps), 1)
    NAVSIM_ROW_LOAD(spose, live.spawn_pose, NAVSIM_ROW_SRC(stage.spawn_pose), (size_t)K * 3)
    NAVSIM_ROW_LOAD(sgoal, live.spawn_goal, NAVSIM_ROW_SRC(stage.spawn_goal), (size_t)K * 2)
    NAVSIM_ROW_LOAD(npeds, peds ? live.n_peds : nullptr, NAVSIM_ROW_SRC(stage.n_peds), 1)
    NAVSIM_ROW_LOAD(ppos, peds ? live.ped_pose : nullptr, NAVSIM_ROW_SRC(stage.ped_pose), (size_t)N * 3)
    NAVSIM_ROW_LOAD(pvel, peds ? live.ped_vel : nullptr, NAVSIM_ROW_SRC(stage.ped_vel), (size_t)N * 2)
    NAVSIM_ROW_LOAD(pyaw, peds ? live.ped_prev_yaw : nullptr, NAVSIM_ROW_SRC(stage.ped_prev_yaw), N)
    NAVSIM_ROW_LOAD(pdist, peds ? live.ped_dist : nullptr, NAVSIM_ROW_SRC(stage.ped_dist), (size_t)N * 3)
    NAVSIM_ROW_LOAD(pvpref, peds ? (double*)live.ped_v_pref : nullptr, NAVSIM_ROW_SRC(stage.ped_v_pref), N)
    NAVSIM_ROW_LOAD(plegs, peds ? (uint8_t*)live.ped_has_legs : nullptr, NAVSIM_ROW_SRC(stage.ped_has_legs), N)
    NAVSIM_ROW_LOAD(pwp, peds ? live.ped_waypoints : nullptr, NAVSIM_ROW_SRC(stage.ped_waypoints), (size_t)N * P * 2)
    NAVSIM_ROW_LOAD(pnwp, peds ? live.ped_n_waypoints : nullptr, NAVSIM_ROW_SRC(stage.ped_n_waypoints), N)
    NAVSIM_ROW_LOAD(phead, peds ? live.ped_wp_head : nullptr, NAVSIM_ROW_SRC(stage.ped_wp_head), N)
    NAVSIM_ROW_LOAD(pgoal, peds ? live.ped_goal : nullptr, NAVSIM_ROW_SRC(stage.ped_goal), (size_t)N * 2)
    NAVSIM_ROW_LOAD(obs, obs_row - (size_t)e * D, in.stage_obs, D)
    NAVSIM_ROW_STORE(noise, live.scan_noise_std, stage.scan_noise_std)
    NAVSIM_ROW_STORE(pose, live.robot_pose, stage.robot_pose)
    NAVSIM_ROW_STORE(goal, live.robot_goal, stage.robot_goal)
    NAVSIM_ROW_STORE(pact, live.prev_action, stage.prev_action)
    NAVSIM_ROW_STORE(ppose, live.prev_pose, stage.prev_pose)
    NAVSIM_ROW_STORE(nhist, live.n_hist, stage.n_hist)
    NAVSIM_ROW_STORE(steps, live.steps, stage.steps)
    NAVSIM_ROW_STORE(spose, (double*)live.spawn_pose, stage.spawn_pose)
    NAVSIM_ROW_STORE(sgoal, (double*)live.spawn_goal, stage.spawn_goal)
    NAVSIM_ROW_STORE(npeds, live.n_peds, stage.n_peds)
    NAVSIM_ROW_STORE(ppos, live.ped_pose, stage.ped_pose)
    NAVSIM_ROW_STORE(pvel, live.ped_vel, stage.ped_vel)
    NAVSIM_ROW_STORE(pyaw, live.ped_prev_yaw, stage.ped_prev_yaw)
    NAVSIM_ROW_STORE(pdist, live.ped_dist, stage.ped_dist)
    NAVSIM_ROW_STORE(pvpref, (double*)live.ped_v_pref, stage.ped_v_pref)
    NAVSIM_ROW_STORE(plegs, (uint8_t*)live.ped_has_legs, stage.ped_has_legs)
    NAVSIM_ROW_STORE(pwp, live.ped_waypoints, stage.ped_waypoints)
    NAVSIM_ROW_STORE(pnwp, live.ped_n_waypoints, stage.ped_n_waypoints)
    NAVSIM_ROW_STORE(phead, live.ped_wp_head, stage.ped_wp_head)
    NAVSIM_ROW_STORE(pgoal, live.ped_goal, stage.ped_goal)
    NAVSIM_ROW_STORE(obs, obs_row - (size_t)e * D, in.stage_obs)
#undef NAVSIM_ROW_LOAD
#undef NAVSIM_ROW_STORE
#undef NAVSIM_ROW_SRC
    if (tid == 0) {
        if (io.achieved_goal) { io.achieved_goal[2 * e] = (float)stage.robot_pose[3 * (size_t)e]; io.achieved_goal[2 * e + 1] = (float)stage.robot_pose[3 * (size_t)e + 1]; }
        if (io.desired_goal) { io.desired_goal[2 * e] = (float)stage.robot_goal[2 * (size_t)e]; io.desired_goal[2 * e + 1] = (float)stage.robot_goal[2 * (size_t)e + 1]; }
        if (live.ped_due) live.ped_due[e] = 0ull;           // new pedestrians: nobody waits for navsim_replan
    }
}

// The fused step of ONE arena by the calling workgroup.  BLOCK threads; PEDS: the pedestrian variants (primitives + culled merge
// in LDS); RULE: the march step rule (NAVSIM_MARCH_*), a compile-time copy of cfg.march_rule so that the probe loop carries no
// select.  PINL (pedestrian variants): the pedestrian phase is compiled in (wavefront 0 runs it beside the scan of the others;
// 36-44 bytes of private scratch per lane under the 64-register cap, none in a hot loop); false = reset-only launches and
// launches behind ped_update_kernel: no pedestrian phase, Scratch_Size 0.
// INSTALL (navsim_step_install): an arena that finishes takes its staged world instead of restarting in place, when its episode
// was long enough (cfg.regen_min_steps) and the staged world is the one for the episode that starts (in->ready).
// FEAT (ABI 6): the terminal observation of an arena that restarts in the step that ends its episode (io.final_obs) and
// NAVSIM_AUTORESET_NEXT_STEP (io.reset_mask) are compiled in.  The pedestrian variants always carry them; the variants without
// pedestrians exist in both forms and the host takes the plain one for calls that use neither -- the code of round 5, whose
// register allocation in the probe loop the few extra live values of these paths disturb: measured on c2, same box, the one
// kernel with everything in it 42.3 M env-steps/s against 43.0 M, and 41.1-41.5 M for two semantically equal formulations of the
// new paths (profiles/r06_abi6/: the probe loop has the same 60 instructions in all of them and other registers).
template <int BLOCK, bool PEDS, typename Field, int RULE, int RECT, bool PINL, bool INSTALL = false, bool FEAT = true>
__device__ __forceinline__ void step_arena(const navsim_config& c, const navsim_state& st, const navsim_step_io& io, const int e,
                                           int reset_only, const int peds_done, const uint8_t* __restrict__ reset_mask,
                                           unsigned dyn_lds_bytes, int park_lanes, unsigned rect_lds_offset,
                                           const StepInstall* in = nullptr, const int base_prio = 0) {
    __shared__ StepShared sh;
    // dynamic LDS: [the arena's index row (RECT = 2)][parked rays][pedestrian variants: float2 dir[B], float rng[B]][PedShared][pair table]
    extern __shared__ __attribute__((aligned(16))) char dyn_lds_all[];
    char* dyn_lds = dyn_lds_all + (RECT == 2 ? rect_lds_offset : 0u);      // (rect_lds_offset = the row's size: the rest sits behind it)
    PedShared ps = {};
    if constexpr (PEDS) ps = ped_lds_carve(dyn_lds + ((dyn_lds_bytes + 15u) & ~15u), c.max_peds);
    const Prims prims = {ps.seg, ps.disc, ps.info};
    const int tid = threadIdx.x;
    // Wave priorities (round 6; s_setprio: the SIMD's arbiter issues from the highest priority first, then the oldest wavefront).
    // A workgroup's START -- the copy of its index row, phase 0's scalars on one lane, the pedestrian phase of wavefront 0 -- is
    // a chain of memory latencies with next to no vector work: at priority 3 it never waits behind the marching wavefronts of
    // the other arenas of the CU, and the workgroup reaches its own march sooner (c2 +1.9 %, c3 +1.9 %, c4 +1.5 %, c5 +2.1 %).
    // The 256-thread variants without pedestrians (several generations of workgroups per launch, parked rays) also march ABOVE
    // the packing phase of the workgroups that are done (2 against 0): c2 +5.9 % in all, 46.4 -> 49.1 M env-steps/s; the other
    // variants lose with that (c4 -1.4 %, c3 +-0) and keep their march at the caller's priority.  Measured forms that lost:
    // the march alone raised (c3 -9 %), everything behind scan A lowered (c4 -6 %), the tail raised (+-0): profiles/r06_c2/.
    constexpr int kMarchPrio = (BLOCK == 256 && !PEDS) ? 2 : 0;
    __builtin_amdgcn_s_setprio(3);
    unsigned long long t_begin = 0;
    if (st.arena_cost && tid == 0) t_begin = __builtin_amdgcn_s_memrealtime();
    const int B = c.n_beams, S = c.n_scan_stack, N = c.max_peds, D = S * B + 7;
    const double dt = c.time_step;
    const uint64_t genv = (uint64_t)(c.env_index_base + e);
    // where the arena's map lives: navsim_state.map_slot is honoured by the INSTALL instantiations only -- every launch on a
    // state with a slot table is routed to them (dispatch_step); the lookup in front of the first field access cost the plain
    // step 2.8 % on c2 (a scalar load per workgroup ahead of phase 0: 43.4 -> 42.2 M env-steps/s, same box)
    const int ms = INSTALL ? map_slot_of(c, st, e) : (c.shared_field ? 0 : e);
    const Field field(st.field, st.field_overflow, ms, c.map_h, c.map_w);
    const uint4* rects = RECT ? (const uint4*)st.rect_table + (size_t)ms * rect_tiles_per_map(c.map_h, c.map_w)
                              : nullptr;
    if constexpr (RECT == 2) {
        // "Map tiles staged through LDS": the INDEX form of the arena's record table (kernels_rect.hpp: 2 KB of distinct
        // rectangles + 2 bytes per 8x8 tile = 10 KB for 500 x 500 cells) is copied into LDS once, by all threads, beside
        // phase 0; every probe of the scans then reads LDS (~0.1 us) instead of global memory (0.5-2 us from L2 / HBM).
        const size_t row_bytes = rect_index_row_bytes(c.map_h, c.map_w);
        const uint4* src = (const uint4*)((const char*)st.rect_index + (size_t)ms * row_bytes);
        uint4* tab_lds = (uint4*)dyn_lds_all;
        for (int i = threadIdx.x; i < (int)(row_bytes / 16); i += BLOCK) tab_lds[i] = src[i];
        rects = tab_lds;                                        // made visible by the barrier that ends phase 0
    }
    float* obs_row = io.obs + (size_t)e * D;
    const float* obs_prev = io.obs_prev ? io.obs_prev + (size_t)e * D : nullptr;
    double* rp_g = st.robot_pose + 3 * (size_t)e;
    double* goal_g = st.robot_goal + 2 * (size_t)e;
    double* pa_g = st.prev_action + 2 * (size_t)e;
    double* pv_g = st.prev_pose + 3 * (size_t)e;

    // NAVSIM_AUTORESET_NEXT_STEP: an arena of io.reset_mask finished in the previous call and restarted in the STATE there; this
    // call RESETS it instead of stepping it -- the reset-only path for this one workgroup (first observation from the reset key,
    // pedestrians not advanced, action ignored), reward / done / info zero
    constexpr bool kFeat = PEDS || FEAT;
    const bool pending = kFeat && !reset_only && io.reset_mask && io.reset_mask[e] != 0;
    if (pending) {
        reset_only = 1;
        if (tid == 0) {
            io.reward[e] = 0.0; io.done[e] = 0; io.is_success[e] = 0.0f; io.is_crash[e] = 0.0f; io.distance[e] = 0.0;
            if (st.ped_due) st.ped_due[e] = 0ull;
        }
        // cfg.defer_reset_scan: the navsim_regen that follows (io->done = the same mask) writes the row
        if (c.defer_reset_scan) return;
        // (compiled into the default march rule's instantiations only: the other rules' kernels stay as small as they were, their
        //  callers get NAVSIM_E_UNSUPPORTED from the host and take the flags + navsim_regen form)
        if constexpr (INSTALL && BLOCK >= 256 && RULE == NAVSIM_MARCH_F32) {
            if (in->lone && in->ready) {
                // staged worlds, no rule, no fallback launch (round 6): this arena restarted -- in the state -- when it finished;
                // now its workgroup takes the staged world, or, should none be staged for the episode that starts, generates
                // that world ITSELF, in place of the step it does not take (regen_lone).  Decided by one thread, through LDS.
                if (tid == 0) {
                    const bool lng = c.regen_min_steps <= 0 || st.done_steps[e] >= c.regen_min_steps;
                    const bool rdy = stage_ready(in->ready, e) == (long long)st.episode[e];
                    sh.respawn = lng ? (rdy ? 3 : 5) : 0;
                    if (!(lng && rdy)) stage_request(in->stage.episode, in->mark, e, st.episode[e] + 1);   // what is staged carries a stale number
                    if (st.counters) {
                        atomicAdd(&st.counters[lng ? (rdy ? NAVSIM_COUNTER_REGEN_SERVED : NAVSIM_COUNTER_REGEN_LATE) : NAVSIM_COUNTER_REGEN_SHORT], 1ull);
                        if (lng && !rdy) atomicAdd(&st.counters[NAVSIM_COUNTER_REGEN_SERVED], 1ull);       // (as navsim_regen counts the arena it serves)
                    }
                }
                __syncthreads();
                const int verdict = sh.respawn;
                __syncthreads();                                // (phase 0 rewrites the field)
                if (verdict == 3) {
                    install_arena<BLOCK>(c, st, io, *in, e, obs_row);
                    install_publish(*in, st, e);
                    return;
                }
                if (verdict == 5) {
                    regen_lone<Field, BLOCK>(c, st, e);
                    if constexpr (RECT == 2) {                  // the new map's index row replaces the old one in LDS
                        const size_t row_bytes = rect_index_row_bytes(c.map_h, c.map_w);
                        const uint4* src = (const uint4*)((const char*)st.rect_index + (size_t)ms * row_bytes);
                        uint4* tab_lds = (uint4*)dyn_lds_all;
                        for (int i = threadIdx.x; i < (int)(row_bytes / 16); i += BLOCK) tab_lds[i] = src[i];
                    }
                }
                // (verdict 0, 5: the ordinary reset path below -- first observation from the state as it stands now)
            }
        }
    }
    if (reset_only && reset_mask && !reset_mask[e]) {          // untouched env: carry the row over
        if (obs_prev && obs_prev != obs_row)                    // (navsim_regen hands the step's own rows in: nothing to move)
            for (int k = tid; k < D; k += BLOCK) obs_row[k] = obs_prev[k];
        return;
    }

    int n = (!PEDS || c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    const float noise_std = (c.add_scan_noise && st.scan_noise_std) ? st.scan_noise_std[e] : 0.0f;

    NAVSIM_STAMP(0);
    // wavefront 0 of the pedestrian variants: pedestrian i on lane i.  Its loads, the waypoint pop and the staging of the
    // agents run beside phase 0, which a lane of wavefront 1 computes.
    double pp[3] = {0.0, 0.0, 0.0};
    double pvel[2] = {0.0, 0.0};
    const bool is_ped = PEDS && tid < n;                        // n <= 64
    const size_t pq = (size_t)e * N + (is_ped ? tid : 0);
    const bool ped_advance = PEDS && PINL && !reset_only && !peds_done;
    int ped_head = 0, ped_nw = 1;
    if constexpr (PEDS) {
        if (tid < 64) {
            if (is_ped) {
                pp[0] = st.ped_pose[pq * 3]; pp[1] = st.ped_pose[pq * 3 + 1]; pp[2] = st.ped_pose[pq * 3 + 2];
                pvel[0] = st.ped_vel[pq * 2]; pvel[1] = st.ped_vel[pq * 2 + 1];
            }
            if constexpr (PINL) {
                if (ped_advance) ped_head = ped_stage_wave(c, st, n, tid, is_ped, pq, rp_g, pa_g[0], ps, pp, pvel, ped_nw);
            }
        }
    }
    constexpr int kPhase0Thread = (PEDS && BLOCK > 64) ? 64 : 0;
    // ---------------------------------------------------------------- phase 0: scalars, robot, lidar pose
    // The robot moves here, ahead of the pedestrians (env.py:664 comes after their commands, but nothing the pedestrian
    // phase computes reads the robot's new pose -- the social force sees old_rp -- and nothing here reads a pedestrian):
    // after ONE barrier the scan can start while wavefront 0 is still with the pedestrians.
    // Its three sincos -- the robot's old heading, its new heading, the float32 lidar heading -- depend on the old heading and
    // the action only: three lanes evaluate one each, at once (a lane's work costs the wavefront's issue slots whether one
    // lane or three are live; two sincos fewer are 180 of an arena's 11 500 vector instructions on c2).
    const int l0 = tid - kPhase0Thread;
    if (l0 >= 0 && l0 < 3) {
        const double th_old = rp_g[2];
        double a0 = 0.0, a1 = 0.0;
        if (!reset_only) {
            a0 = io.action[2 * e]; a1 = io.action[2 * e + 1];
            if (c.action_kind == NAVSIM_ACTION_WHEELS) {       // skid-steer wheel speeds (left, right) -> twist (include/navsim.h)
                const double wl = a0, wr = a1;
                a0 = c.wheel_radius * (wl + wr) * 0.5;
                a1 = c.wheel_radius * (wr - wl) / c.wheel_track;
            }
            if (c.clamp_action) {                              // build option; the reference never clips (env.py:611-613)
                a0 = a0 < c.linvel_lo ? c.linvel_lo : (a0 > c.linvel_hi ? c.linvel_hi : a0);
                a1 = a1 < c.rotvel_lo ? c.rotvel_lo : (a1 > c.rotvel_hi ? c.rotvel_hi : a1);
            }
            if (c.min_turning_radius > 0.0) {                  // env.py:595-600
                double lim = fabs(a1) * c.min_turning_radius;
                if (a0 >= 0.0) a0 = (a0 > lim) ? a0 : lim;
                else           a0 = (a0 < -lim) ? a0 : -lim;
            }
        }
        const double th = th_old + a1 * dt;                     // set_vel's intermediate heading
        const double th_new = reset_only ? th_old : nv::mod_2pi(th);
        const float lth = (float)th_new;                        // env.py:386
        const double ang = (l0 == 0) ? th_old : ((l0 == 1) ? th : (double)lth);
        double sn, cs;
        nv::sincos(ang, sn, cs);
        const int w0 = kPhase0Thread & 63;                      // lane of l0 = 0 inside its wavefront (0)
        const double s0 = __shfl(sn, w0), c0 = __shfl(cs, w0), s1 = __shfl(sn, w0 + 1), c1 = __shfl(cs, w0 + 1);
        const double sT = __shfl(sn, w0 + 2), cT = __shfl(cs, w0 + 2);
        if (l0 == 0) {
            sh.old_rp[0] = rp_g[0]; sh.old_rp[1] = rp_g[1]; sh.old_rp[2] = th_old;
            sh.nseg = 0; sh.ndisc = 0; sh.rescan = 0; sh.respawn = 0; sh.term = 0;
            if constexpr (INSTALL) {
                if (pending && !in->lone) {
                    // NEXT_STEP with staged worlds: the arena restarted (state: episode number, done_steps) when it finished; now
                    // it takes its staged world if that is the one for the episode that starts (and the ended episode was long
                    // enough, cfg.regen_min_steps), else it starts in place and -- no rule -- asks the caller's navsim_regen
                    if (in->late) in->late[e] = 0;
                    if (in->late_next) in->late_next[e] = 0;
                    if (in->late_prev && in->late_prev[e]) {
                        // flagged when its episode ended (below, `restart_next`): its new world is being generated right now, beside
                        // this launch, by the caller's navsim_regen -- state, map and first observation are that call's
                        sh.respawn = 4;
                        stage_request(in->stage.episode, in->mark, e, st.episode[e] + 1);       // what is staged carries a stale number
                        if (st.counters) atomicAdd(&st.counters[NAVSIM_COUNTER_REGEN_LATE], 1ull);
                    } else if (in->ready) {
                        const bool lng = c.regen_min_steps <= 0 || st.done_steps[e] >= c.regen_min_steps;
                        const bool rdy = stage_ready(in->ready, e) == (long long)st.episode[e];
                        if (lng && rdy) sh.respawn = 3;                                // installed behind the barrier below
                        else {
                            if (lng && in->late) in->late[e] = 1;
                            stage_request(in->stage.episode, in->mark, e, st.episode[e] + 1);   // what is staged carries a stale number
                        }
                        if (st.counters)
                            atomicAdd(&st.counters[lng ? (rdy ? NAVSIM_COUNTER_REGEN_SERVED : NAVSIM_COUNTER_REGEN_LATE) : NAVSIM_COUNTER_REGEN_SHORT], 1ull);
                    }
                }
            }
            if (!reset_only) {
                const int64_t steps_now = st.steps[e] + 1;     // env.py:592
                st.steps[e] = steps_now;
                sh.steps_now = (int)steps_now;
                sh.act[0] = a0; sh.act[1] = a1;
                double p[3] = {sh.old_rp[0], sh.old_rp[1], sh.old_rp[2]};        // env.py:664
                nv::set_vel_with(p, a0, a1, dt, c.axle_offset, s0, c0, s1, c1, nullptr);
                sh.rp[0] = p[0]; sh.rp[1] = p[1]; sh.rp[2] = p[2];
            } else {
                sh.rp[0] = sh.old_rp[0]; sh.rp[1] = sh.old_rp[1]; sh.rp[2] = sh.old_rp[2];
            }
            sh.lx = (float)sh.rp[0]; sh.ly = (float)sh.rp[1]; sh.lth = (float)sh.rp[2];   // env.py:386
            nv::xy_to_ij_f32(sh.lx, sh.ly, c, sh.i0, sh.j0);                              // env.py:419
            sh.sT = sT; sh.cT = cT;                            // sincos((double)sh.lth): sh.rp[2] is th_new
            first_probe<RULE>(field, sh.i0, sh.j0, (float)((long long)c.map_h * c.map_w), sh.t1, sh.r_all);
            sh.step_key = (unsigned long long)st.episode[e] * 0x100000000ULL +
                          (unsigned long long)(reset_only ? 0 : st.steps[e]) * 2ULL;
            sh.rescan_key = sh.step_key + 1;
            sh.next_chunk = 0; sh.park_count = 0; sh.park_next = 0; sh.pair_done = 0;
        }
    }
    __syncthreads();
    if constexpr (INSTALL) {
        if (pending && sh.respawn == 3) {                       // NEXT_STEP: the reset of this arena IS the install of its staged world
            install_arena<BLOCK>(c, st, io, *in, e, obs_row);
            install_publish(*in, st, e);
            return;
        }
        if (pending && sh.respawn == 4) return;                 // ... or the caller's navsim_regen beside this launch
    }

    NAVSIM_STAMP(1);
    // ---------------------------------------------------------------- phases 1 + 2: pedestrians, what the lidar sees of them
    // wavefront 0 alone (pedestrian i on lane i); the others go straight to the scan, whose barrier before the merge
    // publishes the primitives
    if constexpr (PEDS) {
        // the pair table: dynamic LDS behind PedShared (the host allocates it for the launches that carry the phase)
        double2* pair = (double2*)(dyn_lds + ((dyn_lds_bytes + 15u) & ~15u) + ((ped_lds_bytes(N) + 15) & ~(size_t)15));
#ifdef NAVSIM_PAIR_SHARE_ALL
        constexpr int kPairWaves = (BLOCK >= NAVSIM_PAIR_SHARE_MIN_BLOCK) ? 4 : 1;       // wavefronts that share the pair terms
#else
        constexpr int kPairWaves = (INSTALL && BLOCK >= NAVSIM_PAIR_SHARE_MIN_BLOCK) ? 4 : 1;       // wavefronts that share the pair terms
#endif
        const bool pair_shared = PINL && kPairWaves > 1 && ped_advance && c.ped_model == NAVSIM_PED_SFM;
        if constexpr (PINL && kPairWaves > 1) {
            if (pair_shared && tid < kPairWaves * 64) ped_pair_share(c, ps, pair, n, tid, kPairWaves, &sh.pair_done);
        }
        if (tid < 64) {
            const int lane = tid;
            if constexpr (PINL) {
                if (ped_advance)
                    ped_advance_wave<Field>(c, st, field, e, n, lane, is_ped, pq, dt, genv, (uint64_t)st.steps[e], ps, pair,
                                            ped_head, ped_nw, pp, pvel, &sh.pair_done, pair_shared ? kPairWaves : 1);
            }
            if (reset_only && is_ped) {                             // env.py:809, 812-820
                st.ped_dist[pq * 3] = 0.0; st.ped_dist[pq * 3 + 1] = 0.0; st.ped_dist[pq * 3 + 2] = 0.0;
                st.ped_prev_yaw[pq] = nv::wrap_pi(pp[2]);
            }
            if (is_ped) {                                           // env.py:392-414
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
            wave_lds_sync();
            prims_prepare<64>(c, sh, prims);                        // in-range flags of the culled merge (this wavefront's lanes)
        }
    }

    NAVSIM_STAMP(2);
    if (kMarchPrio == 2 || base_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (base_prio == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    // ---------------------------------------------------------------- phase 3: scan A
    int n_hist = reset_only ? 0 : st.n_hist[e];
    int crash = 0, discomfort = 0;
    const uint64_t step_key = sh.step_key;
    // dynamic LDS: [parked rays][pedestrian variants: float2 dir[B], float rng[B]][PedShared][pair table (fused pedestrian phase)]
    float4* park = (float4*)dyn_lds;
    char* scan_lds = dyn_lds + park_lds_bytes(B, park_lanes, PEDS);
    float2* dir_lds = (float2*)scan_lds;
    float* rng_lds = (float*)(scan_lds + sizeof(float2) * (size_t)B);
    scan_beams_pred<BLOCK, Field, PEDS, RULE, RECT>(c, sh, field, rects, st.beam_table, prims, dir_lds, rng_lds, park, park_lanes, st.scan_threshold,
                                                         st.scan_discomfort, obs_row, n_hist, noise_std, step_key, genv, crash, discomfort, false);

    NAVSIM_STAMP(3);
    bool restart_next = false;                                  // (thread 0) NEXT_STEP: the state restarts behind phase 6
    if (!reset_only) {
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
            if constexpr (INSTALL) { if (in->late) in->late[e] = 0; if (in->late_next) in->late_next[e] = 0; }
            io.is_success[e] = o.success;
            io.is_crash[e] = o.crash;
            io.distance[e] = o.distance;
            const bool restart = o.done && c.auto_reset != NAVSIM_AUTORESET_NONE && c.n_spawn > 0;
            if (restart && c.auto_reset == NAVSIM_AUTORESET_SAME_STEP) {      // build-defined respawn
                if (kFeat && io.final_obs) {
                    // the observation the reference's step() returns with done = True (env.py:700-728), before the restart takes
                    // the row: tail and goals here; the scan rows from scan A (term 1) or, after a crash, from the re-scan at
                    // the reverted pose (term 2, env.py:707-723) below
                    const bool crashed = o.crash != 0.0f;
                    const double tx = crashed ? pv_g[0] : sh.rp[0], ty = crashed ? pv_g[1] : sh.rp[1];
                    const double tyaw = nv::wrap_pi(crashed ? pv_g[2] : sh.rp[2]);
                    float* ft = io.final_obs + (size_t)e * D + (size_t)S * B;
                    ft[0] = (float)pv_g[0]; ft[1] = (float)pv_g[1];
                    ft[2] = (float)tx; ft[3] = (float)ty;
                    ft[4] = (float)pa_g[0]; ft[5] = (float)pa_g[1];
                    ft[6] = (float)tyaw;
                    if (io.final_goals) {
                        float* fg = io.final_goals + 4 * (size_t)e;
                        fg[0] = (float)tx; fg[1] = (float)ty; fg[2] = (float)goal_g[0]; fg[3] = (float)goal_g[1];
                    }
                    sh.term = crashed ? 2 : 1;
                }
                uint64_t h = nv::hash4(c.seed, genv, (uint64_t)st.episode[e], 0x5eedULL);
                int idx = (int)(h % (uint64_t)c.n_spawn);
                const double* sp = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
                const double* sg = st.spawn_goal + ((size_t)e * c.n_spawn + idx) * 2;
                sh.rp[0] = sp[0]; sh.rp[1] = sp[1]; sh.rp[2] = sp[2];
                goal_g[0] = sg[0]; goal_g[1] = sg[1];
#ifndef NAVSIM_DIAG_NO_DONE_STEPS    // diagnostic build only: what does the store cost the plain step's code?
                if (st.done_steps) st.done_steps[e] = sh.steps_now;               // how long the episode lasted (cfg.regen_min_steps)
#endif
                st.episode[e] += 1;
                st.steps[e] = 0;
                // the first observation of the new episode draws its noise from the key a reset-only launch would use -- the
                // same whether this launch scans it or navsim_regen does (cfg.defer_reset_scan), i.e. whatever the shard size
                // (round-4 advisor: the two paths used different keys)
                sh.rescan_key = (unsigned long long)st.episode[e] * 0x100000000ULL;
                // cfg.defer_reset_scan: the first observation of the new episode comes from the navsim_regen call that follows
                // (one masked launch for every finished arena): no second scan here, the rows stay as scan A left them
                sh.respawn = 1; sh.rescan = c.defer_reset_scan ? 0 : 1;
                if constexpr (INSTALL) if (in->ready) {                 // (ready == NULL: a plain step on a state with slot tables)
                    // (done_steps[e] and episode[e] as this block just left them: the ended episode's length, the new number)
                    const bool lng = c.regen_min_steps <= 0 || st.done_steps[e] >= c.regen_min_steps;
                    const bool rdy = stage_ready(in->ready, e) == (long long)st.episode[e];
                    if (lng && rdy) { sh.respawn = 3; sh.rescan = 0; }          // 3: the staged world is installed below
                    else if (lng && in->late) in->late[e] = 1;                  // ... or generated now, by the caller's navsim_regen
                    if (st.counters)
                        atomicAdd(&st.counters[lng ? (rdy ? NAVSIM_COUNTER_REGEN_SERVED : NAVSIM_COUNTER_REGEN_LATE) : NAVSIM_COUNTER_REGEN_SHORT], 1ull);
                    // not installed: the arena plays its next episode in place, what is staged for it carries a stale number --
                    // stage it again.  Installed: the request for the world after this one leaves BEHIND the install
                    // (install_publish), when nothing of the staged world is read any more
                    if (!(lng && rdy)) stage_request(in->stage.episode, in->mark, e, st.episode[e] + 1);
                }
            } else {
                // NONE / NEXT_STEP: the outputs are the reference's (env.py:700-728); NEXT_STEP restarts the STATE behind phase 6
                restart_next = kFeat && restart;
                if (o.crash != 0.0f) {                          // env.py:707-717
                    sh.rp[0] = pv_g[0]; sh.rp[1] = pv_g[1]; sh.rp[2] = pv_g[2];
                    sh.rescan = 1;
                }
            }
#ifdef NAVSIM_DIAG_NO_RESCAN      // diagnostic build only (WRONG observations after a crash): what do the second scans cost the launch?
            sh.rescan = 0;               // (round 4, profiles/r04_jobs/ab_norescan.txt: c2 113.8 -> 105.3 us, c2 at 512 arenas 34.0 -> 31.1, c5 58 -> 47)
#endif
            if constexpr (!kFeat) {
                if (sh.rescan) {                                // the plain form: the second scan's set-up here, one barrier
                    sh.next_chunk = 0; sh.park_count = 0; sh.park_next = 0;
                    sh.lx = (float)sh.rp[0]; sh.ly = (float)sh.rp[1]; sh.lth = (float)sh.rp[2];
                    nv::xy_to_ij_f32(sh.lx, sh.ly, c, sh.i0, sh.j0);
                    nv::sincos((double)sh.lth, sh.sT, sh.cT);
                    first_probe<RULE>(field, sh.i0, sh.j0, (float)((long long)c.map_h * c.map_w), sh.t1, sh.r_all);
                }
            }
        }
        __syncthreads();
        if constexpr (!kFeat) {
            // ------------------------------------------------------------ phase 5: scan B (env.py:718-723) -- the plain form
            if (sh.rescan) {
                if (sh.respawn) n_hist = 0;
                int c2, d2;
                scan_beams_pred<BLOCK, Field, PEDS, RULE, RECT>(c, sh, field, rects, st.beam_table, prims, dir_lds, rng_lds, park, park_lanes, st.scan_threshold,
                                                                     st.scan_discomfort, obs_row, n_hist, noise_std, sh.rescan_key, genv, c2, d2);
            }
        } else {
            // ------------------------------------------------------------ the terminal row of an arena that restarts in this step
            float* final_row = io.final_obs ? io.final_obs + (size_t)e * D : nullptr;
            const int term = sh.term;
            if (term == 1) {
                // no crash: the reference returns scan A (this launch wrote it to the row's last slot; visible behind the barrier),
                // stacked on the previous rows (env.py:257-279)
                for (int j = 0; j < S; ++j) {
                    const bool old = j < S - 1 && S - 1 - j <= n_hist;
                    for (int k = tid; k < B; k += BLOCK)
                        final_row[(size_t)j * B + k] = old ? obs_prev[(size_t)(j + 1) * B + k] : obs_row[(size_t)(S - 1) * B + k];
                }
                __syncthreads();                                    // ... before scan B overwrites that slot
            }
            // ------------------------------------------------------------ phase 5: scan B (env.py:718-723)
            // pass 0 (only sh.term == 2: a crash ends the episode of an arena that restarts here): the reference's re-scan at the
            // reverted pose, into the terminal row; pass 1: the scan of the row itself -- crash revert, or the restart's first
            // observation.  ONE call site of the scan for both.
            for (int pass = (term == 2) ? 0 : 1; pass < 2; ++pass) {
                if (pass == 1 && !sh.rescan) break;
                if (tid == 0) {
                    const double* at = (pass == 0) ? pv_g : sh.rp;  // (pv_g: prev_pose, untouched until phase 6)
                    sh.next_chunk = 0; sh.park_count = 0; sh.park_next = 0;
                    sh.lx = (float)at[0]; sh.ly = (float)at[1]; sh.lth = (float)at[2];
                    nv::xy_to_ij_f32(sh.lx, sh.ly, c, sh.i0, sh.j0);
                    nv::sincos((double)sh.lth, sh.sT, sh.cT);
                    first_probe<RULE>(field, sh.i0, sh.j0, (float)((long long)c.map_h * c.map_w), sh.t1, sh.r_all);
                }
                __syncthreads();
                if (pass == 1 && sh.respawn) n_hist = 0;
                int c2, d2;
                scan_beams_pred<BLOCK, Field, PEDS, RULE, RECT>(c, sh, field, rects, st.beam_table, prims, dir_lds, rng_lds, park, park_lanes, st.scan_threshold,
                                                                     st.scan_discomfort, pass == 0 ? final_row : obs_row, n_hist, noise_std,
                                                                     pass == 0 ? step_key + 1 : sh.rescan_key, genv, c2, d2);
                if (pass == 0) {
                    for (int j = 0; j < S - 1; ++j)
                        if (S - 1 - j <= n_hist)
                            for (int k = tid; k < B; k += BLOCK) final_row[(size_t)j * B + k] = obs_prev[(size_t)(j + 1) * B + k];
                    __syncthreads();                                // every wavefront is out of the scan before its set-up is rewritten
                }
            }
        }
    }

    NAVSIM_STAMP(5);
    if (kMarchPrio != 0 && base_prio < kMarchPrio) {
        if (base_prio == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
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
        if (restart_next) {
            // NAVSIM_AUTORESET_NEXT_STEP: the row, reward, done and info above are the ended episode's; the STATE already belongs
            // to the next one -- the start / goal pair the same-step restart would take, the next episode number -- so that the
            // next call's reset of this arena (io.reset_mask; or the navsim_regen keyed on it) depends on nothing else
            uint64_t h = nv::hash4(c.seed, genv, (uint64_t)st.episode[e], 0x5eedULL);
            int idx = (int)(h % (uint64_t)c.n_spawn);
            const double* sp = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
            const double* sg = st.spawn_goal + ((size_t)e * c.n_spawn + idx) * 2;
            rp_g[0] = sp[0]; rp_g[1] = sp[1]; rp_g[2] = sp[2];
            goal_g[0] = sg[0]; goal_g[1] = sg[1];
            if (st.done_steps) st.done_steps[e] = sh.steps_now;
            st.episode[e] += 1;
            st.steps[e] = 0;
            if constexpr (INSTALL) {
                if (in->late_next && in->ready) {
                    // is the world for the episode that will start staged?  Decided NOW, a call ahead of the reset: an arena that
                    // finds nothing is regenerated by the caller BESIDE the next launch instead of behind it (and what this
                    // decides stands whatever a staging pass finishes in between: the regenerated world is the same world)
                    const bool lng = c.regen_min_steps <= 0 || sh.steps_now >= c.regen_min_steps;
                    const bool rdy = stage_ready(in->ready, e) == (long long)st.episode[e];
                    in->late_next[e] = (lng && !rdy) ? 1 : 0;
                }
            }
        }
        if (st.arena_cost && !reset_only)
            st.arena_cost[e] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_begin);
    }
    NAVSIM_STAMP(6);
    if constexpr (INSTALL) {
        __syncthreads();                                       // what phase 6 wrote for the restart in place is overwritten
        if (sh.respawn == 3) {
            install_arena<BLOCK>(c, st, io, *in, e, obs_row);
            install_publish(*in, st, e);
        }
        NAVSIM_STAMP(7);
    }
}

// Kernel arguments, read WHERE THEY ARE USED (round 6).  A struct passed to a kernel by value is copied out of the kernarg
// segment at the kernel's entry -- the compiler turns the copy into scalar loads of every field the kernel uses, all of them in
// the entry block, ~150 scalar registers of them for step_arena, which it then spills to lanes of vector registers
// (v_writelane) and fetches back one by one where the fields are used (v_readlane): 563 such instructions in the c2 kernel, a
// tenth of the vector instructions an arena's wavefronts issue, in a kernel the vector unit bounds.  Bound as references into
// the kernarg segment itself (constant address space: uniform s_load at the use, rematerializable), the same fields cost no
// vector instruction: 167 lane moves are left; c2 43.2 -> 45.0 M env-steps/s, c4 34.7 -> 36.5, c5 5.40 -> 5.60, and with the
// step kernels' STEP_FLAGS of the Makefile c2 45.7, c3 26.1 -> 26.8 (same box: profiles/r06_c2/ab_kernargs.txt).
// The views repeat the kernels' leading parameters: the kernarg segment lays arguments out like a struct's members.
#ifndef NAVSIM_KARG_VIEW
#define NAVSIM_KARG_VIEW 1
#endif
struct StepKernargs { navsim_config c; navsim_state st; navsim_step_io io; };
struct StepInstallKernargs { navsim_config c; navsim_state st; navsim_step_io io; StepInstall in; };
#if NAVSIM_KARG_VIEW
#define NAVSIM_KERNARGS(View, ...) const View& ka = *(const View*)__builtin_amdgcn_kernarg_segment_ptr()
#else
#define NAVSIM_KERNARGS(View, ...) const View ka = {__VA_ARGS__}
#endif
// The fused step.  One workgroup = one arena (template arguments: step_arena).  reset_only: bit 0 = a reset-only launch, bit 1 =
// ped_update_kernel has advanced the pedestrians, bits 2-3 = the NAVSIM_STEP_* part of navsim_step_part.
template <int BLOCK, bool PEDS, typename Field, int RULE, int RECT, bool PINL, bool FEAT = true>
// Wavefronts per SIMD the kernel is compiled for = its register budget.  The 256-thread pedestrian variants with the index
// rows in LDS are resident at FIVE workgroups per CU (plan_step: 30 KB of LDS each) -- five wavefronts per SIMD, 96 registers
// each; compiled for eight (64 registers, like every other variant, whose residency the wave slots bound) they spilled
// 36 bytes per lane.  c3: 24.65 -> 26.4 M env-steps/s (profiles/r05_c3/ab_waves.txt; 4 / 5 / 6: 26.3 / 26.4 / 25.9).
#ifndef NAVSIM_PEDS_WAVES_MIN
#define NAVSIM_PEDS_WAVES_MIN 5
#endif
#ifndef NAVSIM_NOPEDS_WAVES_MIN
#define NAVSIM_NOPEDS_WAVES_MIN 8
#endif
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((PEDS && PINL && RECT == 2 && BLOCK == 256) ? NAVSIM_PEDS_WAVES_MIN : (!PEDS ? NAVSIM_NOPEDS_WAVES_MIN : 8), 8)))
void navsim_step_kernel(navsim_config c_, navsim_state st_,
                                                            navsim_step_io io_, int reset_only,
                                                            const uint8_t* __restrict__ reset_mask,
                                                            unsigned dyn_lds_bytes, int park_lanes, unsigned rect_lds_offset) {
    NAVSIM_KERNARGS(StepKernargs, c_, st_, io_);
    const navsim_config& c = ka.c; const navsim_state& st = ka.st; const navsim_step_io& io = ka.io;
    const int peds_done = (reset_only >> 1) & 1;
    const int part = (reset_only >> 2) & 3;
    reset_only &= 1;
    // longest-first launch order (a scheduling hint: which arena a workgroup takes never changes a result)
    const int e = st.launch_order ? st.launch_order[blockIdx.x] : (int)blockIdx.x;
    if (e < 0) return;              // navsim_regen's first-observation launch: one workgroup per list slot, -1 = empty slot
    // navsim_step_part, NAVSIM_STEP_NOT_DUE: the other launch of the pair (navsim_step_due_kernel) steps the arenas with a
    // pedestrian that waited for navsim_replan when the previous step ended
    if (part == NAVSIM_STEP_NOT_DUE && st.ped_due_prev[e] != 0ull) return;
    step_arena<BLOCK, PEDS, Field, RULE, RECT, PINL, false, FEAT>(c, st, io, e, reset_only, peds_done, reset_mask, dyn_lds_bytes, park_lanes, rect_lds_offset);
}

// navsim_step_install: the step whose finished arenas install their staged worlds themselves (step_arena INSTALL).  Also every
// other launch of the step kernel on a state with slot tables (navsim_state.map_slot): plain steps (in.ready == NULL: nothing is
// installed) and reset-only launches.
template <int BLOCK, bool PEDS, typename Field, int RULE, int RECT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((PEDS && RECT == 2 && BLOCK == 256) ? NAVSIM_PEDS_WAVES_MIN : (!PEDS ? NAVSIM_NOPEDS_WAVES_MIN : 8), 8)))
void navsim_step_install_kernel(navsim_config c_, navsim_state st_, navsim_step_io io_, StepInstall in_, int reset_only,
                                const uint8_t* __restrict__ reset_mask,
                                unsigned dyn_lds_bytes, int park_lanes, unsigned rect_lds_offset) {
    NAVSIM_KERNARGS(StepInstallKernargs, c_, st_, io_, in_);
    const navsim_config& c = ka.c; const navsim_state& st = ka.st; const navsim_step_io& io = ka.io; const StepInstall& in = ka.in;
    const int e = st.launch_order ? st.launch_order[blockIdx.x] : (int)blockIdx.x;
    if (e < 0) return;
    step_arena<BLOCK, PEDS, Field, RULE, RECT, PEDS, true>(c, st, io, e, reset_only & 1, 0, reset_mask, dyn_lds_bytes, park_lanes, rect_lds_offset, &in);
}

// navsim_step_part, NAVSIM_STEP_DUE: the few arenas that waited for navsim_replan, as a COMPACT launch -- workgroup b steps the
// b-th, (b + gridDim.x)-th, ... arena whose word of st.ped_due_prev is set (due_arena_pick).  They end the step, so their
// wavefronts go first wherever they share a SIMD with the other part's.  A kernel of its own: the loop around the arena's body
// costs the body registers (round 5: inside navsim_step_kernel it took c2 from 92 to 114 us per launch).
template <int BLOCK, typename Field, int RULE, int RECT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void navsim_step_due_kernel(navsim_config c_, navsim_state st_,
                                                            navsim_step_io io_, unsigned dyn_lds_bytes, int park_lanes,
                                                            unsigned rect_lds_offset) {
    NAVSIM_KERNARGS(StepKernargs, c_, st_, io_);
    const navsim_config& c = ka.c; const navsim_state& st = ka.st; const navsim_step_io& io = ka.io;
    __builtin_amdgcn_s_setprio(2);
    for (int slot = (int)blockIdx.x;; slot += (int)gridDim.x) {
        const int e = due_arena_pick<BLOCK>(st.ped_due_prev, c.n_envs, slot);
        if (e < 0) return;
        step_arena<BLOCK, true, Field, RULE, RECT, true>(c, st, io, e, 0, 0, nullptr, dyn_lds_bytes, park_lanes, rect_lds_offset, nullptr, 2);
        __syncthreads();                                 // the arena's LDS is reused by the next one
    }
}

// navsim_step_replan: navsim_replan of the previous step's flags INSIDE the step's launch (round 5).  env.py:667-680 plans a
// new route inside step() for the pedestrian that reached its goal; as kernels of its own that search sat serially behind the
// step (round 4: 166 + 98 us per step on the c3 world) or beside it on a second stream (navsim_step_part: ~207 us -- two
// cross-stream waits per step and three launches sharing the chip).  Here the arena's own workgroup does it before it steps
// the arena, and those workgroups go FIRST: the launch opens with G front workgroups, front workgroup b takes the b-th arena
// with a waiting pedestrian (due_arena_pick), re-plans its pedestrians one after the other with all its threads
// (replan_one: the search's LDS is the step's own dynamic LDS, not yet in use) and steps the arena; the other workgroups step
// the arenas nobody waits in.  An arena of the front takes a search longer (~60-90 us) than the others and starts first: it is
// done before the launch's last generation is.  One launch, one stream, no flags to wait for; per arena the order is still
// step, replan, step.  The cap of navsim_replan (max_queries, in (arena, pedestrian) order, the rest wait and are counted)
// is kept through the pedestrians' ranks.  More waiting arenas than front workgroups: the arena's own back workgroup does it.
// which arena this workgroup steps, and whether it plans for it first (navsim_step_replan_kernel); false: nothing to do
template <int BLOCK>
__device__ __forceinline__ bool step_replan_pick(const navsim_config& c, const navsim_state& st, int G, int cap, int& e, bool& plan, int& before) {
    before = 0;
    plan = false;
    if ((int)blockIdx.x < G) {
        int n_peds;
        e = due_arena_pick<BLOCK>(st.ped_due_prev, c.n_envs, (int)blockIdx.x, &before, &n_peds);
        if (blockIdx.x == 0 && threadIdx.x == 0 && st.counters && n_peds > 0) {       // what navsim_replan counts
            const int served = n_peds < cap ? n_peds : cap;
            if (served > 0) atomicAdd(&st.counters[NAVSIM_COUNTER_REPLAN_SERVED], (unsigned long long)served);
            if (n_peds > served) atomicAdd(&st.counters[NAVSIM_COUNTER_REPLAN_UNSERVED], (unsigned long long)(n_peds - served));
        }
        if (e < 0) return false;
        plan = true;
    } else {
        const int b = (int)blockIdx.x - G;
        e = st.launch_order ? st.launch_order[b] : b;     // longest-first launch order (a scheduling hint)
        if (e < 0) return false;
        if (st.ped_due_prev[e] != 0ull) {                 // a front workgroup's arena -- unless more arenas wait than the front holds
            if (due_arena_rank<BLOCK>(st.ped_due_prev, e, &before) < G) return false;
            plan = true;
        }
    }
    return true;
}
template <int BLOCK, typename Field, int RULE, int RECT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((RECT == 2 && BLOCK == 256) ? NAVSIM_PEDS_WAVES_MIN : 8, 8)))
void navsim_step_replan_kernel(navsim_config c_, navsim_state st_, navsim_step_io io_, unsigned dyn_lds_bytes, int park_lanes,
                               unsigned rect_lds_offset, int G, int cap) {
    NAVSIM_KERNARGS(StepKernargs, c_, st_, io_);
    const navsim_config& c = ka.c; const navsim_state& st = ka.st; const navsim_step_io& io = ka.io;
    int e, before;
    bool plan;
    if (!step_replan_pick<BLOCK>(c, st, G, cap, e, plan, before)) return;
    if (plan) {
        for (unsigned long long m = st.ped_due_prev[e]; m != 0ull; m &= m - 1ull, ++before) {      // block-uniform
            if (before < cap) replan_one<BLOCK, 1>(c, st, e, (int)__builtin_ctzll(m));     // (one costmap word per thread: the host checked)
            __syncthreads();                             // the search's LDS is reused by the next one and by the step
        }
    }
    step_arena<BLOCK, true, Field, RULE, RECT, true>(c, st, io, e, 0, 0, nullptr, dyn_lds_bytes, park_lanes, rect_lds_offset);
}
// ... and with the install of the staged worlds (navsim_step_install with max_queries >= 0): the kernel of the pipelined reset
// path for worlds whose pedestrians follow planned routes
template <int BLOCK, typename Field, int RULE, int RECT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((RECT == 2 && BLOCK == 256) ? NAVSIM_PEDS_WAVES_MIN : 8, 8)))
void navsim_step_replan_install_kernel(navsim_config c_, navsim_state st_, navsim_step_io io_, StepInstall in_, unsigned dyn_lds_bytes,
                                       int park_lanes, unsigned rect_lds_offset, int G, int cap) {
    NAVSIM_KERNARGS(StepInstallKernargs, c_, st_, io_, in_);
    const navsim_config& c = ka.c; const navsim_state& st = ka.st; const navsim_step_io& io = ka.io; const StepInstall& in = ka.in;
    int e, before;
    bool plan;
    if (!step_replan_pick<BLOCK>(c, st, G, cap, e, plan, before)) return;
    if (plan) {
        for (unsigned long long m = st.ped_due_prev[e]; m != 0ull; m &= m - 1ull, ++before) {
            if (before < cap) replan_one<BLOCK, 1>(c, st, e, (int)__builtin_ctzll(m));
            __syncthreads();
        }
    }
    step_arena<BLOCK, true, Field, RULE, RECT, true, true>(c, st, io, e, 0, 0, nullptr, dyn_lds_bytes, park_lanes, rect_lds_offset, &in);
}

// navsim_launch_order: arenas by descending cost.  One workgroup: maximum, 1024-bucket histogram on the cost
// scaled to the maximum, exclusive scan from the expensive end, scatter.  Order inside a bucket is free.
__global__ __launch_bounds__(1024) void launch_order_kernel(const uint32_t* __restrict__ cost, int32_t* __restrict__ order,
                                                            int n) {
    __shared__ unsigned hist[1024], base[1024], wave_tot[16];
    __shared__ unsigned max_s;
    __shared__ unsigned long long sum_s;
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) { max_s = 1; sum_s = 0; }
    __syncthreads();
    unsigned mx = 0;
    unsigned long long sm = 0;
    for (int e = tid; e < n; e += 1024) { mx = cost[e] > mx ? cost[e] : mx; sm += cost[e]; }
    atomicMax(&max_s, mx);
    atomicAdd(&sum_s, sm);
    __syncthreads();
    // buckets span [0, min(max, 4 * mean)]: one outlier (a workgroup that was held up) must not squeeze every
    // other arena into a handful of buckets
    unsigned long long m = max_s;
    const unsigned long long cap = 4ull * (sum_s / (unsigned long long)n) + 1ull;
    m = m < cap ? m : cap;
    // float arithmetic (a 64-bit integer division per arena and pass was a third of this kernel); any monotone map
    // serves as long as both passes use the same one
    const float scale = 1023.0f / (float)m;
    auto bucket = [&](unsigned cst) {                                                                  // 0 = costliest
        const unsigned long long cc = cst < m ? cst : m;
        const int b = (int)((float)cc * scale);
        return 1023 - (b > 1023 ? 1023 : b);
    };
    for (int e = tid; e < n; e += 1024) atomicAdd(&hist[bucket(cost[e])], 1u);
    __syncthreads();
    // exclusive scan of the 1024 bucket counts: inside each wavefront by shuffles, the 16 wavefront totals by
    // wavefront 0 (two barriers; the ten-step scan through LDS with its twenty barriers was half of the kernel)
    const unsigned mine = hist[tid];
    unsigned incl = mine;
    const int lane = tid & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    if (tid < 64) {
        const unsigned t = tid < 16 ? wave_tot[tid] : 0u;
        unsigned ti = t;
        for (int off = 1; off < 16; off <<= 1) {
            const unsigned v = __shfl_up(ti, off, 64);
            if (tid >= off) ti += v;
        }
        if (tid < 16) wave_tot[tid] = ti - t;                   // exclusive over the wavefronts
    }
    __syncthreads();
    base[tid] = wave_tot[tid >> 6] + incl - mine;
    __syncthreads();
    for (int e = tid; e < n; e += 1024) order[atomicAdd(&base[bucket(cost[e])], 1u)] = e;
}
