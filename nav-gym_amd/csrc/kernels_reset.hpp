// kernels_reset.hpp -- reset path on the device: navsim_regen, costmap, planner, navsim_replan.
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).

// ============================================================================================
// navsim_regen: reset() of finished arenas on the device with a new random map.
// Specification: oracle/navsim_ref.c navsim_regen_cpu (same hash-keyed uniforms, same tries).
// ============================================================================================
// The reset kernels run a handful of workgroups once per call: what they cost is not arithmetic but the FETCH of
// their instructions (a cold 64-byte line per ~8 instructions from L2 / HBM: regen_maps_kernel's 600 straight-line
// instructions per thread took 13 us, profiles/_diag/regen_stamps.py) -- keep what a lane executes short.  (Real
// function calls are NOT the way: noinline helpers made regen_commit_kernel 38 -> 72 us through the call ABI's
// scratch traffic.)
// Ordered compaction of the arenas that finished in this step, without a kernel of its own: every workgroup of the
// first regen kernel finds ITS arena -- the b-th finished one in index order, or -1 -- from the done flags (a few
// hundred bytes to a few KB).  total = min(finished, cap): lowest indices first, the rest wait for the next call.
// Whole 256-thread workgroup; two barriers.
// `skip` (optional): arenas with skip[e] != 0 are not eligible (navsim_regen_swap: their staged world is not ready).
// excl_out / lo_out / hi_out (optional): this thread's slice of the arenas and the number of eligible ones before it.
// Eligibility beyond the done flag (round 5): cfg.regen_min_steps -- an arena whose episode lasted fewer steps restarts in
// place (done_steps: what the step recorded) -- and, for the pipelined swap, the staged world's completeness (ready[e] = the
// episode number the last finished staging pass generated for arena e, episode[e] = the one the arena now starts).
struct RegenRule { const int32_t* done_steps; int min_steps; const long long* ready; const int64_t* episode; };
__device__ __forceinline__ bool regen_long_enough(const RegenRule& r, int e) {
    return r.min_steps <= 0 || !r.done_steps || r.done_steps[e] >= r.min_steps;
}
__device__ __forceinline__ int regen_slot(const uint8_t* __restrict__ done, int E, int cap, int b, int& total,
                                          const uint8_t* __restrict__ skip = nullptr, int* excl_out = nullptr,
                                          int* lo_out = nullptr, int* hi_out = nullptr, int* all_out = nullptr,
                                          const RegenRule rule = RegenRule{nullptr, 0, nullptr, nullptr}) {
    __shared__ int wave_tot[4], found_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (E + 255) / 256;
    const int lo = tid * per, hi = (lo + per < E) ? lo + per : E;
    auto eligible = [&](int e) {
        return done[e] != 0 && !(skip && skip[e] != 0) && regen_long_enough(rule, e) &&
               !(rule.ready && rule.ready[e] != (long long)rule.episode[e]);
    };
    int n = 0;
    for (int e = lo; e < hi; ++e) n += eligible(e);
    int incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[tid >> 6] = incl;
    if (tid == 0) found_s = -1;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < 4; ++w) { before += (w < (tid >> 6)) ? wave_tot[w] : 0; all += wave_tot[w]; }
    incl += before;
    const int excl = incl - n;
    if (b >= excl && b < incl) {                              // exactly one thread (if any)
        int pos = excl;
        for (int e = lo; e < hi; ++e)
            if (eligible(e)) { if (pos == b) { found_s = e; break; } ++pos; }
    }
    __syncthreads();
    total = all < cap ? all : cap;
    if (excl_out) { *excl_out = excl; *lo_out = lo; *hi_out = hi; }
    if (all_out) *all_out = all;
    return b < total ? found_s : -1;
}

// create_indoor_map (map_generator.py:97-123; oracle regen_map_indoor): the corridor tree on the coarse grid,
// one workgroup per regenerated arena.  The tree grows one node per iteration (nearest node by a workgroup
// min-reduction on (L1 distance, node index), then the two corridor rectangles carved by all threads);
// the grid lives in LDS and is written to grid_all[b] (G*G bytes, stride 100*100).  kind[b] = G for a
// corridor map, 0 for an outdoor one (regen_maps_kernel then draws the outdoor map as before).
// navsim_state.counters: what a call served and what its cap left waiting (one thread of the opening kernel)
__device__ __forceinline__ void count_served(const navsim_state& st, int served_slot, int served, int unserved) {
    if (!st.counters) return;
    if (served > 0) atomicAdd(&st.counters[served_slot], (unsigned long long)served);
    if (unserved > 0) atomicAdd(&st.counters[served_slot + 1], (unsigned long long)unserved);
}

// counters[NAVSIM_COUNTER_REGEN_SHORT]: finished arenas whose episode was shorter than cfg.regen_min_steps (one 256-thread workgroup)
__device__ __forceinline__ void count_short(const navsim_config& c, const navsim_state& st, const uint8_t* __restrict__ done) {
    if (!st.counters || c.regen_min_steps <= 0 || !st.done_steps) return;
    int n = 0;
    for (int e = (int)threadIdx.x; e < c.n_envs; e += 256) n += done[e] != 0 && st.done_steps[e] < c.regen_min_steps;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if ((threadIdx.x & 63) == 0 && n > 0) atomicAdd(&st.counters[NAVSIM_COUNTER_REGEN_SHORT], (unsigned long long)n);
}

// The kernel also OPENS navsim_regen: workgroup b selects its arena (regen_slot), publishes list[b] (-1: none) and,
// workgroup 0, the count; every later kernel of the call reads those.  (Worlds of outdoor maps only skip this kernel:
// regen_maps_kernel opens the call itself.)
__global__ __launch_bounds__(256) void regen_indoor_kernel(navsim_config c, navsim_state st,
                                                           const uint8_t* __restrict__ done, int cap,
                                                           int* __restrict__ count, int* __restrict__ list,
                                                           uint8_t* __restrict__ grid_all, int* __restrict__ kind) {
    __shared__ uint8_t g[100 * 100];
    __shared__ int tx[152], ty[152];
    const int b = blockIdx.x, tid = threadIdx.x;
    int total, all;
    const RegenRule rule = {st.done_steps, c.regen_min_steps, nullptr, nullptr};
    const int e = regen_slot(done, c.n_envs, cap, b, total, nullptr, nullptr, nullptr, nullptr, &all, rule);
    if (tid == 0) { list[b] = e; if (b == 0) { *count = total; count_served(st, NAVSIM_COUNTER_REGEN_SERVED, total, all - total); } }
    if (b == 0) count_short(c, st, done);
    if (e < 0) return;
    const int size = c.map_w;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    const double* tape = rg_tape(st, e);
    if (tid == 0) regen_params(c, st, e);
    const bool indoor = c.regen_indoor_ratio > 0.0 &&       // env.py:295
                        rg_t(tape, NAVSIM_DRAW_KIND, rg_key(c.seed, genv, ep, 0x4B494E44ULL), 0) < c.regen_indoor_ratio;
    if (!indoor) { if (tid == 0) kind[b] = 0; return; }                  // block-uniform
    const uint64_t key = rg_key(c.seed, genv, ep, 0x494E44ULL);
    uint64_t n = 0;
    const int r = c.corridor_width_lo + (int)(rg_t(tape, NAVSIM_DRAW_CORRIDOR_WIDTH, key, n) * (double)(c.corridor_width_hi - c.corridor_width_lo + 1));
    const int it = c.iterations_lo + (int)(rg_t(tape, NAVSIM_DRAW_ITERATIONS, key, n + 1) * (double)(c.iterations_hi - c.iterations_lo + 1));
    n += 2;
    int G = size / 10;
    G = G < 2 * r + 8 ? 2 * r + 8 : G;
    G = G > 100 ? 100 : G;
    int n_it = (it * G * G + 5000) / 10000;
    n_it = n_it < 4 ? 4 : (n_it > 150 ? 150 : n_it);
    // create_indoor_map (map_generator.py:97-123) grows a tree: iteration k draws a point p_k, joins it to the NEAREST node so
    // far (nodes 0 .. k: the centre and p_0 .. p_k-1; ties to the oldest) by an L-shaped corridor and adds it to the tree.  The
    // points are draws, not results: every iteration's nearest node depends on the points alone, and carving only ever clears
    // cells -- so the iterations are independent.  Round 5: all points, then all nearest nodes (a thread per iteration), then
    // all corridors, three barriers in all.  (Rounds 3-4 ran the loop as written, three barriers and an LDS atomic per
    // iteration: 313 us per call at 1000 x 1000 cells -- the second-largest kernel of a step of the reference's own
    // configuration, profiles/r05_refdef/.  Same grid, cell for cell: the reset goldens replay the reference's generator.)
    __shared__ unsigned char coin_s[152], near_s[152];
    for (int k = tid; k < G * G; k += 256) g[k] = 1;
    const int span = G - 2 * r - 3;
    if (tid == 0) { tx[0] = G / 2; ty[0] = G / 2; }
    for (int k = tid; k < n_it; k += 256) {
        const uint64_t nk = n + 3 * (uint64_t)k;
        tx[k + 1] = r + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 3 * k, key, nk) * span);
        ty[k + 1] = r + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 3 * k + 1, key, nk + 1) * span);
        coin_s[k] = rg_t(tape, NAVSIM_DRAW_MAP + 3 * k + 2, key, nk + 2) >= 0.5;
    }
    __syncthreads();
    for (int k = tid; k < n_it; k += 256) {
        const int px = tx[k + 1], py = ty[k + 1];
        unsigned best = 0xFFFFFFFFu;
        for (int j = 0; j <= k; ++j) {
            const unsigned key_j = ((unsigned)(abs(px - tx[j]) + abs(py - ty[j])) << 8) | (unsigned)j;
            best = key_j < best ? key_j : best;
        }
        near_s[k] = (unsigned char)(best & 0xFFu);
    }
    if (tid == 0) g[(G / 2) * G + G / 2] = 0;
    __syncthreads();
    // (carving only clears cells: the corridors are independent too -- one per WAVEFRONT at a time, 64 lanes on its two
    //  rectangles; all 256 threads on one corridor after the other was 60 of the kernel's 85 us at 150 iterations)
    const int wave = tid >> 6, lane = tid & 63;
    for (int k = wave; k < n_it; k += 4) {
        const int px = tx[k + 1], py = ty[k + 1];
        const bool coin = coin_s[k] != 0;
        const int qx = tx[near_s[k]], qy = ty[near_s[k]];
        const int x1 = px < qx ? px : qx, x2 = px < qx ? qx : px;
        const int y1 = py < qy ? py : qy, y2 = py < qy ? qy : py;
        const bool constellation1 = (px > qx && py < qy) || (px < qx && py > qy);
        const int hx = coin ? x1 : x2;
        const int cy = coin ? (constellation1 ? y1 : y2) : (constellation1 ? y2 : y1);
        const int wh = y2 - y1 + 2 * r + 1, hv = x2 - x1 + 2 * r + 1, side = 2 * r + 1;
        for (int idx = lane; idx < side * wh; idx += 64) {
            int a = hx - r + idx / wh, bq = y1 - r + idx % wh;
            if (a >= 0 && a < G && bq >= 0 && bq < G) g[a * G + bq] = 0;
        }
        for (int idx = lane; idx < hv * side; idx += 64) {
            int a = x1 - r + idx / side, bq = cy - r + idx % side;
            if (a >= 0 && a < G && bq >= 0 && bq < G) g[a * G + bq] = 0;
        }
        if (lane == 0) g[px * G + py] = 0;
    }
    __syncthreads();
    uint8_t* out = grid_all + (size_t)b * 10000;
    for (int k = tid; k < G * G; k += 256) out[k] = g[k];
    if (tid == 0) kind[b] = G;
}

// `done` != NULL: the kernel OPENS the call (worlds of outdoor maps only: no regen_indoor_kernel launch): every
// workgroup finds the arena of its item from the done flags itself (regen_slot), the workgroup of a slot's first band
// publishes list[b], kind[b] = 0 and draws the per-episode parameters, workgroup 0 publishes the count and marks the
// empty slots.
__global__ __launch_bounds__(256) void regen_maps_kernel(navsim_config c, navsim_state st,
                                                         int* __restrict__ count, int* __restrict__ list,
                                                         uint8_t* __restrict__ occ_all,
                                                         const uint8_t* __restrict__ grid_all, int* __restrict__ kind,
                                                         char* __restrict__ field_scratch, size_t field_bytes,
                                                         float* __restrict__ ovf_scratch, int direct,
                                                         const uint8_t* __restrict__ done, int cap, uint4* __restrict__ rect_all,
                                                         char* __restrict__ index_all) {
    __shared__ int ocx[64], ocy[64];
    int n_items;
    if (done) {
        int total, all;
        (void)regen_slot(done, c.n_envs, cap, 0, total, nullptr, nullptr, nullptr, nullptr, &all,
                         RegenRule{st.done_steps, c.regen_min_steps, nullptr, nullptr});
        n_items = total * kRegenSlices;
        if (blockIdx.x == 0) {
            count_short(c, st, done);
            if (threadIdx.x == 0) { *count = total; count_served(st, NAVSIM_COUNTER_REGEN_SERVED, total, all - total); }
            for (int b = total + (int)threadIdx.x; b < cap; b += 256) list[b] = -1;
        }
    } else {
        n_items = *count * kRegenSlices;
    }
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {     // block-uniform loop
        const int b = item / kRegenSlices, slice = item % kRegenSlices;
        int e;
        if (done) {
            int total;
            e = regen_slot(done, c.n_envs, cap, b, total, nullptr, nullptr, nullptr, nullptr, nullptr,
                           RegenRule{st.done_steps, c.regen_min_steps, nullptr, nullptr});
            if (slice == 0 && threadIdx.x == 0) { list[b] = e; kind[b] = 0; regen_params(c, st, e); }
        } else {
            e = list[b];
        }
        regen_maps_item(c, st, b, slice, e, occ_all, grid_all, kind, field_scratch, field_bytes, ovf_scratch, direct,
                        done != nullptr, rect_all, index_all, ocx, ocy);       // (kind[b] of this call may not be written yet: not read then)
        __syncthreads();                                                 // ocx / ocy are rewritten by the next item
    }
}

// navsim_state.map_slot: the kernels below that are handed an array's base pointer and the list find a map through mlist[b],
// the slot that holds the map of arena list[b] (without a table the list itself serves)
__global__ __launch_bounds__(256) void regen_map_list_kernel(const int* __restrict__ list, const int32_t* __restrict__ map_slot,
                                                             int* __restrict__ mlist, int cap) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < cap) mlist[b] = list[b] >= 0 ? map_slot[list[b]] : -1;
}

// install the new distance field of every regenerated arena (kRegenSlices workgroups per map, 16-byte copies)
__global__ __launch_bounds__(256) void regen_field_kernel(char* __restrict__ dst_base, const int* __restrict__ count,
                                                          const int* __restrict__ list,
                                                          const char* __restrict__ field_scratch, size_t field_bytes) {
  const int n_items = *count * kRegenSlices;                 // bounded grid, like regen_maps_kernel
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kRegenSlices, slice = item % kRegenSlices;
    const int e = list[b], tid = threadIdx.x;
    const char* src_b = field_scratch + (size_t)b * field_bytes;
    char* dst_b = dst_base + (size_t)e * field_bytes;
    if (((field_bytes | (size_t)(uintptr_t)src_b | (size_t)(uintptr_t)dst_b) & 15) == 0) {
        const size_t n16 = field_bytes / 16;
        const size_t per = (n16 + kRegenSlices - 1) / kRegenSlices;
        const size_t lo = slice * per, hi = (lo + per < n16) ? lo + per : n16;
        const uint4* src = (const uint4*)src_b;
        uint4* dst = (uint4*)dst_b;
        for (size_t i = lo + tid; i < hi; i += 256) dst[i] = src[i];
    } else {                                              // odd map sizes: every field format is 2-byte granular
        const size_t n2 = field_bytes / 2;
        const size_t per = (n2 + kRegenSlices - 1) / kRegenSlices;
        const size_t lo = slice * per, hi = (lo + per < n2) ? lo + per : n2;
        for (size_t i = lo + tid; i < hi; i += 256) ((uint16_t*)dst_b)[i] = ((const uint16_t*)src_b)[i];
    }
  }
}

// draw the start / goal table, the robot (with reset()'s first-scan test) and the pedestrians of a regenerated arena.
// Round 3: 1024 threads = 16 wavefronts; a table entry or a pedestrian is ONE wavefront (rg_sample_wave: its 64 tries
// at once) and the first-scan test runs a beam per thread -- 68 us of dependent reads became what follows.
constexpr int kCommitBlock = 1024;
template <typename Field>
__global__ __launch_bounds__(kCommitBlock) void regen_commit_kernel(navsim_config c, navsim_state st,
                                                           const int* __restrict__ count, const int* __restrict__ list,
                                                           const char* __restrict__ field_scratch, size_t field_bytes,
                                                           const int* __restrict__ kind) {
    __shared__ double robot_xy[2];
    const int b = blockIdx.x;
    if (b >= *count) return;
    regen_commit_arena<Field>(c, st, list[b], live_size(c, kind, b), robot_xy);
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
// Round 5: the later rounds are SPECULATED.  A slot's candidate of round r is a function of (seed, arena, episode, r, slot) and
// the costmap alone -- not of what the earlier rounds found -- so all four candidates of every slot are drawn at once; the
// first round is planned, then rounds 1-3 of the slots it left open in ONE launch, and the first round that passes is taken:
// what the sequential loop of the reference (env.py:748-762, 786-804) and of the oracle ends with, after two search latencies
// instead of four.  (Rounds 3-4 ran {accept + draw, plan} x 4 per stage, 18 launches: 8 x 111-154 us of search per step of
// the reference's own configuration, profiles/r05_refdef/.)  Query index q = (b R + r) Q + k: slot b of the call, round r,
// table entry / pedestrian k.
constexpr int kRegenRounds = 4;
struct RegenPlanWs {
    uint8_t* cost;        // [M, Hc, Wc] scratch, or the resident st.costmap (then indexed by arena)
    int cost_by_arena;
    double* qstart;       // [M, R, Q, 2]
    double* qgoal;        // [M, R, Q, 2]
    double* qtheta;       // [M, R, Q]      robot stage: the heading drawn with the candidate
    double* qwp;          // [M, R, Q, P, 2]
    const int* kind;      // [M] regen_indoor_kernel's map kind (live map size)
    int32_t* qnwp;        // [M, R, Q]
    double* qlen;         // [M, R, Q]
    unsigned long long* qcut;   // [M, R, Q]  1 = the query's route was longer than P waypoints (counted for the round that is taken)
    uint8_t* active;      // [M, R, Q]
    uint8_t* res_robot;   // [M, K]
    int Q;
};

// robot stage: the four candidates (start, heading, goal) of every entry of the arena's start / goal table
__global__ __launch_bounds__(256) void regen_robot_sample_kernel(navsim_config c, navsim_state st,
                                                                 const int* __restrict__ count,
                                                                 const int* __restrict__ list, RegenPlanWs ws) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int K = c.n_spawn, Q = ws.Q;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const int live_c = live_size(c, ws.kind, b) / 5;         // candidates are cells of the live map's costmap
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = ws.cost + (size_t)(ws.cost_by_arena ? map_slot_of(c, st, e) : b) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    for (int idx = tid; idx < kRegenRounds * Q; idx += 256) {
        const int round = idx / Q, k = idx - round * Q;
        const size_t q = ((size_t)b * kRegenRounds + round) * Q + k;
        ws.qcut[q] = 0ull; ws.qnwp[q] = 0;                   // (a later round is only planned where the first one failed)
        if (k >= K) { ws.active[q] = 0; continue; }
        uint64_t key = rg_key(c.seed, genv, ep, 0x52504C00ULL + (uint64_t)round * 256 + (uint64_t)k), n = 0;
        double x, y, gx, gy;
        rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, n, 0, 0, 0, 0, 0, x, y);
        rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, n, 2, x, y, c.min_goal_dist, c.max_goal_dist, gx, gy);
        ws.qtheta[q] = nv::kTwoPi * rg_u(key, n++);
        ws.qstart[2 * q] = x; ws.qstart[2 * q + 1] = y;
        ws.qgoal[2 * q] = gx; ws.qgoal[2 * q + 1] = gy;
        ws.active[q] = 1;
    }
}

// robot stage: every table entry takes the candidate of the first round whose path exists and is short enough
// (env.py:756-762), else the last round's; then the robot is picked and the pedestrians' parameters are drawn
template <typename Field>
__global__ __launch_bounds__(256) void regen_robot_accept_kernel(navsim_config c, navsim_state st,
                                                                 const int* __restrict__ count,
                                                                 const int* __restrict__ list, RegenPlanWs ws) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, K = c.n_spawn, Q = ws.Q;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    double* sp = (double*)st.spawn_pose + (size_t)e * K * 3;
    double* sg = (double*)st.spawn_goal + (size_t)e * K * 2;
    for (int k = tid; k < K; k += 256) {
        int take = kRegenRounds - 1;
        uint8_t res = 0;
        for (int round = 0; round < kRegenRounds; ++round) {
            const size_t q = ((size_t)b * kRegenRounds + round) * Q + k;
            if (ws.qnwp[q] > 0 && rg_robot_path_ok(ws.qlen[q], ws.qstart[2 * q], ws.qstart[2 * q + 1], ws.qgoal[2 * q], ws.qgoal[2 * q + 1])) {
                take = round; res = 1;
                break;
            }
        }
        const size_t q = ((size_t)b * kRegenRounds + take) * Q + k;
        sp[3 * k] = ws.qstart[2 * q]; sp[3 * k + 1] = ws.qstart[2 * q + 1]; sp[3 * k + 2] = ws.qtheta[q];
        sg[2 * k] = ws.qgoal[2 * q]; sg[2 * k + 1] = ws.qgoal[2 * q + 1];
        ws.res_robot[(size_t)b * K + k] = res;
    }
    __threadfence_block();
    __syncthreads();
    int idx = (int)(rg_key(c.seed, genv, ep, 0x5eedULL) % (uint64_t)K);
    {   // first resolved pair from idx on (cyclic) whose first scan is outside the discomfort zone (env.py:776-781);
        // none: the first resolved one; none resolved: idx.  Block-uniform control flow.
        const Field f(st.field, st.field_overflow, map_slot_of(c, st, e), c.map_h, c.map_w);
        const uint8_t* res = ws.res_robot + (size_t)b * K;
        int first_res = -1, pick = -1;
        for (int s_ = 0; s_ < K && pick < 0; ++s_) {
            const int j = (idx + s_) % K;
            if (!res[j]) continue;
            if (first_res < 0) first_res = j;
            if (!c.regen_check_discomfort || !spawn_in_discomfort(c, st, f, sp + 3 * j)) pick = j;
        }
        idx = pick >= 0 ? pick : (first_res >= 0 ? first_res : idx);
    }
    if (tid == 0) {
        double* rp = st.robot_pose + 3 * (size_t)e;
        rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
        st.robot_goal[2 * e] = sg[2 * idx]; st.robot_goal[2 * e + 1] = sg[2 * idx + 1];
    }
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    for (int i = tid; i < n; i += 256) {
        size_t q = (size_t)e * N + i;
        uint64_t k0 = rg_key(c.seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
        st.ped_pose[q * 3 + 2] = nv::kTwoPi * rg_u(k0, m++);
        ((double*)st.ped_v_pref)[q] = c.v_pref_lo + (c.v_pref_hi - c.v_pref_lo) * rg_u(k0, m++);
        ((uint8_t*)st.ped_has_legs)[q] = rg_u(k0, m++) < c.has_legs_ratio;
        st.ped_vel[q * 2] = 0.0; st.ped_vel[q * 2 + 1] = 0.0;
    }
}

// pedestrian stage: the four candidates (start at least ped_min_robot_dist from the robot, goal) of every pedestrian
__global__ __launch_bounds__(256) void regen_ped_sample_kernel(navsim_config c, navsim_state st,
                                                               const int* __restrict__ count,
                                                               const int* __restrict__ list, RegenPlanWs ws) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, Q = ws.Q;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const int live_c = live_size(c, ws.kind, b) / 5;
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = ws.cost + (size_t)(ws.cost_by_arena ? map_slot_of(c, st, e) : b) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    const double rx = st.robot_pose[3 * (size_t)e], ry = st.robot_pose[3 * (size_t)e + 1];
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    if (tid == 0 && st.ped_due) st.ped_due[e] = 0ull;      // new pedestrians: nobody waits for navsim_replan
    for (int idx = tid; idx < kRegenRounds * Q; idx += 256) {
        const int round = idx / Q, i = idx - round * Q;
        const size_t q = ((size_t)b * kRegenRounds + round) * Q + i;
        ws.qcut[q] = 0ull; ws.qnwp[q] = 0;
        if (i >= n) { ws.active[q] = 0; continue; }
        uint64_t key = rg_key(c.seed, genv, ep, 0x50504C00ULL + (uint64_t)round * 256 + (uint64_t)i), nn = 0;
        double x, y, gx, gy;
        rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, nn, 1, rx, ry, c.ped_min_robot_dist, 0, x, y);
        rgp_cell(c, cost, Wc, live_c, live_c, res_c, key, nn, 2, x, y, c.ped_min_goal_dist, 1.0e300, gx, gy);
        ws.qstart[2 * q] = x; ws.qstart[2 * q + 1] = y;
        ws.qgoal[2 * q] = gx; ws.qgoal[2 * q + 1] = gy;
        ws.active[q] = 1;
    }
}

// pedestrian stage: every pedestrian takes the candidate of the first round whose path exists, with its waypoints every 2 m
// (env.py:788-804); none: the last round's start and its goal as the only waypoint
__global__ __launch_bounds__(256) void regen_ped_accept_kernel(navsim_config c, navsim_state st,
                                                               const int* __restrict__ count,
                                                               const int* __restrict__ list, RegenPlanWs ws) {
    const int b = blockIdx.x;
    if (b >= *count) return;
    const int e = list[b], tid = threadIdx.x;
    const int N = c.max_peds, Q = ws.Q, P = c.max_waypoints;
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    __shared__ int take_s[NAVSIM_MAX_PEDS];
    for (int i = tid; i < n; i += 256) {
        int take = -1;
        for (int round = 0; round < kRegenRounds && take < 0; ++round)
            if (ws.qnwp[((size_t)b * kRegenRounds + round) * Q + i] > 0) take = round;
        take_s[i] = take;
        const size_t q = ((size_t)b * kRegenRounds + (take < 0 ? kRegenRounds - 1 : take)) * Q + i;
        const size_t pq = (size_t)e * N + i;
        st.ped_pose[pq * 3] = ws.qstart[2 * q]; st.ped_pose[pq * 3 + 1] = ws.qstart[2 * q + 1];
        st.ped_n_waypoints[pq] = take < 0 ? 1 : ws.qnwp[q];
        st.ped_wp_head[pq] = 0;
        if (st.ped_goal) { st.ped_goal[pq * 2] = ws.qgoal[2 * q]; st.ped_goal[pq * 2 + 1] = ws.qgoal[2 * q + 1]; }
        if (take < 0) {
            double* w = st.ped_waypoints + (pq * P) * 2;
            w[0] = ws.qgoal[2 * q]; w[1] = ws.qgoal[2 * q + 1];
        } else if (ws.qcut[q] && st.counters) {
            atomicAdd(&st.counters[NAVSIM_COUNTER_ROUTES_CUT], 1ull);      // a route stored cut (include/navsim.h)
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * P; idx += 256) {             // the taken rounds' waypoints into the state
        const int i = idx / P, k = idx - i * P;
        const int take = take_s[i];
        if (take < 0) continue;
        const size_t q = ((size_t)b * kRegenRounds + take) * Q + i;
        if (k >= ws.qnwp[q]) continue;
        const size_t pq = (size_t)e * N + i;
        double* w = st.ped_waypoints + (pq * P) * 2;
        w[2 * k] = ws.qwp[(q * P + k) * 2]; w[2 * k + 1] = ws.qwp[(q * P + k) * 2 + 1];
    }
}

// plan the candidates of a stage; ped_stage: waypoints every 2 m, else every 5 m (env.py:349-354).  pass 0: the first round's
// candidates; pass 1: the later rounds' candidates of the slots whose first round failed -- all three at once.  (Speculating
// all four rounds of every slot in one launch was measured and dropped: a launch of a few hundred searches is bound by
// their latency, one of four times as many by their arithmetic -- 2 x 507 us against 8 x 111, profiles/r05_refdef/.)
// A workgroup per (slot, round, entry) item of this stage, Qs entries per arena (table entries / pedestrians); idle ones cost
// ~10 ns each.  (Measured and dropped: a bounded launch whose workgroups loop over the items of the arenas that really
// finished -- the loop around the search costs its registers: 162 -> 226 us per launch, profiles/r05_refdef/.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK == 512 ? 8 : 1, 8))) void regen_plan_kernel(navsim_config c, navsim_state st, const int* __restrict__ count,
                                                         const int* __restrict__ list, RegenPlanWs ws, int ped_stage, int pass, int Qs) {
    const int item = (int)blockIdx.x;
    const int b = item / (kRegenRounds * Qs), rem = item - b * (kRegenRounds * Qs), round = rem / Qs, k = rem - round * Qs;
    if (b >= *count) return;
    const size_t q = ((size_t)b * kRegenRounds + round) * ws.Q + k;
    if (!ws.active[q] || (pass == 0) != (round == 0)) return;      // uniform per workgroup
    if (pass) {                                          // did the slot's first round pass?  (the accept kernels' own tests)
        const size_t q0 = q - (size_t)round * ws.Q;
        const bool ok0 = ws.qnwp[q0] > 0 && (ped_stage || rg_robot_path_ok(ws.qlen[q0], ws.qstart[2 * q0], ws.qstart[2 * q0 + 1],
                                                                            ws.qgoal[2 * q0], ws.qgoal[2 * q0 + 1]));
        if (ok0) return;
    }
    const int Hc = c.map_h / 5, Wc = c.map_w / 5, P = c.max_waypoints;
    plan_query<BLOCK, BLOCK == 512 ? 2 : kPlanMaxWpt>(ws.cost + (size_t)(ws.cost_by_arena ? map_slot_of(c, st, list[b]) : b) * Hc * Wc, Hc, Wc, c.resolution * 5.0, c.origin_x, c.origin_y,
                      ws.qstart[2 * q], ws.qstart[2 * q + 1], ws.qgoal[2 * q], ws.qgoal[2 * q + 1], ped_stage ? 2.0 : 5.0, P,
                      ws.qwp + q * P * 2, ws.qnwp + q, nullptr, ped_stage ? nullptr : ws.qlen + q, ped_stage ? ws.qcut + q : nullptr);
}

// --------------------------------------------------------------------------------------------
// navsim_replan (env.py:667-680; oracle navsim_replan_cpu): ordered list of the pedestrians standing on
// their final waypoint, then one workgroup per listed pedestrian: draw a goal, plan, up to 4 rounds.
// --------------------------------------------------------------------------------------------
// one wavefront per arena: bit i of due[e] = pedestrian i stands within 0.5 m of its final waypoint
__global__ __launch_bounds__(64) void replan_flag_kernel(navsim_config c, navsim_state st, uint64_t* __restrict__ due) {
    const int e = blockIdx.x, i = threadIdx.x, N = c.max_peds, P = c.max_waypoints;
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

// The b-th flagged pedestrian in (arena, pedestrian) order, as q = e N + i, or -1 when fewer are flagged; total = how many
// are.  Every workgroup of replan_kernel finds its own query this way (round 5): the ordered compaction used to be a
// kernel of its own (replan_select_kernel, one workgroup) in front of the searches -- 5 us on an idle chip, 33 us beside a
// step kernel that fills it, and serial either way.  32 KB of flags per 4096 arenas, read from L2 by every workgroup.
template <int kReplanBlock>
__device__ __forceinline__ int replan_pick(const uint64_t* __restrict__ due, int E, int N, int b, int& total) {
    __shared__ int wave_tot[kReplanBlock / 64], q_s, total_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (E + kReplanBlock - 1) / kReplanBlock;
    const int lo = tid * per, hi = (lo + per < E) ? lo + per : E;
    int n = 0;
    for (int e = lo; e < hi; ++e) n += __popcll(due[e]);
    int incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[tid >> 6] = incl;
    if (tid == 0) q_s = -1;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < kReplanBlock / 64; ++w) { if (w < (tid >> 6)) before += wave_tot[w]; all += wave_tot[w]; }
    int pos = before + incl - n;                          // flagged pedestrians in front of this thread's words
    if (b >= pos && b < pos + n)
        for (int e = lo; e < hi; ++e) {
            uint64_t m = due[e];
            const int k = __popcll(m);
            if (b < pos + k) {
                for (int skip = b - pos; skip > 0; --skip) m &= m - 1;
                q_s = e * N + (__ffsll((unsigned long long)m) - 1);
                break;
            }
            pos += k;
        }
    if (tid == 0) total_s = all;
    __syncthreads();
    total = total_s;
    return q_s;
}

template <int kReplanBlock>
__global__ __launch_bounds__(kReplanBlock) void replan_kernel(navsim_config c, navsim_state st, const uint64_t* __restrict__ due,
                                                              int cap) {
    // a handful of small workgroups beside a step kernel that keeps every SIMD's issue slots busy: first in line
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, N = c.max_peds, tid = threadIdx.x;
    int total;
    const int q = replan_pick<kReplanBlock>(due, c.n_envs, N, b, total);
    if (b == 0 && tid == 0) {
        const int served = total < cap ? total : cap;
        count_served(st, NAVSIM_COUNTER_REPLAN_SERVED, served, total - served);
    }
    if (q < 0 || b >= cap) return;                        // (cap = 0: the one workgroup of the launch only counts)
    replan_one<kReplanBlock>(c, st, q / N, q - (q / N) * N);
}

// --------------------------------------------------------------------------------------------
// navsim_debug_spawn_decisions (tests only; oracle navsim_spawn_decisions_cpu): the spawn loops' acceptance rules
// on SUPPLIED candidates, one workgroup per candidate, through the very functions the samplers above call.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spawn_decisions_kernel(navsim_config c, const uint8_t* __restrict__ cost, int Hc, int Wc,
                                                              const int32_t* __restrict__ kind, const double* __restrict__ start,
                                                              const double* __restrict__ goal, const double* __restrict__ robot,
                                                              double* __restrict__ wp_scratch, int32_t* __restrict__ code) {
    __shared__ int32_t nwp_s;
    __shared__ double plen_s;
    const int m = blockIdx.x, tid = threadIdx.x;
    const double sx = start[2 * m], sy = start[2 * m + 1], gx = goal[2 * m], gy = goal[2 * m + 1];
    const bool ped = kind[m] == 1;
    int rc = 0;
    if (ped && robot && !rg_start_ok(sx, sy, robot[2 * m], robot[2 * m + 1], c.ped_min_robot_dist)) rc = 1;
    else if (!rg_goal_ok(sx, sy, gx, gy, ped ? c.ped_min_goal_dist : c.min_goal_dist, ped ? 1.0e300 : c.max_goal_dist)) rc = 2;
    if (rc) { if (tid == 0) code[m] = rc; return; }                      // uniform per workgroup
    plan_query(cost, Hc, Wc, c.resolution * 5.0, c.origin_x, c.origin_y, sx, sy, gx, gy, ped ? 2.0 : 5.0,
               c.max_waypoints, wp_scratch + (size_t)m * c.max_waypoints * 2, &nwp_s, nullptr, &plen_s);
    __syncthreads();
    if (tid == 0) code[m] = nwp_s <= 0 ? 3 : ((!ped && !rg_robot_path_ok(plen_s, sx, sy, gx, gy)) ? 4 : 0);
}

// --------------------------------------------------------------------------------------------
// navsim_regen_swap (round 3): navsim_regen taken off the step's critical path.  The world an arena gets at the end of
// its episode depends on (seed, global arena, episode number) only -- never on how the episode went -- so it is
// generated AHEAD of time into a second, staged navsim_state (the ordinary navsim_regen on that state, episode + 1,
// on a side stream, while steps run).  When an arena finishes, this kernel installs the staged world: the arena's rows
// of every array navsim_regen writes are copied from the staged state into the live one, the first observation
// included, and the arena is marked (want[e] = 1, staged episode + 1) for the next staging pass.  Same selection rule
// as navsim_regen (lowest indices first, at most cap), same resulting state, bit for bit.
// Grid (cap, kRegenSlices): slice 0 of a slot moves the small arrays, all slices share the large buffers.
// --------------------------------------------------------------------------------------------
struct SwapBig { char* dst; const char* src; size_t bytes; };          // per-arena stride = bytes
// mark[]: one byte per arena, set and consumed through the 32-bit word that holds it (atomics), so that a swap may set a flag
// while a staging pass merges the array (the pipelined form): stage_request, kernels_step.hpp
__global__ __launch_bounds__(256) void regen_swap_kernel(navsim_config c, navsim_state live, navsim_state stage,
                                                         navsim_step_io io, const float* __restrict__ stage_obs,
                                                         const uint8_t* __restrict__ want, uint8_t* __restrict__ mark,
                                                         const long long* __restrict__ ready,
                                                         int cap, SwapBig b0, SwapBig b1, SwapBig b2, SwapBig b3, SwapBig b4) {
    const int b = blockIdx.x, tid = threadIdx.x;                      // (grid (cap, slices): opt-in path, not bounded yet)
    int total, excl, lo, hi;
    // Pipelined form (ready != NULL): eligible = finished, its episode lasted cfg.regen_min_steps steps (the simulation's rule)
    // and the staged world is the one for the episode that starts now (the safety net); want[] is not consulted -- a staging
    // pass may be merging it right now.
    const RegenRule rule = {live.done_steps, c.regen_min_steps, ready, live.episode};
    // eligible: finished AND staged (want[e] == 0).  want[] is only READ here -- every workgroup of the launch must see
    // the same flags to agree on the selection; what this launch decides is written to mark[] and merged into want[] by
    // the staging pass (navsim_regen_stage), which the next swap waits for.
    const int e = regen_slot(io.done, c.n_envs, cap, b, total, ready ? nullptr : want, &excl, &lo, &hi, nullptr, rule);
    if (b == 0 && blockIdx.y == 0) {
        // a finished arena that is not installed now plays its next episode in place (the step respawned it on the old
        // map and advanced episode[e]); the world staged for it carries a stale episode number: stage it again
        int pos = excl, n_in = 0, n_out = 0, n_short = 0, n_late = 0;
        for (int a = lo; a < hi; ++a) {
            if (io.done[a] == 0) continue;
            const bool lng = regen_long_enough(rule, a);
            const bool rdy = ready ? ready[a] == (long long)live.episode[a] : want[a] == 0;
            const bool elig = lng && rdy;
            n_short += !lng; n_late += lng && !rdy;
            const bool installed = elig && pos < cap;
            pos += elig;
            if (!installed) stage_request(stage.episode, mark, a, live.episode[a] + 1);      // (kernels_step.hpp: number first, then the flag)
            n_in += installed; n_out += !installed && elig;
        }
        count_served(live, NAVSIM_COUNTER_REGEN_SERVED, n_in, n_out);
        if (live.counters && n_short) atomicAdd(&live.counters[NAVSIM_COUNTER_REGEN_SHORT], (unsigned long long)n_short);
        if (live.counters && n_late) atomicAdd(&live.counters[NAVSIM_COUNTER_REGEN_LATE], (unsigned long long)n_late);
    }
    if (e < 0) return;
    if (live.map_slot && stage.map_slot && blockIdx.y == 0 && tid == 0) {       // the maps change places where they lie (navsim_state.map_slot)
        const int32_t a = live.map_slot[e], b2_ = stage.map_slot[e];
        live.map_slot[e] = b2_; stage.map_slot[e] = a;
    }
    const SwapBig big[5] = {b0, b1, b2, b3, b4};
    for (int k = 0; k < 5; ++k) {                            // field, overflow plane, rect records, costmap, rect index rows
        if (!big[k].dst) continue;
        const size_t n16 = big[k].bytes / 16;
        const char* src = big[k].src + (size_t)e * big[k].bytes;
        char* dst = big[k].dst + (size_t)e * big[k].bytes;
        if ((((size_t)(uintptr_t)src | (size_t)(uintptr_t)dst | big[k].bytes) & 15) == 0) {
            const size_t per = (n16 + gridDim.y - 1) / gridDim.y;
            const size_t lo = blockIdx.y * per, hi = (lo + per < n16) ? lo + per : n16;
            for (size_t i = lo + tid; i < hi; i += 256) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        } else {
            const size_t per = (big[k].bytes + gridDim.y - 1) / gridDim.y;
            const size_t lo = blockIdx.y * per, hi = (lo + per < big[k].bytes) ? lo + per : big[k].bytes;
            for (size_t i = lo + tid; i < hi; i += 256) dst[i] = src[i];
        }
    }
    // the small arrays, by slice 0 (spreading them over the slices, one array each, was measured: the kernel went from 22 to
    // 32 us on c5 -- what it costs is the 4096 workgroups that each select their arena, not these rows)
    if (blockIdx.y != 0) return;
    const int N = c.max_peds, K = c.n_spawn, P = c.max_waypoints, D = c.n_scan_stack * c.n_beams + NAVSIM_OBS_TAIL;
    auto row = [&](auto* dst, const auto* src, size_t n) {    // n elements of arena e
        if (!dst || !src) return;
        for (size_t i = tid; i < n; i += 256) dst[(size_t)e * n + i] = src[(size_t)e * n + i];
    };
    row(live.scan_noise_std, stage.scan_noise_std, 1);
    row(live.robot_pose, stage.robot_pose, 3);
    row(live.robot_goal, stage.robot_goal, 2);
    row(live.prev_action, stage.prev_action, 2);
    row(live.prev_pose, stage.prev_pose, 3);
    row(live.n_hist, stage.n_hist, 1);
    row(live.steps, stage.steps, 1);
    row((double*)live.spawn_pose, stage.spawn_pose, (size_t)K * 3);
    row((double*)live.spawn_goal, stage.spawn_goal, (size_t)K * 2);
    if (c.ped_model != NAVSIM_PED_NONE) {
        row(live.n_peds, stage.n_peds, 1);
        row(live.ped_pose, stage.ped_pose, (size_t)N * 3);
        row(live.ped_vel, stage.ped_vel, (size_t)N * 2);
        row(live.ped_prev_yaw, stage.ped_prev_yaw, N);
        row(live.ped_dist, stage.ped_dist, (size_t)N * 3);
        row((double*)live.ped_v_pref, stage.ped_v_pref, N);
        row((uint8_t*)live.ped_has_legs, stage.ped_has_legs, N);
        row(live.ped_waypoints, stage.ped_waypoints, (size_t)N * P * 2);
        row(live.ped_n_waypoints, stage.ped_n_waypoints, N);
        row(live.ped_wp_head, stage.ped_wp_head, N);
        row(live.ped_goal, stage.ped_goal, (size_t)N * 2);
    }
    row(io.obs, stage_obs, D);
    if (tid == 0) {
        if (io.achieved_goal) { io.achieved_goal[2 * e] = (float)stage.robot_pose[3 * (size_t)e]; io.achieved_goal[2 * e + 1] = (float)stage.robot_pose[3 * (size_t)e + 1]; }
        if (io.desired_goal) { io.desired_goal[2 * e] = (float)stage.robot_goal[2 * (size_t)e]; io.desired_goal[2 * e + 1] = (float)stage.robot_goal[2 * (size_t)e + 1]; }
        if (live.ped_due) live.ped_due[e] = 0ull;           // new pedestrians: nobody waits for navsim_replan
    }
    // the world after THIS one -- requested when nothing of the staged world is read any more: in the pipelined form a staging
    // pass may take the flag while this launch still runs (round-5 advisor).  This workgroup's own copies are behind the barrier;
    // the other slices' are not, which is why navsim_regen_swap launches the pipelined form without slot tables as ONE slice
    // per arena (with slot tables the slices have nothing to copy; the plain form's passes are ordered behind the swap by events).
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        stage_request(stage.episode, mark, e, live.episode[e] + 1);
    }
}

// opens a staging pass: what the last swap decided becomes part of want[]
__global__ __launch_bounds__(256) void regen_merge_want_kernel(uint8_t* __restrict__ want, uint8_t* __restrict__ mark, int E,
                                                               long long* __restrict__ ready, const int64_t* __restrict__ stage_episode,
                                                               int part = 0, int n_parts = 1) {
    const int w = blockIdx.x * 256 + threadIdx.x;            // one 32-bit word of mark[] = four arenas
    if (4 * w >= E) return;
    if (n_parts > 1 && w % n_parts != part) return;          // navsim_regen_stage_part: this pass's share of the arenas (by mark word)
    const unsigned m = atomicExch((unsigned*)mark + w, 0u);  // (a swap may be setting flags of this word right now)
    __threadfence();
    for (int k = 0; k < 4; ++k) {
        const int e = 4 * w + k;
        if (e >= E) break;
        if ((m >> (8 * k)) & 0xFFu) want[e] = 1;
        // the episode numbers this pass generates for, as they stand NOW (ready[E + e]): a swap that re-marks an arena while the
        // pass runs changes stage_episode[e] under it -- the world it leaves is then for no episode at all, and must not be
        // recorded as the new number's (the arena is marked again and staged by the next pass)
        if (ready) ready[E + e] = (long long)__hip_atomic_load(&stage_episode[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// closes a staging pass (navsim_regen on the staged state with want[] as its done flags): the arenas it served are staged,
// for the episode numbers the pass started with
__global__ __launch_bounds__(256) void regen_clear_want_kernel(const int* __restrict__ count, const int* __restrict__ list,
                                                               uint8_t* __restrict__ want, long long* __restrict__ ready, int E) {
    for (int b = threadIdx.x; b < *count; b += 256) {
        const int e = list[b];
        want[e] = 0;
        if (ready) ready[e] = ready[E + e];
    }
}

// navsim_restart: reset() of the arenas of `mask`, part one (env.py:730-746) -- the next start / goal pair of the arena's table
// and the next episode number: what the step does at `done` under auto-reset, as a call of its own
__global__ __launch_bounds__(256) void restart_kernel(navsim_config c, navsim_state st, const uint8_t* __restrict__ mask) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= c.n_envs || !mask[e]) return;
    const uint64_t genv = (uint64_t)(c.env_index_base + e);
    const uint64_t h = nv::hash4(c.seed, genv, (uint64_t)st.episode[e], 0x5eedULL);
    const int idx = (int)(h % (uint64_t)c.n_spawn);
    const double* sp = st.spawn_pose + ((size_t)e * c.n_spawn + idx) * 3;
    const double* sg = st.spawn_goal + ((size_t)e * c.n_spawn + idx) * 2;
    st.robot_pose[3 * (size_t)e] = sp[0]; st.robot_pose[3 * (size_t)e + 1] = sp[1]; st.robot_pose[3 * (size_t)e + 2] = sp[2];
    st.robot_goal[2 * (size_t)e] = sg[0]; st.robot_goal[2 * (size_t)e + 1] = sg[1];
    if (st.done_steps) st.done_steps[e] = (int32_t)st.steps[e];
    st.episode[e] += 1;
    st.steps[e] = 0;
}
