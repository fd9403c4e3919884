// kernels_plan.hpp -- the planner: shortest 4-connected paths on the 0.25 m costmap and their waypoints (pyastar2d.astar_path +
// path_to_waypoints, env.py:343-354, 1261-1277), the goal draw and the re-plan of one pedestrian (env.py:667-680).  Shared by
// the reset path (kernels_reset.hpp: navsim_plan, navsim_regen with cfg.regen_plan, navsim_replan) and by the fused step that
// re-plans inside its own launch (kernels_step.hpp navsim_step_replan_kernel).  Included inside a translation unit's
// anonymous namespace; not a standalone header.

// hash-keyed uniforms of the reset path (oracle/navsim_ref.c rg_u / nvr_hash4: same keys, same order)
__device__ __forceinline__ double rg_u(uint64_t key, uint64_t i) {
    return (double)(nv::mix64(key + i * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ uint64_t rg_key(uint64_t seed, uint64_t genv, uint64_t ep, uint64_t purpose) {
    return nv::hash4(seed, genv, ep, purpose);
}

// The acceptance rules of _sample_start_goal_path (env.py:366-383) -- oracle rg_start_ok / rg_goal_ok /
// rg_robot_path_ok; navsim_debug_spawn_decisions evaluates these very functions on supplied candidates.
__device__ __forceinline__ bool rg_start_ok(double x, double y, double rx, double ry, double min_robot) {
    const double ddx = rx - x, ddy = ry - y;
    return !(sqrt(ddx * ddx + ddy * ddy) < min_robot);                 // env.py:371-373: dropped when dist < 4
}
__device__ __forceinline__ bool rg_goal_ok(double sx, double sy, double gx, double gy, double dmin, double dmax) {
    const double ddx = sx - gx, ddy = sy - gy;
    const double dist = sqrt(ddx * ddx + ddy * ddy);
    return dmin < dist && dist < dmax;                                 // env.py:379
}
__device__ __forceinline__ bool rg_robot_path_ok(double plen, double sx, double sy, double gx, double gy) {
    const double ddx = gx - sx, ddy = gy - sy;
    return !(plen > 2.0 * sqrt(ddx * ddx + ddy * ddy));                // env.py:761
}


// One query, executed by a whole workgroup: shortest 4-connected path from the start's to the goal's costmap cell and its
// waypoints (pyastar2d.astar_path + path_to_waypoints, env.py:343-354, 1261-1277; tie-break: oracle/navsim_ref.c).
// c = this query's costmap, w = its waypoint row (max_wp x 2); n_wp / path_cells / path_len point at its slots.
//
// The search is level-synchronous from the GOAL on bitmaps: a row of the costmap is ceil(Wc / 64) 64-bit words (bit b of word w
// = cell i = 64 w + b), every thread owns WPT of the words (free cells, what is still unreached, and the two DIRECTION planes of
// its cells in registers); a level is, per word, (frontier << 1 | frontier >> 1 | the words above and below) & unreached -- five
// LDS reads, a dozen bit operations, one LDS write, one barrier.
// Round 5: no hop field.  Rounds 1-4 wrote every reached cell's hop count (int16 dist[], a divergent loop over the new bits of
// every word in every level) and the walk looked for "the first neighbour one hop closer, in the order +i, -i, +j, -j".  A
// neighbour is one hop closer exactly when it belongs to the frontier of the level BEFORE the cell's own, and that is what
// the level's update has in its hands: the cell reached through (frontier >> 1) has its +i neighbour in the frontier, through
// (frontier << 1) its -i neighbour, through the word below / above its +j / -j neighbour.  So the update records, for every cell
// it reaches, the first of those four in the walk's order -- two bits per cell, two bit operations per plane and word -- and
// the walk reads its next step instead of probing three neighbours' hop counts.  Same path, cell for cell (the planner, route
// and trace tests are bit-identical); LDS per query 4 words of 8 bytes per costmap word: 6.4 KB for a 100 x 100 costmap (was 40).
// Thread 0 then walks the path and cuts the waypoints on the fly (nothing is stored per path cell).
constexpr size_t kPlanLdsMax = 160 * 1024 - 256;       // LDS per CU minus the static variables
constexpr int kPlanMaxWpt = 8;                         // costmap words a thread owns at most
inline size_t plan_words(int Hc, int Wc) { return (size_t)Hc * ((Wc + 63) / 64); }
// two frontier buffers and the two direction planes (the free-cell bitmap is assembled in the planes' area first)
inline size_t plan_lds(int Hc, int Wc) { return 4 * plan_words(Hc, Wc) * sizeof(unsigned long long); }
// threads per query: one costmap word per thread up to 1024 words (a 200 x 200 costmap -- 1000 x 1000 cells, the reference's
// corridor maps -- is 800 words: at 256 threads a thread owned four and a level cost four words' arithmetic; round 5)
inline int plan_block(int Hc, int Wc) { return plan_words(Hc, Wc) > 256 ? 1024 : 256; }
inline bool plan_fits(int Hc, int Wc) {
    return (size_t)Hc * Wc <= 65535 && plan_words(Hc, Wc) <= (size_t)kPlanMaxWpt * 256 && plan_lds(Hc, Wc) <= kPlanLdsMax;
}

// the levels: thread t owns the words t, t + BLOCK, ... (WPT of them).  Returns nothing; `reached` says whether the start was.
template <int BLOCK, int WPT>
__device__ __forceinline__ void plan_levels(unsigned long long* __restrict__ fa, unsigned long long* __restrict__ fb,
                                            unsigned long long* __restrict__ planes, int Hc, int Wc, int Ww, int n_words,
                                            int s_word, unsigned long long s_bit, int* any_s, int* reached) {
    typedef unsigned long long u64;
    const int tid = threadIdx.x;
    u64 avail[WPT], dA[WPT], dB[WPT];                    // unreached free cells, direction planes (bit 0, bit 1)
    // The neighbour words of a word on the map's edge do not exist: their index is clamped to the word itself and the value
    // masked, so that a level's reads are straight-line code -- five ds_read issued back to back, ONE wait.  (Written as
    // `has_left ? cur[x - 1] : 0` every read sat in its own exec-masked branch with its own wait: four LDS round trips per level.)
    int xo[WPT], xl[WPT], xr[WPT], xu[WPT], xd[WPT];
    u64 mo[WPT], ml[WPT], mr[WPT], mu[WPT], md[WPT];
    bool own[WPT];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const int x = tid + k * BLOCK;
        own[k] = x < n_words;
        xo[k] = own[k] ? x : 0;
        const int j = xo[k] / Ww, w = xo[k] - j * Ww;
        const bool hl = own[k] && w > 0, hr = own[k] && w + 1 < Ww, hu = own[k] && j > 0, hd = own[k] && j + 1 < Hc;
        xl[k] = hl ? xo[k] - 1 : xo[k]; xr[k] = hr ? xo[k] + 1 : xo[k]; xu[k] = hu ? xo[k] - Ww : xo[k]; xd[k] = hd ? xo[k] + Ww : xo[k];
        mo[k] = own[k] ? ~0ull : 0ull; ml[k] = hl ? 1ull : 0ull; mr[k] = hr ? 1ull << 63 : 0ull;
        mu[k] = hu ? ~0ull : 0ull; md[k] = hd ? ~0ull : 0ull;
        // free cells (assembled in the planes' area by the caller) minus the goal's bit, which is level 0
        avail[k] = planes[xo[k]] & ~fa[xo[k]] & mo[k];
        dA[k] = 0ull; dB[k] = 0ull;
    }
    __syncthreads();                                     // everybody has read its free words: the area is the planes' from here on
    u64* cur = fa, *nxt = fb;
    bool stale[WPT];                                     // my word of `nxt` may hold a frontier of two levels ago (fb: anything)
#pragma unroll
    for (int k = 0; k < WPT; ++k) stale[k] = true;
    for (int level = 1; level < 65536; ++level) {
        // the two flags and the frontier words in ONE LDS round trip (the reads are issued together, the exit test follows)
        const int stop = *reached, more = any_s[level % 3];
        u64 f[WPT], l[WPT], r[WPT], u[WPT], d[WPT];
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            f[k] = cur[xo[k]]; l[k] = cur[xl[k]]; r[k] = cur[xr[k]]; u[k] = cur[xu[k]]; d[k] = cur[xd[k]];
        }
        if (stop || !more) break;
        if (tid == 0) any_s[(level + 2) % 3] = 0;      // the flag of level + 1 (last read two barriers ago)
        bool found = false;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            // a wavefront none of whose 64 words has a frontier bit in itself or a neighbour word skips the update (cand = 0
            // changes nothing); it must still overwrite an OLD frontier in its word of `nxt` (`cur` of two levels ago): stale
            const bool work = (f[k] | l[k] | r[k] | u[k] | d[k]) != 0ull || stale[k];
            stale[k] = f[k] != 0ull;                            // `cur` is the next level's `nxt`
            if (__builtin_amdgcn_ballot_w64(work) == 0ull) continue;
            const u64 fk = f[k] & mo[k], dk = d[k] & md[k];
            const u64 R = (fk >> 1) | ((r[k] << 63) & mr[k]);   // cells whose +i neighbour is in the frontier
            const u64 L = (fk << 1) | ((l[k] >> 63) & ml[k]);   // ... -i neighbour
            const u64 cand = (R | L | (u[k] & mu[k]) | dk) & avail[k];
            if (own[k]) nxt[xo[k]] = cand;
            avail[k] &= ~cand;
            // the walk's order +i, -i, +j, -j as two bits per cell: 0, 1, 2, 3
            const u64 t = cand & ~R;
            dA[k] |= t & (L | ~dk);
            dB[k] |= t & ~L;
            found |= cand != 0ull;
            if (xo[k] == s_word && (cand & s_bit)) *reached = 1;
        }
        if (found) any_s[(level + 1) % 3] = 1;
        __syncthreads();
        u64* t = cur; cur = nxt; nxt = t;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const int x = tid + k * BLOCK;
        if (own[k]) { planes[2 * x] = dA[k]; planes[2 * x + 1] = dB[k]; }
    }
}

// BLOCK: threads of the workgroup that runs the query (round 4: 64 / 128 / 256 threads per query 130 / 112 / 112 us of
// navsim_replan per step on the c3 world, profiles/r04_replan/ab_block.txt)
// MAXWPT: the most costmap words a thread may own in this instantiation (the fused step's re-plan is compiled for one word per
// thread only: the level loop's registers are the kernel's)
template <int BLOCK = 256, int MAXWPT = kPlanMaxWpt>
__device__ __forceinline__ void plan_query(const uint8_t* __restrict__ c, int Hc, int Wc, double res_c, double ox,
                                           double oy, double sx_, double sy_, double gx_, double gy_, double interval,
                                           int max_wp, double* __restrict__ w, int32_t* __restrict__ n_wp,
                                           int32_t* __restrict__ path_cells, double* __restrict__ path_len,
                                           unsigned long long* __restrict__ cut_counter = nullptr) {
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) u64 plan_dyn[];
    __shared__ int any_s[3], reached;
    const int tid = threadIdx.x;
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
    const int Ww = (Wc + 63) >> 6, n_words = Hc * Ww;
    u64* fa = plan_dyn, *fb = plan_dyn + n_words, *planes = plan_dyn + 2 * n_words;
    {   // the free-cell bitmap (into the planes' area) and the goal as level 0: a wavefront reads 64 cells of a row with one
        // coalesced byte load and ballots them into the word, sixteen words per wavefront in flight (round 4: one thread
        // assembled a word from 64 dependent byte loads -- 20 us of every query, profiles/r04_replan/levels_*.txt's offset)
        constexpr int U = 16, kWaves = BLOCK / 64;
        const int wave = tid >> 6, lane = tid & 63;
        const int g_word = gj * Ww + (gi >> 6);
        for (int x0 = wave * U; x0 < n_words; x0 += kWaves * U) {
            bool fr[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int x = x0 + q, j = x / Ww, ww = x - j * Ww, i = (ww << 6) + lane;
                fr[q] = (x < n_words && i < Wc) ? (c[(size_t)j * Wc + i] == 0) : false;
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const u64 fw = __ballot(fr[q]);
                const int x = x0 + q;
                if (lane == 0 && x < n_words) { planes[x] = fw; fa[x] = (x == g_word) ? 1ull << (gi & 63) : 0ull; fb[x] = 0ull; }
            }
        }
        if (tid == 0) { any_s[0] = 0; any_s[1] = 1; any_s[2] = 0; reached = (si == gi && sj == gj); }
    }
    __syncthreads();
    {
        const int s_word = sj * Ww + (si >> 6);
        const u64 s_bit = 1ull << (si & 63);
        if constexpr (MAXWPT == 1) {
            plan_levels<BLOCK, 1>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);      // (the caller checked n_words <= BLOCK)
        } else if constexpr (MAXWPT == 2) {                  // (the caller checked n_words <= 2 BLOCK: the 512-thread searches of large launches)
            if (n_words <= BLOCK) plan_levels<BLOCK, 1>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
            else                  plan_levels<BLOCK, 2>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
        } else {
            // (plan_fits: at most kPlanMaxWpt * 256 words -- a 1024-thread workgroup never owns more than two per thread; with
            //  the four- and eight-word forms compiled in, its 128-register budget spilled 36-96 registers per lane)
            if (n_words <= BLOCK)          plan_levels<BLOCK, 1>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
            else if (n_words <= 2 * BLOCK || BLOCK * 2 >= kPlanMaxWpt * 256)
                                           plan_levels<BLOCK, 2>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
            else if (n_words <= 4 * BLOCK || BLOCK * 4 >= kPlanMaxWpt * 256)
                                           plan_levels<BLOCK, 4>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
            else                           plan_levels<BLOCK, kPlanMaxWpt>(fa, fb, planes, Hc, Wc, Ww, n_words, s_word, s_bit, any_s, &reached);
        }
    }
    __syncthreads();                                     // the planes (and the last level's `reached`)
#ifdef NAVSIM_DIAG_NO_WALK
    return;
#endif
    if (tid != 0 || !reached) return;
    int n = 0, count = 0, ci = si, cj = sj;
    const double fx0 = ((double)si + 0.5) * res_c + ox, fy0 = ((double)sj + 0.5) * res_c + oy;
    double fx = fx0, fy = fy0;                           // env.py:1261-1277, cut while walking
    // path_distance (env.py:757-759: |start - wp0| + sum |wp_k+1 - wp_k|) over EVERY waypoint of the path, accumulated
    // as they are cut -- also over those beyond max_wp, which are counted but not stored
    double L = 0.0, lx = sx_, ly = sy_;
    auto emit = [&](double cx, double cy) {
        if (count < max_wp) { w[2 * count] = cx; w[2 * count + 1] = cy; }
        ++count;
        const double ax = cx - lx, ay = cy - ly;
        L += sqrt(ax * ax + ay * ay);
        lx = cx; ly = cy;
    };
    const double i2_hi = interval * interval * (1.0 + 1.0e-12), i2_lo = interval * interval * (1.0 - 1.0e-12);
    // The anchor (fx, fy) is always a cell centre, so the distance to it is res_c * sqrt(di^2 + dj^2) up to rounding: the
    // integer sum decides `far` except within 1e-6 of the threshold, where the float64 expression of env.py:1261-1277 does
    // (a walked cell then costs integer work only: 0.32 -> 0.2 us, profiles/r04_replan/)
    const double q = interval / res_c, q2_hi = q * q * (1.0 + 1.0e-6), q2_lo = q * q * (1.0 - 1.0e-6);
    int ai = si, aj = sj;                                // the anchor's cell
    // the direction planes of the current cell's word stay in registers while the walk stays inside the word
    int wx = -1;
    u64 pa = 0ull, pb = 0ull;
    for (;;) {
        ++n;
        const int di = ci - ai, dj = cj - aj;
        const double n2 = (double)(di * di + dj * dj);
        const bool at_goal = ci == gi && cj == gj;
        bool far = n2 > q2_hi;
        if (far | at_goal | !(n2 < q2_lo)) {
            const double cx = ((double)ci + 0.5) * res_c + ox, cy = ((double)cj + 0.5) * res_c + oy;
            if (!far && !(n2 < q2_lo)) {
                // sqrt(d2) > interval, decided on d2 unless it sits within 1e-12 of interval^2 (sqrt is monotone
                // and correctly rounded, so the two tests agree outside that band)
                const double dx = fx - cx, dy = fy - cy;
                const double d2 = dx * dx + dy * dy;
                far = (d2 > i2_hi) || (!(d2 < i2_lo) && sqrt(d2) > interval);
            }
            if (far) { emit(cx, cy); fx = cx; fy = cy; ai = ci; aj = cj; }
            if (at_goal) { emit(cx, cy); break; }        // the goal cell closes the list
        }
        const int x = cj * Ww + (ci >> 6);               // the cell's step: +i, -i, +j, -j as the levels recorded it
        if (x != wx) { pa = planes[2 * x]; pb = planes[2 * x + 1]; wx = x; }
        const int bit = ci & 63;
        const int dir = (int)((pa >> bit) & 1ull) | ((int)((pb >> bit) & 1ull) << 1);
        if (dir == 0) ++ci; else if (dir == 1) --ci; else if (dir == 2) ++cj; else --cj;
    }
    *n_wp = count < max_wp ? count : max_wp;
    if (path_cells) *path_cells = n;
    if (path_len) *path_len = L;
    if (count > max_wp && cut_counter) atomicAdd(cut_counter, 1ull);       // a route stored cut (include/navsim.h)
}


__device__ __forceinline__ void rgp_cell(const navsim_config& c, const uint8_t* __restrict__ cost, int Wc, int live_w,
                                         int live_h, double res_c, uint64_t key, uint64_t& n, int kind, double rx,
                                         double ry, double dmin, double dmax, double& x, double& y) {
    for (int t = 0; t < 16; ++t) {
        int I = (int)(rg_u(key, n++) * live_w), J = (int)(rg_u(key, n++) * live_h);
        x = ((double)I + 0.5) * res_c + c.origin_x;
        y = ((double)J + 0.5) * res_c + c.origin_y;
        if (cost[(size_t)J * Wc + I]) continue;
        if (kind == 1 && !rg_start_ok(x, y, rx, ry, dmin)) continue;
        if (kind == 2 && !rg_goal_ok(rx, ry, x, y, dmin, dmax)) continue;
        return;
    }
}


// navsim_replan for ONE pedestrian (env.py:667-680; oracle navsim_replan_cpu), by the whole calling workgroup of BLOCK threads:
// pedestrian i of arena e stands on its final waypoint -- a new goal (a free costmap cell more than cfg.ped_min_goal_dist away,
// up to 4 rounds of 16 draws) and the waypoints of the shortest path to it; it keeps its old waypoint when no round finds a
// path.  Called by replan_kernel (one workgroup per waiting pedestrian) and by the front workgroups of
// navsim_step_replan_kernel (the arena's own workgroup, before it steps the arena).  Dynamic LDS: plan_lds(Hc, Wc).
template <int BLOCK, int MAXWPT = kPlanMaxWpt>
__device__ __forceinline__ void replan_one(const navsim_config& c, const navsim_state& st, const int e, const int i) {
    __shared__ double goal_s[2];
    __shared__ int32_t nwp_s;
    const int N = c.max_peds, P = c.max_waypoints, tid = threadIdx.x;
    const int q = e * N + i;
    const int Hc = c.map_h / 5, Wc = c.map_w / 5;
    const double res_c = c.resolution * 5.0;
    const uint8_t* cost = st.costmap + (size_t)map_slot_of(c, st, e) * Hc * Wc;
    const uint64_t genv = (uint64_t)(c.env_index_base + e);
    const uint64_t when = (uint64_t)st.steps[e] + ((uint64_t)st.episode[e] << 40);
    const double px = st.ped_pose[(size_t)q * 3], py = st.ped_pose[(size_t)q * 3 + 1];
    double* w = st.ped_waypoints + ((size_t)q * P) * 2;
    unsigned long long* cut_counter = st.counters ? st.counters + NAVSIM_COUNTER_ROUTES_CUT : nullptr;
    // The end of a route that was stored CUT (its last stored waypoint is not its goal): the pedestrian walks on to the
    // goal it had (round -1: no draw); only if no path joins them does it draw a new goal like the others.  Every thread
    // reads the same two words before any of them is rewritten (block-uniform).
    bool cut = false;
    {
        const int nw = st.ped_n_waypoints[q];
        // a candidate taken from st.ped_due may have been served or given a new world since the step flagged it: the
        // arrival test of env.py:667 (replan_flag_kernel's) on the CURRENT state decides (block-uniform)
        const double ddx = px - w[2 * (nw - 1)], ddy = py - w[2 * (nw - 1) + 1];
        if (!(sqrt(ddx * ddx + ddy * ddy) < 0.5)) return;
        if (st.ped_goal)
            cut = w[2 * (nw - 1)] != st.ped_goal[(size_t)q * 2] || w[2 * (nw - 1) + 1] != st.ped_goal[(size_t)q * 2 + 1];
    }
    __syncthreads();
    for (int round = cut ? -1 : 0; round < 4; ++round) {
        if (tid == 0) {
            double gx, gy;
            if (round < 0) {
                gx = st.ped_goal[(size_t)q * 2]; gy = st.ped_goal[(size_t)q * 2 + 1];
            } else {
                uint64_t key = rg_key(c.seed, genv, when, 0x52504E00ULL + (uint64_t)round * 256 + (uint64_t)i), m = 0;
                rgp_cell(c, cost, Wc, Wc, Hc, res_c, key, m, 2, px, py, c.ped_min_goal_dist, 1.0e300, gx, gy);
            }
            goal_s[0] = gx; goal_s[1] = gy;
        }
        __syncthreads();
        plan_query<BLOCK, MAXWPT>(cost, Hc, Wc, res_c, c.origin_x, c.origin_y, px, py, goal_s[0], goal_s[1], 2.0, P, w, &nwp_s,
                   nullptr, nullptr, cut_counter);
        __syncthreads();
        if (nwp_s > 0) {
            if (tid == 0) {
                st.ped_n_waypoints[q] = nwp_s;
                st.ped_wp_head[q] = 0;
                if (st.ped_goal) { st.ped_goal[(size_t)q * 2] = goal_s[0]; st.ped_goal[(size_t)q * 2 + 1] = goal_s[1]; }
                if (round < 0 && st.counters) atomicAdd(&st.counters[NAVSIM_COUNTER_ROUTES_RESUMED], 1ull);
            }
            break;
        }
        __syncthreads();                                 // nwp_s is rewritten by the next round
    }
}
