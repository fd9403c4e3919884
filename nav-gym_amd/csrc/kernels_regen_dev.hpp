// kernels_regen_dev.hpp -- device functions of the reset path that BOTH the reset kernels (kernels_reset.hpp) and the fused step
// (kernels_step.hpp: an arena that regenerates its own world, round 6) call: hash-keyed draws, the outdoor map with its exact
// field and rect records, spawn sampling, reset()'s first-scan test, the start / goal table + robot + pedestrians of one arena.
// Included inside the translation units' anonymous namespace behind kernels_plan.hpp (rg_u, rg_key, rg_start_ok ...); not a
// standalone header.  Specification: oracle/navsim_ref.c navsim_regen_cpu.
// a draw: the supplied one (tests only: navsim_state.regen_draws, NAVSIM_DRAW_* layout) or the hash-keyed one
__device__ __forceinline__ double rg_t(const double* __restrict__ tape, int slot, uint64_t key, uint64_t i) {
    return tape ? tape[slot] : rg_u(key, i);
}
__device__ __forceinline__ const double* rg_tape(const navsim_state& st, int e) {
    return st.regen_draws ? st.regen_draws + (size_t)e * NAVSIM_DRAWS_PER_ARENA : nullptr;
}
// side of the map an outdoor episode draws (cfg.outdoor_map_size; the reference: 400 inside its 1000-cell arenas)
__host__ __device__ inline int outdoor_size(const navsim_config& c) {
    return (c.outdoor_map_size > 0 && c.outdoor_map_size < c.map_w) ? c.outdoor_map_size : c.map_w;
}
// side of the live map of regenerated slot b: kind[b] = G > 0 for a corridor map (the whole arena), 0 for an outdoor one
__device__ __forceinline__ int live_size(const navsim_config& c, const int* __restrict__ kind, int b) {
    return kind[b] ? c.map_w : outdoor_size(c);
}

// per-episode env_param draws that are plain state (env.py:281-292, 786, 439): one thread per regenerated arena
__device__ __forceinline__ void regen_params(const navsim_config& c, const navsim_state& st, int e) {
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    const double* tape = rg_tape(st, e);
    const uint64_t pk = rg_key(c.seed, genv, ep, 0x50524DULL);
    if (c.num_humans_hi > 0 && c.ped_model != NAVSIM_PED_NONE && st.n_peds) {
        int n = c.num_humans_lo + (int)(rg_t(tape, NAVSIM_DRAW_NUM_HUMANS, pk, 1) * (double)(c.num_humans_hi - c.num_humans_lo + 1));
        st.n_peds[e] = n > c.max_peds ? c.max_peds : n;
    }
    if (c.scan_noise_std_hi >= 0.0 && st.scan_noise_std)
        st.scan_noise_std[e] = (float)(c.scan_noise_std_lo + (c.scan_noise_std_hi - c.scan_noise_std_lo) *
                                                                 rg_t(tape, NAVSIM_DRAW_SCAN_NOISE_STD, pk, 2));
}

// create_outdoor_map (map_generator.py:126-143), hash-keyed: kRegenSlices workgroups per map, each owning a band of
// stored rows.  For an OUTDOOR map the kernel writes the exact distance field straight from the generator's geometry:
// the obstacles are the border frame and n boxes, and the squared distance of a cell to a rectangle of occupied
// cells is max(x0 - x, x - x1, 0)^2 + max(y0 - y, y - y1, 0)^2, so d2 = min over them -- the same integers the
// distance transform finds, with no dependent scan (the two transform kernels then skip the slot).
//
// Round 3: a thread owns EIGHT consecutive cells of a row -- one 16-byte tile row of the packed field (one store),
// one 8-byte store of the occupancy -- without a division per cell; 128 slices per map walk its (tile row, 32 tiles)
// units, the eight rows of a tile on adjacent lanes.  41 -> 22 us for the 5 maps of a c5 step.  (Measured and dropped: one 8x8 tile per wavefront with a cell per lane,
// 46 us; 32 or 512 bands, 23 / 33 us.  The kernel's time does not follow its arithmetic: profiles/README.md.)
// `direct`: every map of this call is an outdoor one and nothing downstream wants the per-slot scratch: the field goes
// straight into the arena's own buffers, no copy kernel; a world that keeps rect records gets the records of the new map
// from the same pass (rect_all; 12.6 -> 13.8 us, against 66-87 us of the verified builder for the same maps).
// The GRID is bounded (kRegenGrid workgroups walk the (slot, band) items of the arenas that really finished): a launch
// of regen_cap x 128 workgroups of which 27 in 32 slots find nothing to do cost 4 ns per such workgroup -- 9.9 / 14.2 /
// 22.6 us for regen_cap 8 / 16 / 32 with the same 5 maps to draw (profiles/_diag/regen_twice.sh).
constexpr int kRegenSlices = 128;
constexpr int kRegenGrid = 1024;
inline int regen_grid(int slots) { const long n = (long)slots * kRegenSlices; return (int)(n < kRegenGrid ? n : kRegenGrid); }
// ---- rect records of an OUTDOOR map from the generator's geometry (round 3).  The map IS a union of rectangles of occupied
// cells -- four border walls (stretched over everything outside the live map) and the clipped boxes -- and its exact d2 is
// the minimum of the rectangle distances (regen_maps_item writes the field that way).  So the record of an 8x8 tile
// (kernels_rect.hpp) needs no search and no verification pass: the tile's record is valid iff the nearest rectangle of
// every in-map cell of the tile (ties: the lowest index) is one of at most two rectangles, and those two ARE the record
// (regen_maps_item: the same pass that writes the field).
struct RectSet2 { int a, b, bad; };
__device__ __forceinline__ void rect_set_insert(RectSet2& s, int x) {
    if (x < 0 || x == s.a || x == s.b) return;
    if (s.a < 0) s.a = x;
    else if (s.b < 0) s.b = x;
    else s.bad = 1;
}
// rectangle `idx` of the map in field coordinates (x = column, y = stored row): 0..3 the walls, 4 + o box o
__device__ __forceinline__ void regen_rect_of(int idx, int live, int size, int hw, const int* ocx, const int* ocy,
                                              unsigned& lo, unsigned& hi) {
    int x0, x1, y0, y1;
    if (idx == 0)      { x0 = 0; x1 = 4; y0 = 0; y1 = size - 1; }                  // q <= 4
    else if (idx == 1) { x0 = live - 5; x1 = size - 1; y0 = 0; y1 = size - 1; }    // q >= live - 5 (and the padding beside)
    else if (idx == 2) { x0 = 0; x1 = size - 1; y0 = live - 5; y1 = size - 1; }    // generator rows r <= 4: stored rows y >= live - 5 (and the padding above)
    else if (idx == 3) { x0 = 0; x1 = size - 1; y0 = 0; y1 = 4; }                  // r >= live - 5: y <= 4
    else {
        const int o = idx - 4;
        int bx0 = ocx[o] - hw, bx1 = ocx[o] + hw, by0 = ocy[o] - hw, by1 = ocy[o] + hw;
        bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
        bx1 = bx1 > live - 1 ? live - 1 : bx1; by1 = by1 > live - 1 ? live - 1 : by1;
        x0 = by0; x1 = by1; y0 = live - 1 - bx1; y1 = live - 1 - bx0;
    }
    lo = ((unsigned)y0 << 16) | (unsigned)x0;
    hi = ((unsigned)y1 << 16) | (unsigned)x1;
}
__device__ __forceinline__ void regen_maps_item(const navsim_config& c, const navsim_state& st, int b, int slice, int e,
                                                uint8_t* __restrict__ occ_all,
                                                const uint8_t* __restrict__ grid_all, const int* __restrict__ kind,
                                                char* __restrict__ field_scratch, size_t field_bytes,
                                                float* __restrict__ ovf_scratch, int direct, bool all_outdoor,
                                                uint4* __restrict__ rect_all, char* __restrict__ index_all, int* ocx, int* ocy,
                                                const int tid_arg = -1, const bool boxes_ready = false) {
    // tid_arg >= 0 (the fused step's lone regen: a workgroup of several 256-thread groups, each taking a slice): this thread's
    // index inside its group; boxes_ready: ocx / ocy already hold the map's boxes (no barrier in this call then)
    const int size = c.map_w, tid = tid_arg >= 0 ? tid_arg : (int)threadIdx.x;
    uint8_t* occ = occ_all ? occ_all + (size_t)b * size * size : nullptr;
    const int rows = (size + kRegenSlices - 1) / kRegenSlices;
    const int y0 = slice * rows, y1 = (y0 + rows < size) ? y0 + rows : size;
    if (const int G = all_outdoor ? 0 : kind[b]) {                     // corridor map: nearest upscaling + flip
        const uint8_t* gsrc = grid_all + (size_t)b * 10000;
        for (int idx = y0 * size + tid; idx < y1 * size; idx += 256) {
            int yy = idx / size, xx = idx - yy * size;
            occ[(size_t)(size - 1 - yy) * size + xx] = gsrc[(int)(((long long)yy * G) / size) * G + (int)(((long long)xx * G) / size)];
        }
        return;
    }
    // ---- outdoor map: `live` x `live` cells in the corner [0, live)^2 of the arena's size x size array (stored row
    // y = live - 1 - r of generator row r: np.flipud over the live rows); everything outside is occupied -- behind the
    // 5-cell border wall no ray and no distance sees it
    const int live = outdoor_size(c);
    const double* tape = rg_tape(st, e);
    const uint64_t key = rg_key(c.seed, (uint64_t)(c.env_index_base + e), (uint64_t)st.episode[e], 0x4D4150ULL);
    double w = c.obstacle_width_lo + (c.obstacle_width_hi - c.obstacle_width_lo) * rg_t(tape, NAVSIM_DRAW_OBSTACLE_WIDTH, key, 0);
    const int hw = (int)(10.0 * w);
    int span = live - 2 * hw - 3;
    span = span < 1 ? 1 : span;
    const int obs_hi = c.obstacle_number_hi > c.obstacle_number ? c.obstacle_number_hi : c.obstacle_number;
    int n_obs = c.obstacle_number + (int)(rg_t(tape, NAVSIM_DRAW_OBSTACLE_NUMBER,
                                               rg_key(c.seed, (uint64_t)(c.env_index_base + e), (uint64_t)st.episode[e], 0x50524DULL), 0) *
                                          (double)(obs_hi - c.obstacle_number + 1));
    n_obs = n_obs < 64 ? n_obs : 64;
    if (!boxes_ready) {
        if ((int)threadIdx.x < n_obs) {
            const int o = (int)threadIdx.x;
            ocx[o] = hw + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 2 * o, key, 1 + 2 * (uint64_t)o) * span);
            ocy[o] = hw + 2 + (int)(rg_t(tape, NAVSIM_DRAW_MAP + 2 * o + 1, key, 2 + 2 * (uint64_t)o) * span);
        }
        __syncthreads();
    }
    // where the field goes: the arena's own buffers, or this slot's scratch (installed by regen_field_kernel)
    const bool f32 = c.field_format == NAVSIM_FIELD_F32;
    const int ms = map_slot_of(c, st, e);                    // where the arena's map lives (navsim_state.map_slot)
    char* fs = direct ? (char*)st.field + (size_t)ms * field_bytes : field_scratch + (size_t)b * field_bytes;
    float* ov = direct ? (st.field_overflow ? (float*)st.field_overflow + (size_t)ms * size * size : nullptr)
                       : (ovf_scratch ? ovf_scratch + (size_t)b * size * size : nullptr);
    // Work units: (tile row, block of 32 tiles) -- 256 threads = 32 tiles x 8 rows, thread = (tile of the block, row of the
    // tile); unit u of the map belongs to slice u % kRegenSlices.  A thread owns EIGHT consecutive cells of a row = one
    // 16-byte tile row of the packed field; the eight threads of a tile are adjacent lanes: their stores fill the tile's
    // 128 bytes, and the tile's rect record (kernels_rect.hpp) falls out of the same pass.
    const int tpr = (size + 7) >> 3, blocks = (tpr + 31) >> 5;
    const int ry = tid & 7, tsub = tid >> 3;
    if (index_all && rect_all && slice == 0) {               // the list of the index form: walls 0..3, box o at 4 + o, the rest defined
        uint2* lst = (uint2*)(index_all + (size_t)map_slot_of(c, st, e) * rect_index_row_bytes(size, size));
        if (tid < kRectListLen) {
            uint2 v = make_uint2(0u, 0u);
            if (tid < 4 + n_obs) regen_rect_of(tid, live, size, hw, ocx, ocy, v.x, v.y);
            lst[tid] = v;
        }
    }
    for (int u = slice; u < tpr * blocks; u += kRegenSlices) {
        const int ty = u / blocks, tx = (u - ty * blocks) * 32 + tsub;
        const int y = ty * 8 + ry, x0 = tx << 3;
        const bool in_map = tx < tpr && y < size;
        RectSet2 set = {-1, -1, 0};
        if (in_map) {
            const int r = live - 1 - y;                                      // generator row (negative: outside the live map)
            int d2[8], arg[8];
            // border frame (and everything outside the live map): m <= 0.  arg: the rectangle the distance comes from
            // (regen_rect_of: 0..3 the walls, 4 + o box o; the lowest index wins a tie)
            int mr = r - 4, ar = 2;
            if (live - 5 - r < mr) { mr = live - 5 - r; ar = 3; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = x0 + j;
                int m = mr, am = ar;
                if (q - 4 < m || (q - 4 == m && 0 < am)) { m = q - 4; am = 0; }
                if (live - 5 - q < m || (live - 5 - q == m && 1 < am)) { m = live - 5 - q; am = 1; }
                d2[j] = (m > 0) ? m * m : 0;
                arg[j] = am;
            }
            for (int o = 0; o < n_obs; ++o) {                                // boxes are drawn clipped to the map
                int bx0 = ocx[o] - hw, bx1 = ocx[o] + hw, by0 = ocy[o] - hw, by1 = ocy[o] + hw;
                bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0;
                bx1 = bx1 > live - 1 ? live - 1 : bx1; by1 = by1 > live - 1 ? live - 1 : by1;
                int dr = bx0 - r > r - bx1 ? bx0 - r : r - bx1;
                dr = dr < 0 ? 0 : dr;
                const int dr2 = dr * dr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = x0 + j;
                    int dq = by0 - q > q - by1 ? by0 - q : q - by1;
                    dq = dq < 0 ? 0 : dq;
                    const int v = dr2 + dq * dq;
                    if (v < d2[j]) { d2[j] = v; arg[j] = 4 + o; }
                }
            }
            const bool whole = x0 + 8 <= size;                               // the last group of a ragged row is partial
            if (occ) {
                if (whole && ((size & 7) == 0)) {
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { lo |= (uint32_t)(d2[j] == 0) << (8 * j); hi |= (uint32_t)(d2[4 + j] == 0) << (8 * j); }
                    *(uint2*)(occ + (size_t)y * size + x0) = make_uint2(lo, hi);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (x0 + j < size) occ[(size_t)y * size + x0 + j] = (uint8_t)(d2[j] == 0);
                }
            }
            if (f32) {
                float* dst = (float*)fs + (size_t)y * size + x0;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (x0 + j < size) dst[j] = sqrtf((float)d2[j]);
            } else {
                // one tile row of the packed field = 8 cells = 16 bytes (padding cells of an edge tile included: they are
                // never read, and d2 of a cell outside the map is 0 here)
                uint32_t pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = (uint32_t)(d2[2 * j] >= 65535 ? 0xFFFF : d2[2 * j]);
                    const uint32_t bq = (uint32_t)(d2[2 * j + 1] >= 65535 ? 0xFFFF : d2[2 * j + 1]);
                    pk[j] = a | (bq << 16);
                }
                *(uint4*)((uint16_t*)fs + FieldU16T::index(x0, y, tpr)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                if (ov) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (x0 + j < size) ov[(size_t)y * size + x0 + j] = sqrtf((float)d2[j]);
                }
            }
            if (rect_all) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x0 + j < size) rect_set_insert(set, arg[j]);
            }
        }
        // The record of the tile, when the world keeps a rect table and this call writes fields directly (outdoor maps only).
        // The map IS a union of rectangles of occupied cells -- four border walls (stretched over everything outside the
        // live map) and the clipped boxes -- and its exact d2 is the minimum of the rectangle distances, so the record needs
        // no search and no verification pass: it is valid iff the nearest rectangle of every in-map cell of the tile is one
        // of at most two rectangles, and those two ARE the record.  Sets of at most two indices, merged over the tile's
        // eight rows by three shuffles.
        if (rect_all) {                                                      // block-uniform
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                const int oa = __shfl_xor(set.a, off), ob = __shfl_xor(set.b, off), obad = __shfl_xor(set.bad, off);
                rect_set_insert(set, oa);
                rect_set_insert(set, ob);
                set.bad |= obad;
            }
            if (ry == 0 && tx < tpr) {
                uint4 rec;
                if (set.bad || set.a < 0) {
                    rec = make_uint4(kRectInvalid, 0u, 0u, 0u);
                } else {
                    regen_rect_of(set.a, live, size, hw, ocx, ocy, rec.x, rec.y);
                    regen_rect_of(set.b < 0 ? set.a : set.b, live, size, hw, ocx, ocy, rec.z, rec.w);
                }
                (rect_all + (size_t)ms * rect_tiles_per_map(size, size))[(size_t)ty * tpr + tx] = rec;
                // the index form of the same record (kernels_rect.hpp): the generator's own rectangle numbers
                if (index_all) {
                    uint16_t* pair = (uint16_t*)(index_all + (size_t)ms * rect_index_row_bytes(size, size) + (size_t)kRectListLen * 8);
                    pair[(size_t)ty * tpr + tx] = (set.bad || set.a < 0) ? (uint16_t)kRectNoIndex
                                                                         : (uint16_t)((unsigned)set.a | ((unsigned)(set.b < 0 ? set.a : set.b) << 8));
                }
            }
        }
    }
}

// kind 0: any free cell; 1: a start, dropped when closer than dmin to (rx, ry); 2: a goal of the start (rx, ry).
// Cells are drawn in the live map [0, size)^2.
template <typename Field>
__device__ __forceinline__ void rg_sample(const navsim_config& c, const Field& f, int size, uint64_t key, uint64_t& n,
                                          double clr, int kind, double rx, double ry, double dmin, double dmax,
                                          double& x, double& y) {
    int bi = 0, bj = 0;
    float bd = -1.0f;
    for (int t = 0; t < 64; ++t) {
        int i = (int)(rg_u(key, n++) * size), j = (int)(rg_u(key, n++) * size);
        float d = f.at(i, j);
        double px = ((double)i + 0.5) * c.resolution + c.origin_x;
        double py = ((double)j + 0.5) * c.resolution + c.origin_y;
        bool ok = (double)d >= clr;
        if (ok && kind == 1) ok = rg_start_ok(px, py, rx, ry, dmin);
        if (ok && kind == 2) ok = rg_goal_ok(rx, ry, px, py, dmin, dmax);
        if (ok) { x = px; y = py; return; }
        if (d > bd) { bd = d; bi = i; bj = j; }
    }
    x = ((double)bi + 0.5) * c.resolution + c.origin_x;
    y = ((double)bj + 0.5) * c.resolution + c.origin_y;
}

// env.py:776-781: reset() re-draws the robot when its first scan -- taken before any pedestrian exists -- has a beam
// inside the discomfort zone.  All threads of the workgroup scan the static map from `pose` (the plain march of
// calc_range through Field::at, full beam directions, march limited like the step's) and agree on the answer.
// No scan noise here (oracle spawn_in_discomfort: build-defined like every random number).
template <typename Field>
__device__ __forceinline__ bool spawn_in_discomfort(const navsim_config& c, const navsim_state& st, const Field& f,
                                                    const double* pose) {
    const int B = c.n_beams, H = c.map_h, W = c.map_w;
    const float lx = (float)pose[0], ly = (float)pose[1], lth = (float)pose[2];      // env.py:386
    int i0, j0;
    nv::xy_to_ij_f32(lx, ly, c, i0, j0);                                             // env.py:419
    const float x0 = (float)i0, y0 = (float)j0;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const float res = (float)c.resolution, rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    int bad = 0;
    for (int k = (int)threadIdx.x; k < B; k += (int)blockDim.x) {
        // Only "is this beam shorter than its discomfort threshold?" is asked, so the march stops early: a hit found at
        // parameter t lies at least t - sqrt(2) cells from the origin, hence from t >= dthr / resolution + 4 on no hit
        // can be inside the zone.  Same answer as the full scan (oracle: robot_scan, then the comparison), a few probes
        // instead of the chain of the longest ray: 27 -> 4 us of regen_commit_kernel.
        const float dthr = st.scan_discomfort[k];
        float lim = dthr / res + 4.0f;
        lim = lim < max_range ? lim : max_range;
        float dx, dy;
        nv::beam_dir((float)(nv::linspace_k(c, k, step) + (double)lth), dx, dy);
        float t = 0.0f, r = max_range;
        const bool fma_pos = c.march_rule == NAVSIM_MARCH_F32_FMA;
        while (t < lim) {
            const int px = (int)(fma_pos ? __builtin_fmaf(dx, t, x0) : x0 + dx * t);
            const int py = (int)(fma_pos ? __builtin_fmaf(dy, t, y0) : y0 + dy * t);
            if (px >= W || px < 0 || py < 0 || py >= H) break;
            const float d = f.at(px, py);
            if (d <= 0.0f) {
                const float xd = (float)px - x0, yd = (float)py - y0;
                r = sqrtf(xd * xd + yd * yd);
                break;
            }
            const float stp = (c.march_rule != NAVSIM_MARCH_F64) ? d * 0.999f : (float)((double)d * 0.999);
            t += (stp > 1.0f) ? stp : 1.0f;
        }
        r = r * res;
        r = r < 0.0f ? 0.0f : r;
        r = r > rmax ? rmax : r;
        bad |= (r < dthr);
    }
    return __syncthreads_or(bad) != 0;
}

// rg_sample with its 64 tries on the 64 lanes of ONE wavefront (the whole wavefront calls; every lane returns the same
// x, y, n).  Try t is lane t: its two draws are rg_u(key, n + 2t) and rg_u(key, n + 2t + 1) either way; the first
// lane whose cell passes is the sequential loop's answer, and without one the FIRST lane holding the best clearance
// (the loop's `d > bd` keeps the earliest maximum).  n advances as the loop would have: 2 per try made.
// The sequential form reads up to 64 cells one after the other (a microsecond each); this one reads them at once.
template <typename Field>
__device__ __forceinline__ void rg_sample_wave(const navsim_config& c, const Field& f, int size, uint64_t key, uint64_t& n,
                                               double clr, int kind, double rx, double ry, double dmin, double dmax,
                                               double& x, double& y) {
    const int lane = (int)threadIdx.x & 63;
    const int i = (int)(rg_u(key, n + 2 * (uint64_t)lane) * size), j = (int)(rg_u(key, n + 2 * (uint64_t)lane + 1) * size);
    const float d = f.at(i, j);
    const double px = ((double)i + 0.5) * c.resolution + c.origin_x;
    const double py = ((double)j + 0.5) * c.resolution + c.origin_y;
    bool ok = (double)d >= clr;
    if (ok && kind == 1) ok = rg_start_ok(px, py, rx, ry, dmin);
    if (ok && kind == 2) ok = rg_goal_ok(rx, ry, px, py, dmin, dmax);
    const unsigned long long okm = __ballot(ok);
    int t;
    if (okm) {
        t = __ffsll(okm) - 1;
        n += 2 * (uint64_t)(t + 1);
    } else {
        float m = d;
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
        t = __ffsll((unsigned long long)__ballot(d == m)) - 1;
        n += 128;
    }
    x = __shfl(px, t, 64);
    y = __shfl(py, t, 64);
}


// the start / goal table, the robot (with reset()'s first-scan test) and the pedestrians of ONE regenerated arena, by the
// calling workgroup (any multiple of 64 threads): the body of regen_commit_kernel, shared with the fused step's lone regen.
// size: side of the live map; robot_xy: two doubles of LDS.
template <typename Field>
__device__ __forceinline__ void regen_commit_arena(const navsim_config& c, const navsim_state& st, const int e, const int size,
                                                   double* robot_xy) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kWaves = (int)blockDim.x >> 6;                  // (1024 threads in regen_commit_kernel; the arena's own workgroup in the fused step)
    const int N = c.max_peds, K = c.n_spawn, P = c.max_waypoints;
    const Field f(st.field, st.field_overflow, map_slot_of(c, st, e), c.map_h, c.map_w);
    const uint64_t genv = (uint64_t)(c.env_index_base + e), ep = (uint64_t)st.episode[e];
    double* sp = (double*)st.spawn_pose + (size_t)e * K * 3;
    double* sg = (double*)st.spawn_goal + (size_t)e * K * 2;
    const double clr = c.spawn_clearance / c.resolution;
    for (int k = wave; k < K; k += kWaves) {
        uint64_t key = rg_key(c.seed, genv, ep, 0x53504157ULL + (uint64_t)k), n = 0;
        double x, y, gx, gy;
        rg_sample_wave(c, f, size, key, n, clr, 0, 0, 0, 0, 0, x, y);
        double th = nv::kTwoPi * rg_u(key, n++);
        rg_sample_wave(c, f, size, key, n, clr, 2, x, y, c.min_goal_dist, c.max_goal_dist, gx, gy);
        if (lane == 0) {
            sp[3 * k] = x; sp[3 * k + 1] = y; sp[3 * k + 2] = th;
            sg[2 * k] = gx; sg[2 * k + 1] = gy;
        }
    }
    __threadfence_block();
    __syncthreads();
    int idx = (int)(rg_key(c.seed, genv, ep, 0x5eedULL) % (uint64_t)K);
    if (c.regen_check_discomfort)                // env.py:776-781: first table entry from idx on whose first scan is clear
        for (int s_ = 0; s_ < K; ++s_) {         // (block-uniform loop)
            const int j = (idx + s_) % K;
            if (!spawn_in_discomfort(c, st, f, sp + 3 * j)) { idx = j; break; }
        }
    if (tid == 0) {
        double* rp = st.robot_pose + 3 * (size_t)e;
        rp[0] = sp[3 * idx]; rp[1] = sp[3 * idx + 1]; rp[2] = sp[3 * idx + 2];
        st.robot_goal[2 * e] = sg[2 * idx]; st.robot_goal[2 * e + 1] = sg[2 * idx + 1];
        robot_xy[0] = rp[0]; robot_xy[1] = rp[1];
    }
    __syncthreads();
    int n = (c.ped_model == NAVSIM_PED_NONE) ? 0 : st.n_peds[e];
    n = n > N ? N : n;
    const double pclr = c.ped_clearance / c.resolution;
    for (int i = wave; i < n; i += kWaves) {
        size_t q = (size_t)e * N + i;
        uint64_t key = rg_key(c.seed, genv, ep, 0x504544ULL + (uint64_t)i), m = 0;
        double x, y, gx, gy;
        rg_sample_wave(c, f, size, key, m, pclr, 1, robot_xy[0], robot_xy[1], c.ped_min_robot_dist, 0, x, y);
        double th = nv::kTwoPi * rg_u(key, m++);
        rg_sample_wave(c, f, size, key, m, pclr, 2, x, y, c.ped_min_goal_dist, 1.0e300, gx, gy);
        if (lane == 0) {
            st.ped_pose[q * 3] = x; st.ped_pose[q * 3 + 1] = y; st.ped_pose[q * 3 + 2] = th;
            st.ped_vel[q * 2] = 0.0; st.ped_vel[q * 2 + 1] = 0.0;
            ((double*)st.ped_v_pref)[q] = c.v_pref_lo + (c.v_pref_hi - c.v_pref_lo) * rg_u(key, m++);
            ((uint8_t*)st.ped_has_legs)[q] = rg_u(key, m++) < c.has_legs_ratio;
            double* wp = st.ped_waypoints + (q * P) * 2;
            wp[0] = gx; wp[1] = gy;
            st.ped_n_waypoints[q] = 1;
            st.ped_wp_head[q] = 0;
            if (st.ped_goal) { st.ped_goal[q * 2] = gx; st.ped_goal[q * 2 + 1] = gy; }
        }
    }
    if (tid == 0 && st.ped_due) st.ped_due[e] = 0ull;      // new pedestrians: nobody waits for navsim_replan
}

// An arena regenerates ITS OWN world (round 6; NAVSIM_AUTORESET_NEXT_STEP with staged worlds, navsim_step_install): the workgroup
// of an arena that is reset in this call and found no world staged does, in place of a step, what navsim_regen's kernels do
// for it -- per-episode parameters, the outdoor map with its exact field / rect records / index row, the start / goal table,
// the robot with reset()'s first-scan test, the pedestrians -- with the SAME device functions, so the result is navsim_regen's
// bit for bit; its first observation is the caller's ordinary reset path.  Slow for the one workgroup (the map's 128 bands one
// after the other: ~0.1 ms) and rare (an arena that ends two episodes within a staging pass's latency: one in a thousand
// steps of c5) -- what it buys is that NO launch of the reset path is left behind the steps, with no rule.
// Worlds of outdoor maps without planning and without a costmap only (the host checks); BLOCK >= 256.
template <typename Field, int BLOCK>
__device__ __forceinline__ void regen_lone(const navsim_config& c, const navsim_state& st, const int e) {
    static_assert(BLOCK >= 256 && BLOCK % 256 == 0, "regen_lone: whole 256-thread groups");
    __shared__ int lone_ocx[64], lone_ocy[64];
    __shared__ double lone_robot_xy[2];
    const int tid = threadIdx.x;
    if (tid == 0) regen_params(c, st, e);
    constexpr int kGroups = BLOCK / 256;
    const size_t fbytes = c.field_format == NAVSIM_FIELD_F32 ? (size_t)c.map_h * c.map_w * sizeof(float)
                                                             : (size_t)((c.map_h + 7) / 8) * ((c.map_w + 7) / 8) * 64 * sizeof(uint16_t);
    for (int s0 = 0; s0 < kRegenSlices; s0 += kGroups)
        regen_maps_item(c, st, 0, s0 + (tid >> 8), e, nullptr, nullptr, nullptr, nullptr, fbytes, nullptr, 1, true,
                        (uint4*)st.rect_table, st.rect_table ? (char*)st.rect_index : nullptr, lone_ocx, lone_ocy, tid & 255, s0 > 0);
    // the new field, records and parameters are read back below (and by the reset path) through this CU's vector cache, which
    // may still hold lines of the OLD map: write back, then invalidate
    __threadfence();
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    regen_commit_arena<Field>(c, st, e, outdoor_size(c), lone_robot_xy);
    __threadfence();
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
