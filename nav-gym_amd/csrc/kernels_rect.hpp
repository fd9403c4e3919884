// kernels_rect.hpp -- analytic rectangle records of the distance field ("rect table"): builder + decode.
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).
//
// The exact squared distance of a free cell p is min over the obstacle cells q of |p - q|^2.  Obstacles of the
// reference's maps (map_generator.py:97-143: border walls, boxes, the complement of corridors) are unions of
// axis-aligned rectangles of cells, and for a rectangle R = [x0, x1] x [y0, y1] of OCCUPIED cells
//     min_{q in R} |p - q|^2 = max(x0 - px, px - x1, 0)^2 + max(y0 - py, py - y1, 0)^2        (integers).
// Inside one 8x8-cell tile the nearest obstacle of every cell almost always belongs to one of at most two such
// rectangles (measured on the bench's maps: 91 % of the lidar's probes fall in one-rectangle tiles, 99.8 % in
// tiles needing at most two).  The record of a tile therefore holds two rectangles -- 4 x int16 each, 16 bytes,
// against the 128 bytes of the tile's uint16 d2 values -- and the march evaluates
//     d2 = min(dist2(A, p), dist2(B, p))
// in registers (v_pk_sub_i16 x2, v_pk_max_i16 x2, v_dot2_i32_i16 per rectangle).  The builder VERIFIES every
// record against the exact field on all in-map cells of its tile and marks the tile invalid otherwise; a probe
// in an invalid tile reads the field as before.  A valid record reproduces the field's integer d2 exactly, so
// the probe sequence -- and every scan -- is unchanged bit for bit.
//
// Any rectangle of occupied cells gives an upper bound of the true d2 at every cell (its cells ARE obstacles),
// so a record can only be wrong by being too large somewhere, which the verification catches.

typedef short rect_s2 __attribute__((ext_vector_type(2)));
constexpr unsigned kRectInvalid = 0x7FFFu;           // x0 of rectangle A of a tile without a valid record
constexpr int kRectShift = 3;                        // log2 of the tile side (cells): the field's 8x8 tiles

__host__ __device__ inline size_t rect_tiles_per_map(int H, int W) {
    return (size_t)((H + 7) >> kRectShift) * ((W + 7) >> kRectShift);
}

// squared distance from cell p = (py << 16 | px) to the rectangle lo = (y0 << 16 | x0), hi = (y1 << 16 | x1)
__device__ __forceinline__ int rect_dist2(unsigned lo, unsigned hi, unsigned p) {
    const rect_s2 L = __builtin_bit_cast(rect_s2, lo), Hh = __builtin_bit_cast(rect_s2, hi), P = __builtin_bit_cast(rect_s2, p);
    // p minus p clamped into the rectangle (lo <= hi per component in every record the builder writes): per
    // component L - P, 0 or P - H up to the sign, which the squares drop.  v_pk_max_i16, v_pk_min_i16, v_pk_sub_i16.
    const rect_s2 m = P - __builtin_elementwise_min(__builtin_elementwise_max(P, L), Hh);
    // v_dot2_i32_i16 with the accumulator as an inline 0 (the builtin selects v_dot2c + a v_mov of the zero)
    int r;
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, m)));
    return r;
}
__device__ __forceinline__ unsigned rect_cell(int px, int py) { return ((unsigned)py << 16) | (unsigned)px; }
__device__ __forceinline__ int rect_record_d2(const uint4 rec, unsigned p) {
    const int a = rect_dist2(rec.x, rec.y, p), b = rect_dist2(rec.z, rec.w, p);
    return a < b ? a : b;
}
__device__ __forceinline__ int rect_record_d2(const uint4 rec, int px, int py) { return rect_record_d2(rec, rect_cell(px, py)); }
__device__ __forceinline__ bool rect_record_invalid(const uint4 rec) { return (rec.x & 0xFFFFu) == kRectInvalid; }

// ============================================================================================
// The INDEX form of an arena's record table (round 4): "map tiles staged through LDS" at eight arenas per CU.
//
// A map holds few distinct rectangles -- an outdoor map 4 walls + its boxes (14 on the bench's maps), a corridor map the
// maximal rectangles of its walls (a few dozen to a couple of hundred) -- and the 16-byte records of its 8x8 tiles name
// the same ones over and over.  So an arena's table is stored a second time as
//     list[256]   the distinct rectangles (8 bytes each: lo = y0 << 16 | x0, hi = y1 << 16 | x1)          2 048 B
//     pair[tile]  two list indices per tile (A | B << 8; 0xFFFF = no valid record: read the field)        2 B per tile
// 10 KB for a 500 x 500 map against the records' 63.5 KB, 34 KB for 1000 x 1000: the fused step copies the arena's row into
// LDS once (all threads, 16-byte loads, beside phase 0) and every probe of its scans is a ds_read_u16 and two ds_read_b64 --
// ~0.1 us instead of a global load's 0.5-2 us -- at the SAME residency as before (eight 256-thread workgroups per CU).
// Why it matters (profiles/r04_split/README.md): a probe round is a dependent chain (position -> tile -> record -> d2 ->
// sqrt -> step -> next position) and with eight wavefronts per SIMD its latency, not its instruction count, sets the pace;
// the record load was the chain's longest link.  HBM traffic of a c2 launch: 220 -> 60 MB.
// Derived from the verified records (rect_index_kernel below) or written next to them by navsim_regen's direct
// generator: lossless by construction -- a tile whose rectangles do not fit the list (more than 255 distinct ones in the
// map) simply has no index and is read from the field, like a tile without a valid record.
// ============================================================================================
constexpr int kRectListLen = 256;                    // list entries in a row (index 255 is the "no record" marker, never a rectangle)
constexpr unsigned kRectNoIndex = 0xFFFFu;
__host__ __device__ inline size_t rect_index_row_bytes(int H, int W) {
    return (size_t)kRectListLen * 8 + ((rect_tiles_per_map(H, W) * 2 + 15) & ~(size_t)15);
}
// one workgroup per map: hash the distinct rectangles of the valid records (1024 slots, linear probing, 64-bit LDS CAS),
// rank the occupied slots, write the list and the tile pairs.  Which rectangle gets which index depends on the order of
// insertion; no result depends on it.
constexpr int kRectHash = 1024;
__global__ __launch_bounds__(256) void rect_index_kernel(const uint4* __restrict__ table, int H, int W, char* __restrict__ rows,
                                                         int32_t* __restrict__ n_rects,
                                                         const int* __restrict__ n_live, const int* __restrict__ list) {
    __shared__ unsigned long long keys[kRectHash];
    __shared__ unsigned short rank_s[kRectHash];
    __shared__ int wave_tot[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    if (n_live && m >= *n_live) return;
    const size_t row = list ? (size_t)list[m] : (size_t)m;
    const int n_tiles = (int)rect_tiles_per_map(H, W);
    const uint4* tab = table + row * (size_t)n_tiles;
    char* out = rows + row * rect_index_row_bytes(H, W);
    constexpr unsigned long long kEmpty = ~0ull;
    for (int k = tid; k < kRectHash; k += 256) keys[k] = kEmpty;
    __syncthreads();
    auto slot_of = [&](unsigned lo, unsigned hi, bool insert) -> int {
        const unsigned long long key = ((unsigned long long)hi << 32) | lo;
        unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ULL) >> 54);         // 10 bits
        for (int tries = 0; tries < kRectHash; ++tries, h = (h + 1) & (kRectHash - 1)) {
            unsigned long long cur = keys[h];
            if (cur == key) return (int)h;
            if (cur == kEmpty) {
                if (!insert) return -1;
                cur = atomicCAS(&keys[h], kEmpty, key);
                if (cur == kEmpty || cur == key) return (int)h;
            }
        }
        return -1;
    };
    for (int t = tid; t < n_tiles; t += 256) {
        const uint4 rec = tab[t];
        if (rect_record_invalid(rec)) continue;
        slot_of(rec.x, rec.y, true);
        slot_of(rec.z, rec.w, true);
    }
    __syncthreads();
    // ranks of the occupied slots in slot order (4 slots per thread, wave scan, 4 wave totals)
    int occ[4], mine = 0;
    for (int j = 0; j < 4; ++j) { occ[j] = keys[tid * 4 + j] != kEmpty; mine += occ[j]; }
    int incl = mine;
    const int lane = tid & 63;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; ++w) { before += (w < (tid >> 6)) ? wave_tot[w] : 0; total += wave_tot[w]; }
    int r = before + incl - mine;
    uint2* lst = (uint2*)out;
    for (int j = 0; j < 4; ++j) {
        unsigned short rk = 0xFFFF;
        if (occ[j]) {
            if (r < kRectListLen - 1) {
                rk = (unsigned short)r;
                const unsigned long long key = keys[tid * 4 + j];
                lst[r] = make_uint2((unsigned)key, (unsigned)(key >> 32));
            }
            ++r;
        }
        rank_s[tid * 4 + j] = rk;
    }
    for (int k = total + tid; k < kRectListLen; k += 256) lst[k] = make_uint2(0u, 0u);      // unused entries stay defined
    if (tid == 0 && n_rects) n_rects[row] = total;
    __syncthreads();
    uint16_t* pair = (uint16_t*)(out + (size_t)kRectListLen * 8);
    for (int t = tid; t < n_tiles; t += 256) {
        const uint4 rec = tab[t];
        unsigned p = kRectNoIndex;
        if (!rect_record_invalid(rec)) {
            const int sa = slot_of(rec.x, rec.y, false), sb = slot_of(rec.z, rec.w, false);
            const unsigned ra = sa >= 0 ? rank_s[sa] : 0xFFFFu, rb = sb >= 0 ? rank_s[sb] : 0xFFFFu;
            if (ra < kRectListLen - 1 && rb < kRectListLen - 1) p = ra | (rb << 8);
        }
        pair[t] = (uint16_t)p;
    }
}

// navsim_maps_closed: is every cell of the map's outer ring of kRectClosedRing cells occupied?  One workgroup per map.
// (What cfg.closed_maps may be set from: the LDS form of the march carries no bounds test, kernels_step.hpp probe_round.)
constexpr int kRectClosedRing = 3;
__global__ __launch_bounds__(256) void maps_closed_kernel(const uint8_t* __restrict__ occ, int H, int W, int32_t* __restrict__ closed) {
    const int m = blockIdx.x, tid = threadIdx.x, ring = kRectClosedRing;
    const uint8_t* o = occ + (size_t)m * H * W;
    int open_cells = 0;
    if (H <= 2 * ring || W <= 2 * ring) open_cells = 1;
    else {
        // the ring as 2 * ring full rows + 2 * ring columns of the rows in between
        const int n_row_cells = 2 * ring * W, n_col_cells = 2 * ring * (H - 2 * ring);
        for (int k = tid; k < n_row_cells + n_col_cells; k += 256) {
            int x, y;
            if (k < n_row_cells) { const int r = k / W; x = k - r * W; y = r < ring ? r : H - 2 * ring + r; }
            else { const int q = k - n_row_cells, r = q / (2 * ring), cidx = q - r * 2 * ring; y = ring + r; x = cidx < ring ? cidx : W - 2 * ring + cidx; }
            if (!o[(size_t)y * W + x]) open_cells = 1;
        }
    }
    open_cells = __syncthreads_or(open_cells);
    if (tid == 0) closed[m] = open_cells ? 0 : 1;
}

// navsim_world_closed: the same test on a world's DISTANCE FIELD (a cell is occupied exactly when its distance is 0), one
// workgroup per arena; *n_open counts the arenas whose ring has a free cell.  What checks a caller's cfg.closed_maps
// (round-4 advisor: the assertion was never verified against the world it was made about).
template <typename Field>
__global__ __launch_bounds__(256) void world_closed_kernel(const void* __restrict__ field, const float* __restrict__ overflow,
                                                           int H, int W, int32_t* __restrict__ n_open,
                                                           const int32_t* __restrict__ map_slot) {
    const int e = blockIdx.x, tid = threadIdx.x, ring = kRectClosedRing;
    const Field f(field, overflow, map_slot ? map_slot[e] : e, H, W);
    int open_cells = 0;
    if (H <= 2 * ring || W <= 2 * ring) open_cells = 1;
    else {
        const int n_row_cells = 2 * ring * W, n_col_cells = 2 * ring * (H - 2 * ring);
        for (int k = tid; k < n_row_cells + n_col_cells; k += 256) {
            int x, y;
            if (k < n_row_cells) { const int r = k / W; x = k - r * W; y = r < ring ? r : H - 2 * ring + r; }
            else { const int q = k - n_row_cells, r = q / (2 * ring), cidx = q - r * 2 * ring; y = ring + r; x = cidx < ring ? cidx : W - 2 * ring + cidx; }
            if (!f.occupied(f.load(x, y))) open_cells = 1;
        }
    }
    open_cells = __syncthreads_or(open_cells);
    if (tid == 0 && open_cells) atomicAdd(n_open, 1);
}

// ---- builder pass 1: transpose of the occupancy grid (32x32 tiles through LDS), so that the vertical runs can be
// found by the same coalesced row kernel
__global__ __launch_bounds__(256) void rect_transpose_kernel(const uint8_t* __restrict__ occ, uint8_t* __restrict__ occT,
                                                             int H, int W, const int* __restrict__ n_live) {
    __shared__ uint8_t t[32][33];
    const size_t m = blockIdx.z;
    if (n_live && (int)m >= *n_live) return;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;               // 32 x 8 threads
    const uint8_t* o = occ + m * (size_t)H * W;
    uint8_t* oT = occT + m * (size_t)H * W;
    for (int r = ty; r < 32; r += 8) {
        const int x = bx + tx, y = by + r;
        t[r][tx] = (x < W && y < H) ? o[(size_t)y * W + x] : 0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int y = by + tx, x = bx + r;                                // occT[x][y], H entries per row
        if (x < W && y < H) oT[(size_t)x * H + y] = t[tx][r];
    }
}

// ---- builder pass 2: maximal runs of occupied cells along a row.  One workgroup per row; run_lo[x] / run_hi[x] =
// first / last cell of the run containing x, or (32767, -1) for a free cell.  Parallel: every occupied cell
// looks for the nearest free cell on either side with two max / min scans in LDS.
__global__ __launch_bounds__(256) void rect_runs_kernel(const uint8_t* __restrict__ occ_h, int16_t* __restrict__ lo_h,
                                                        int16_t* __restrict__ hi_h, const uint8_t* __restrict__ occ_v,
                                                        int16_t* __restrict__ lo_v, int16_t* __restrict__ hi_v,
                                                        int H, int W, const int* __restrict__ n_live) {
    extern __shared__ int16_t runs_lds[];                                // [2][len]: last free <= x, first free >= x
    const size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;
    // blockIdx.z = 0: rows of the grid (horizontal runs); 1: rows of the transposed grid (vertical runs)
    const bool vert = blockIdx.z != 0;
    const uint8_t* occ = vert ? occ_v : occ_h;
    int16_t* run_lo = vert ? lo_v : lo_h;
    int16_t* run_hi = vert ? hi_v : hi_h;
    const int n_rows = vert ? W : H, len = vert ? H : W;
    const int row = blockIdx.x;
    if (row >= n_rows) return;
    const uint8_t* o = occ + (m * (size_t)n_rows + row) * len;
    int16_t* lastf = runs_lds;
    int16_t* nextf = runs_lds + len;
    for (int x = threadIdx.x; x < len; x += blockDim.x) {
        const bool f = o[x] == 0;
        lastf[x] = f ? (int16_t)x : (int16_t)-1;
        nextf[x] = f ? (int16_t)x : (int16_t)len;
    }
    __syncthreads();
    for (int off = 1; off < len; off <<= 1) {                            // Hillis-Steele max / min scans
        int16_t a[4], b[4];                                              // len <= 4 * blockDim.x (checked by the host)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = (int)threadIdx.x + k * (int)blockDim.x;
            a[k] = (x < len && x >= off) ? lastf[x - off] : (int16_t)-1;
            b[k] = (x + off < len) ? nextf[x + off] : (int16_t)len;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = (int)threadIdx.x + k * (int)blockDim.x;
            if (x < len) {
                if (a[k] > lastf[x]) lastf[x] = a[k];
                if (b[k] < nextf[x]) nextf[x] = b[k];
            }
        }
        __syncthreads();
    }
    int16_t* lo = run_lo + (m * (size_t)n_rows + row) * len;
    int16_t* hi = run_hi + (m * (size_t)n_rows + row) * len;
    for (int x = threadIdx.x; x < len; x += blockDim.x) {
        const bool f = o[x] == 0;
        lo[x] = f ? (int16_t)32767 : (int16_t)(lastf[x] + 1);
        hi[x] = f ? (int16_t)-1 : (int16_t)(nextf[x] - 1);
    }
}

// exact integer d2 of a cell from the arena's distance field (the verification target)
struct RectD2FromU16T {
    const uint16_t* p; const float* ovf; int W, tpr;
    __device__ __forceinline__ int at(int px, int py) const {
        const unsigned v = p[FieldU16T::index(px, py, tpr)];
        if (v != 0xFFFFu) return (int)v;
        if (!ovf) return -1;                                             // saturated and no exact plane: unknown
        const double f = (double)ovf[(size_t)py * W + px];
        return (int)(f * f + 0.5);
    }
};
struct RectD2FromF32 {
    const float* p; int W;
    __device__ __forceinline__ int at(int px, int py) const {
        const double f = (double)p[(size_t)py * W + px];
        return (f * f < 2.0e9) ? (int)(f * f + 0.5) : -1;               // "no obstacle anywhere": unknown
    }
};

struct RectRunArrays {
    const uint8_t* occ;         // [H][W]
    const int16_t* hl; const int16_t* hr;      // [H][W] horizontal run of every occupied cell
    const int16_t* vt; const int16_t* vb;      // [W][H] vertical run, TRANSPOSED storage
    int H, W;
};

// Maximal occupied rectangle around the occupied cell (ox, oy), whole wavefront cooperating.  order 0: the cell's
// horizontal run, extended up and down while the rows' runs through column ox cover it; order 1: the vertical run,
// extended left and right.  Lanes 0..31 look one way, lanes 32..63 the other, 32 rows (columns) per round.
__device__ __forceinline__ void rect_grow(const RectRunArrays& r, int ox, int oy, int order, int lane,
                                          int& x0, int& x1, int& y0, int& y1) {
    const int half = lane & 31;
    const bool fwd = lane >= 32;
    if (order == 0) {
        const int xl = r.hl[(size_t)oy * r.W + ox], xr = r.hr[(size_t)oy * r.W + ox];
        int up = 0, dn = 0;
        bool go_up = true, go_dn = true;
        while (go_up || go_dn) {
            const int y = fwd ? oy + 1 + dn + half : oy - 1 - up - half;
            bool ok = false;
            if ((fwd ? go_dn : go_up) && y >= 0 && y < r.H)
                ok = r.hl[(size_t)y * r.W + ox] <= xl && r.hr[(size_t)y * r.W + ox] >= xr;
            const unsigned long long b = __ballot(ok);
            const unsigned bu = (unsigned)b, bd = (unsigned)(b >> 32);
            if (go_up) { int n = (bu == 0xFFFFFFFFu) ? 32 : __builtin_ctz(~bu); up += n; go_up = n == 32; }
            if (go_dn) { int n = (bd == 0xFFFFFFFFu) ? 32 : __builtin_ctz(~bd); dn += n; go_dn = n == 32; }
        }
        x0 = xl; x1 = xr; y0 = oy - up; y1 = oy + dn;
    } else {
        const int yt = r.vt[(size_t)ox * r.H + oy], yb = r.vb[(size_t)ox * r.H + oy];
        int lf = 0, rt = 0;
        bool go_l = true, go_r = true;
        while (go_l || go_r) {
            const int x = fwd ? ox + 1 + rt + half : ox - 1 - lf - half;
            bool ok = false;
            if ((fwd ? go_r : go_l) && x >= 0 && x < r.W)
                ok = r.vt[(size_t)x * r.H + oy] <= yt && r.vb[(size_t)x * r.H + oy] >= yb;
            const unsigned long long b = __ballot(ok);
            const unsigned bl = (unsigned)b, br = (unsigned)(b >> 32);
            if (go_l) { int n = (bl == 0xFFFFFFFFu) ? 32 : __builtin_ctz(~bl); lf += n; go_l = n == 32; }
            if (go_r) { int n = (br == 0xFFFFFFFFu) ? 32 : __builtin_ctz(~br); rt += n; go_r = n == 32; }
        }
        y0 = yt; y1 = yb; x0 = ox - lf; x1 = ox + rt;
    }
}

// A nearest obstacle cell of (px, py) given its exact d2: some lattice point of the circle dx^2 + dy^2 = d2 around
// the cell is occupied (that is what d2 means).  The wavefront scans dx = 0, 1, ... 64 values per round.
// Returns false only for inconsistent input.
__device__ __forceinline__ bool rect_nearest_obstacle(const RectRunArrays& r, int px, int py, int d2, int lane,
                                                      int& ox, int& oy) {
    if (d2 == 0) { ox = px; oy = py; return true; }
    const int dmax = (int)sqrtf((float)d2) + 1;
    for (int base = 0; base <= dmax; base += 64) {
        const int dx = base + lane;
        int fx = -1, fy = -1;
        const int rem = d2 - dx * dx;
        if (rem >= 0) {
            int dy = (int)sqrtf((float)rem);
            while (dy * dy > rem) --dy;
            while ((dy + 1) * (dy + 1) <= rem) ++dy;
            if (dy * dy == rem) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int x = px + ((s & 1) ? -dx : dx), y = py + ((s & 2) ? -dy : dy);
                    if (fx < 0 && x >= 0 && x < r.W && y >= 0 && y < r.H && r.occ[(size_t)y * r.W + x]) { fx = x; fy = y; }
                }
            }
        }
        const unsigned long long b = __ballot(fx >= 0);
        if (b) {
            const int src = __builtin_ctzll(b);
            ox = __shfl(fx, src, 64);
            oy = __shfl(fy, src, 64);
            return true;
        }
    }
    return false;
}

// ---- builder pass 3: one wavefront per tile, one lane per cell.  Up to two greedy rounds: candidates are the
// maximal rectangles (both growth orders) around a nearest obstacle of the first and of the last uncovered cell;
// the one reproducing d2 on most uncovered cells is kept.  The record is valid when both rounds together cover
// every in-map cell of the tile -- i.e. it has been checked cell by cell against the exact field.
template <typename D2Src>
__device__ __forceinline__ void rect_tile_build(const D2Src& src, const RectRunArrays& r, int tile, uint4* __restrict__ out) {
    const int tpr = (r.W + 7) >> kRectShift;
    const int ty = tile / tpr, tx = tile - ty * tpr;
    const int lane = threadIdx.x & 63;
    const int px = (tx << kRectShift) + (lane & 7), py = (ty << kRectShift) + (lane >> 3);
    const bool in_map = px < r.W && py < r.H;
    const int d2 = in_map ? src.at(px, py) : 0;
    const unsigned p = ((unsigned)py << 16) | (unsigned)px;
    unsigned long long uncovered = __ballot(in_map);
    bool fail = __ballot(in_map && d2 < 0) != 0;                           // some cell's distance is unknown
    unsigned rec[4] = {0, 0, 0, 0};
    int nrect = 0;
    for (int round = 0; round < 2 && uncovered && !fail; ++round) {
        unsigned best_lo = 0, best_hi = 0;
        unsigned long long best_cov = 0;
        const int first = __builtin_ctzll(uncovered), last = 63 - __builtin_clzll(uncovered);
        for (int which = 0; which < 2 && best_cov != uncovered; ++which) {
            const int leader = which ? last : first;
            if (which && last == first) break;                             // one uncovered cell: same candidates
            const int lpx = __shfl(px, leader, 64), lpy = __shfl(py, leader, 64), ld2 = __shfl(d2, leader, 64);
            int ox, oy;
            if (!rect_nearest_obstacle(r, lpx, lpy, ld2, lane, ox, oy)) { fail = true; break; }
            // start with the growth order whose RUN is the longer one: the other direction is then a wall's
            // thickness (one round of the growth loop) instead of its length (one round per 32 cells)
            const int hlen = r.hr[(size_t)oy * r.W + ox] - r.hl[(size_t)oy * r.W + ox];
            const int vlen = r.vb[(size_t)ox * r.H + oy] - r.vt[(size_t)ox * r.H + oy];
            const int first_order = (hlen >= vlen) ? 0 : 1;
            for (int oi = 0; oi < 2; ++oi) {
                const int order = first_order ^ oi;
                int x0, x1, y0, y1;
                rect_grow(r, ox, oy, order, lane, x0, x1, y0, y1);
                const unsigned lo = ((unsigned)y0 << 16) | (unsigned)x0, hi = ((unsigned)y1 << 16) | (unsigned)x1;
                const unsigned long long cov = __ballot(in_map && rect_dist2(lo, hi, p) == d2) & uncovered;
                if (__popcll(cov) > __popcll(best_cov)) { best_cov = cov; best_lo = lo; best_hi = hi; }
                if (cov == uncovered) break;                               // nothing left to gain
            }
        }
        if (fail || best_cov == 0) { fail = true; break; }
        rec[2 * nrect] = best_lo; rec[2 * nrect + 1] = best_hi;
        ++nrect;
        uncovered &= ~best_cov;
    }
    if (uncovered || fail || nrect == 0) { rec[0] = kRectInvalid; rec[1] = 0; rec[2] = 0; rec[3] = 0; }
    else if (nrect == 1) { rec[2] = rec[0]; rec[3] = rec[1]; }
    if (lane == 0) out[tile] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

// format 0: float32 field, 1: packed uint16 tiles (+ optional overflow plane).  Map m of the batch reads field /
// writes table row `list ? list[m] : m` (navsim_regen rebuilds the records of the arenas it regenerated, whose new
// fields still sit in its scratch, indexed by slot: field_by_slot = 1).
__global__ __launch_bounds__(256) void rect_tiles_kernel(const uint8_t* __restrict__ occ, const int16_t* __restrict__ hl,
                                                         const int16_t* __restrict__ hr, const int16_t* __restrict__ vt,
                                                         const int16_t* __restrict__ vb, int H, int W,
                                                         const void* __restrict__ field, const float* __restrict__ overflow,
                                                         int format, size_t field_bytes_per_map, uint4* __restrict__ table,
                                                         const int* __restrict__ n_live, const int* __restrict__ list) {
    const size_t m = blockIdx.y;
    if (n_live && (int)m >= *n_live) return;
    const int n_tiles = (int)rect_tiles_per_map(H, W);
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const size_t cells = (size_t)H * W;
    RectRunArrays r = {occ + m * cells, hl + m * cells, hr + m * cells, vt + m * cells, vb + m * cells, H, W};
    const size_t row = list ? (size_t)list[m] : m;
    uint4* out = table + row * (size_t)n_tiles;
    const char* f = (const char*)field + m * field_bytes_per_map;
    if (format == NAVSIM_FIELD_U16T) {
        RectD2FromU16T src = {(const uint16_t*)f, overflow ? overflow + m * cells : nullptr, W, (W + 7) >> 3};
        rect_tile_build(src, r, tile, out);
    } else {
        RectD2FromF32 src = {(const float*)f, W};
        rect_tile_build(src, r, tile, out);
    }
}
