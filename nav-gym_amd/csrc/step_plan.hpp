// step_plan.hpp -- launch geometry of the fused step (threads per arena, where the probes find the rect records, dynamic
// LDS), shared by the C ABI (navsim_kernels.hip) and the units that instantiate the step kernel (navsim_step_inst.hip).
// Included inside the anonymous namespace, behind kernels_step.hpp.
// kernels that want more than 64 KB of dynamic LDS must say so once per (device, kernel); more than the CU has is refused.
// What has been granted is remembered (round-4 advisor: every call above 64 KB used to repeat hipFuncSetAttribute, also
// inside a hipGraph capture): after navsim_prepare no later launch of the same configuration touches an attribute.
constexpr size_t kLdsPerCu = 160 * 1024;
int allow_lds(const void* kernel, size_t lds) {
    if (lds <= 64 * 1024) return NAVSIM_OK;
    if (lds > kLdsPerCu) return NAVSIM_E_UNSUPPORTED;
    struct Granted { const void* kernel; int device; size_t lds; };
    constexpr int kSlots = 256;
    static Granted table[kSlots];
    static int used = 0;
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> guard(lock);
    Granted* slot = nullptr;
    for (int i = 0; i < used; ++i)
        if (table[i].kernel == kernel && table[i].device == dev) { slot = &table[i]; break; }
    if (slot && slot->lds >= lds) return NAVSIM_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NAVSIM_E_UNSUPPORTED;
    if (!slot && used < kSlots) { slot = &table[used++]; slot->kernel = kernel; slot->device = dev; }
    if (slot) slot->lds = lds;              // (a full table only means the attribute is set again next time)
    return NAVSIM_OK;
}

// compute units of the CURRENT device (cached per device ordinal; idempotent, so a race only repeats the query)
int device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// dynamic LDS of the fused step: parked rays; beam directions + ranges (pedestrian variants merge from LDS); PedShared
// rays a wavefront parks per chunk (kernels_step.hpp "Parking"): only where a launch runs several generations of
// 256-thread workgroups, and not in the pedestrian variants
int step_park_lanes(const navsim_config* c, int block) {
    const bool peds = c->ped_model != NAVSIM_PED_NONE;
    if (!step_parks(block, peds)) return 0;
    if (peds) return c->n_beams <= 65535 ? NAVSIM_PARK_LANES_PEDS : 0;      // the park area keeps 16-bit beam indices there
    return kParkLanesMax;
}
size_t step_lds_scan_bytes(const navsim_config* c, int park_lanes) {
    const bool peds = c->ped_model != NAVSIM_PED_NONE;
    return park_lds_bytes(c->n_beams, park_lanes, peds) + (peds ? (size_t)c->n_beams * (sizeof(float2) + sizeof(float)) : 0);
}
size_t ped_update_lds_bytes(const navsim_config* c) {                    // ped_update_kernel: a pack of arenas per wavefront
    return (size_t)ped_pack(c->max_peds) * ped_slot_bytes(c->max_peds);
}
// pedestrians ahead of the step, a pack of arenas per workgroup (ped_update_kernel), instead of inside it: only on request
// (cfg.ped_split = 2).  Round 2 split large batches automatically (the fused phase held three of four wavefronts at a
// barrier: c3 13.2 -> 14.0 M env-steps/s); since the phase runs on wavefront 0 BESIDE the scan of the others the fused
// form wins everywhere (c3, same box: split 21.0, fused 21.9 M; c5 fused 3.84 -> 4.28 M).
bool ped_split_on(const navsim_config* c) {
    if (c->ped_model == NAVSIM_PED_NONE || ped_update_lds_bytes(c) > 64 * 1024) return false;
    return c->ped_split == 2;
}
size_t step_lds_bytes(const navsim_config* c, int park_lanes) {
    size_t lds = step_lds_scan_bytes(c, park_lanes);
    if (c->ped_model != NAVSIM_PED_NONE) {
        lds = ((lds + 15) & ~(size_t)15) + ped_lds_bytes(c->max_peds);
        // the fused pedestrian phase keeps its pair table behind PedShared (kernels_step.hpp ped_phase_wave)
        if (!ped_split_on(c)) lds = ((lds + 15) & ~(size_t)15) + ped_pair_bytes(c->max_peds);
    }
    return lds;
}

// Threads per arena (cfg.step_block = 0).  With >= 12 arenas per CU the chip is kept full by 256-thread workgroups (8 per
// CU, several generations).  With fewer, wider workgroups shorten a launch that is as long as its slowest workgroup's own
// march: 1024 threads while ALL arenas are resident at two per CU (<= 2 arenas per CU), else 512.
// Round 4 re-sweep with the index rows in LDS (profiles/r04_blocks/, M env-steps/s at 256 / 512 / 1024 threads), c2 world:
// 512 arenas 11.9 / 14.9 / 15.3; 640: - / 17.0 / 15.7; 768: 16.8 / 20.6 / 20.0; 1024: 21.2 / 25.0 / 22.4; 1536: 28.3 / 31.4 / 25.0;
// 2048: 33.0 / 34.5 / 26.2; 3072: 40.2 / 37.5 / 27.2; 4096: 44.2 / 39.3 / 27.8.  c3 world (20 pedestrians): 512 arenas
// - / 10.4 / 10.3; 640: - / 12.0 / 9.3; 1024: - / 16.0 / 11.9; 1536: 16.1 / 18.3 / -; 2048: 19.4 / 20.2 / -; 3072: 22.3 / 22.2 / -.
// (Rounds 2-3 took 1024 threads up to 6 arenas per CU: a second generation of 1024-thread workgroups costs more than it saves.)
int pick_step_block(const navsim_config* c) {
    if (c->step_block) return c->step_block;
    const int B = c->n_beams;
    const long cus = device_cu_count();
    if (B <= 64) return 64;
    if ((long)c->n_envs >= 12 * cus || B <= 256) return 256;
    if ((long)c->n_envs > 2 * cus || B <= 512) return 512;
    return 1024;
}

// which compiled form of the march step serves cfg.march_rule (kernels_field.hpp march_step): the float32-only
// evaluation of NAVSIM_MARCH_F64 where every distance is sqrtf of an integer below 2^22
int march_rule_variant(const navsim_config* c) {
    if (c->march_rule == NAVSIM_MARCH_F32 || c->march_rule == NAVSIM_MARCH_F32_FMA) return c->march_rule;
    const int side = c->map_h > c->map_w ? c->map_h : c->map_w;
    return (c->field_format == NAVSIM_FIELD_U16T && side <= 1448) ? kMarchF64Exact32 : NAVSIM_MARCH_F64;
}

// How a step is launched: threads per arena, where the probes find the rect records, the dynamic LDS.
struct StepPlan {
    int block;              // threads per arena
    int rect;               // 0 no rect records, 1 records read from global memory, 2 the arena's table staged in LDS
    int park;               // rays a wavefront parks per chunk
    size_t lds;             // dynamic LDS per workgroup
    unsigned rect_off;      // byte offset of the staged table inside it
};
// The record table in LDS ("map tiles staged through LDS").  Round 3 staged the 16-byte records themselves: 63.5 KB per
// 500 x 500 arena, two 1024-thread workgroups per CU, +10-13 % for launches of up to 4 arenas per CU and a loss beyond
// (profiles/r03_rect_lds/).  Round 4 stages the INDEX form (kernels_rect.hpp: 10 KB) at the residency the block size implies
// anyway; measured, records in global memory -> index rows in LDS (profiles/r04_idx/ab.txt, M env-steps/s): c2 36.4 -> 40.9,
// 1024 arenas 17.8 -> 21.4, 512 arenas 13.3 -> 14.4, c4 28.2 -> 33.8, c5 4.36 -> 4.73.
StepPlan plan_step(const navsim_config* c, const navsim_state* st, int grid = 0) {
    StepPlan p;
    p.block = pick_step_block(c);
    p.rect = st->rect_table ? 1 : 0;
    p.park = step_park_lanes(c, p.block);
    p.lds = step_lds_bytes(c, p.park);
    p.rect_off = 0;
    // The index form of the arena's table staged in LDS (kernels_rect.hpp; round 4): 10 KB for a 500 x 500 map, so it fits at
    // the residency the block size already implies -- eight 256-thread, four 512-thread or two 1024-thread workgroups per CU --
    // and every probe reads LDS.  (Round 3 staged the 16-byte records, 63.5 KB: two workgroups per CU, small launches only.)
    if (p.rect && st->rect_index && c->closed_maps && c->rect_lds != 1 && c->field_format == NAVSIM_FIELD_U16T) {
        const size_t row = rect_index_row_bytes(c->map_h, c->map_w);
        const size_t base = (p.lds + 15) & ~(size_t)15;
        const size_t total = base + row + 1024;                     // + the kernel's static LDS, allocation granules
        // workgroups per CU the rows must leave room for: all that the wave slots allow at 512 / 1024 threads; at 256 threads
        // five of the eight are enough -- c3 (20 pedestrians: 20 KB of scan copy, pedestrian scratch and pair table per
        // arena) fits five with the rows and runs 24.3 M env-steps/s against 23.7 M at eight with the records in global
        // memory (profiles/r04_idx/ab2.txt, r04_defer/ab_c3_hazardfix.txt)
        const int per_cu = p.block >= 1024 ? 2 : (p.block == 512 ? 4 : 5);
        // a launch of at most one workgroup per CU (navsim_regen's first observations: a handful of lone scans) may take
        // a CU's whole LDS
        const bool lone = grid > 0 && grid <= device_cu_count();
        const bool fits = c->rect_lds == 2 || lone ? total <= kLdsPerCu : (size_t)per_cu * total <= kLdsPerCu;
        if (fits) { p.rect = 2; p.lds = base + row; p.rect_off = (unsigned)row; }     // the row comes first, everything else behind it
    }
    return p;
}
