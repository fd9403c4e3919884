// kernels_team.hpp -- the fused step of worlds without pedestrians as a TEAM of wavefronts per workgroup.
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace; not a standalone
// header).
//
// navsim_step_kernel gives every arena its own workgroup: a launch of 4096 arenas is two generations of 2048 resident
// workgroups, and it ends when the slot that held the longest arena has also finished the shortest one -- the last
// third of the launch runs on a draining chip (profiles/README.md, occupancy timeline: 78 % of the slots on average).
// Here a workgroup is 16 wavefronts that keep kTeamSlots arenas in flight and never meet at a barrier: a wavefront
// looks for the next piece of work in LDS -- set an arena up (what thread 0 of the per-arena kernel does), march a
// 64-beam chunk of any arena in flight, march a group of parked rays, or finish an arena (reward / done / rescan /
// pack) -- and arenas are drawn from one counter in global memory in launch order.  At the end of the launch all 16
// wavefronts of a workgroup work on its last arenas, so the chip drains within one arena's work / 16 instead of one
// arena's lifetime.  Every per-beam and per-arena computation is the per-arena kernel's; only who executes it, and
// when, differs, which changes no result.
//
// Data a wavefront hands to another one goes through LDS (slot fields, parked rays) or through global memory written
// and read inside ONE workgroup (the observation row, for the discomfort ratio); across workgroups there is only the
// arena counter (agent-scope atomics).

constexpr int kTeamSlots = 4;                        // arenas in flight per workgroup
constexpr int kTeamThreads = 1024;                   // 16 wavefronts: two workgroups per CU at 8 waves per SIMD
enum { kSlotEmpty = 0, kSlotBusy = 1, kSlotScan = 2, kSlotDead = 3 };
enum { kTeamNone = 0, kTeamSetup, kTeamChunk, kTeamParked, kTeamFinish, kTeamExit };

struct TeamSlot {
    int state;                                       // kSlot*
    int hint;                                        // 1: a look at this slot may find work (cheap to poll)
    int pollers;                                     // wavefronts inside their look at this slot (see team_step_kernel)
    int e;                                           // arena
    int next_chunk, done_chunks;                     // 64-beam chunks handed out / completed
    int park_count, park_next, groups_done;          // parked rays written / handed out, groups of 64 completed
    int flags;                                       // scan A: bit 0 crash, bit 1 discomfort
    int stage;                                       // 0: scan A, 1: scan B (reward already written)
    int respawn;
    int n_hist;
    float noise_std;
    float lx, ly, lth, t1, r_all;                    // lidar pose (float32, env.py:386), shared first probe
    int i0, j0;
    double cT, sT;
    unsigned long long nkey;                         // noise stream of the scan in flight
    unsigned long long step_key;
    unsigned long long t_begin;
    double rp[3], act[2];
};

// [0]: next position of the launch order, [1]: wavefronts that have left; the last one to leave zeroes both
__device__ unsigned g_team_counter[2];

// slot words other wavefronts act on: sequentially consistent at workgroup scope (LDS serves a wavefront's
// instructions in order; the ordering keeps the compiler from moving one such access across another)
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool lds_cas(int* p, int expect, int v) {
    return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A value read from a slot is the same in every lane, but only a readfirstlane tells the compiler: without it the arena
// index and everything derived from it (field and record bases, the lidar origin) live in vector registers and the
// address arithmetic of the probe loop runs on the vector units (measured: twice the vector instructions).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double uni(double v) { return __longlong_as_double((long long)uni((unsigned long long)__double_as_longlong(v))); }
__device__ __forceinline__ void team_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void team_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// the lidar side of a pose: float32 pose, integer origin, sin / cos of the heading, the shared first probe
template <int RULE, typename Field>
__device__ __forceinline__ void team_lidar_pose(const navsim_config& c, const Field& field, TeamSlot& s) {
    s.lx = (float)s.rp[0]; s.ly = (float)s.rp[1]; s.lth = (float)s.rp[2];        // env.py:386
    int i0, j0;
    nv::xy_to_ij_f32(s.lx, s.ly, c, i0, j0);                                      // env.py:419
    s.i0 = i0; s.j0 = j0;
    double sT, cT;
    nv::sincos((double)s.lth, sT, cT);
    s.sT = sT; s.cT = cT;
    float t1, r_all;
    first_probe<RULE>(field, i0, j0, (float)((long long)c.map_h * c.map_w), t1, r_all);
    s.t1 = t1; s.r_all = r_all;
}

// one 64-beam chunk (beam0 >= 0) or one group of parked rays (beam0 < 0, entries [g, g + 64) of the slot's queue)
template <typename Field, int RULE, bool RECT>
__device__ __forceinline__ void team_march(const navsim_config& c, const navsim_state& st, const navsim_step_io& io,
                                           TeamSlot& s, float4* __restrict__ park, int chunk, int group, int lane) {
    const int B = c.n_beams, S = c.n_scan_stack, H = c.map_h, W = c.map_w, D = S * B + 7;
    const int e = uni(s.e);
    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, H, W);
    const char* rects = RECT ? (const char*)st.rect_table + (size_t)(c.shared_field ? 0 : e) * rect_tiles_per_map(H, W) * sizeof(uint4)
                             : nullptr;
    float* obs_row = io.obs + (size_t)e * D;
    const float* thr = st.scan_threshold;
    const float* dthr = st.scan_discomfort;
    const float max_range = march_limit(H, W, c.range_max, c.resolution);
    const float res = (float)c.resolution, rmax = (float)c.range_max;
    const double step = nv::linspace_step(c);
    const float x0 = (float)uni(s.i0), y0 = (float)uni(s.j0);
    const unsigned uW = (unsigned)W, uH = (unsigned)H;
    const unsigned tpr = (unsigned)((W + 7) >> kRectShift);
    const float t1 = uni(s.t1), r_all = uni(s.r_all), noise_std = uni(s.noise_std);
    const uint64_t nkey = uni(s.nkey);
    const int n_hist = uni(s.n_hist);
    const double lth = (double)uni(s.lth), cT = uni(s.cT), sT = uni(s.sT);
    const float miss = (r_all >= 0.0f) ? r_all : max_range;
    int cr = 0, dc = 0;
    auto finish = [&](int k, float rr) {
        rr = rr * res;                                          // env.py:426
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
    if (group < 0) {
        const int k = chunk * 64 + lane;
        const bool valid = k < B;
        float dx, dy;
        beam_dir_k(c, st.beam_table, valid ? k : B - 1, step, lth, cT, sT, dx, dy);
        float t = t1;
        lanemask_t active = mask_of(valid & (r_all < 0.0f));
        lanemask_t hit = 0;
        while ((int)__builtin_popcountll(active) > kParkLanesMax)
            probe_round<Field, RULE, RECT>(field, rects, tpr, x0, y0, dx, dy, uW, uH, max_range, t, active, hit);
        const bool marching = mask_lane(active);
        if (active != 0) {                                       // park what is still marching
            int base = 0;
            if (lane == 0) base = lds_add(&s.park_count, (int)__builtin_popcountll(active));
            base = __builtin_amdgcn_readfirstlane(base);
            if (marching) park[base + lanes_below(active)] = make_float4(__int_as_float(k), t, dx, dy);
        }
        if (valid & !marching) finish(k, ray_result(hit, x0, y0, dx, dy, t, miss));
    } else {
        const int n_parked = uni(lds_ld(&s.park_count));
        const bool valid = group + lane < n_parked;
        const float4 r = park[valid ? group + lane : group];
        const int k = __float_as_int(r.x);
        float t = r.y;
        const float dx = r.z, dy = r.w;
        lanemask_t active = mask_of(valid);
        lanemask_t hit = 0;
        while (active != 0)
            probe_round<Field, RULE, RECT>(field, rects, tpr, x0, y0, dx, dy, uW, uH, max_range, t, active, hit);
        if (valid) finish(k, ray_result(hit, x0, y0, dx, dy, t, miss));
    }
    const int f = (mask_of(cr != 0) != 0 ? 1 : 0) | (mask_of(dc != 0) != 0 ? 2 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wavefront's observation stores are out
    team_release();
    if (lane == 0) {
        if (f) __hip_atomic_fetch_or(&s.flags, f, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_add(group < 0 ? &s.done_chunks : &s.groups_done, 1);
        lds_st(&s.hint, 1);                                      // parked groups / the finish may be available now
    }
}

template <typename Field, int RULE, bool RECT>
__global__ __launch_bounds__(kTeamThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
void team_step_kernel(navsim_config c, navsim_state st, navsim_step_io io, int n_wavefronts_total) {
    __shared__ TeamSlot slots[kTeamSlots];
    __shared__ int dead_slots;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int B = c.n_beams, S = c.n_scan_stack, E = c.n_envs, D = S * B + 7;
    const int n_chunks = (B + 63) / 64;
    const size_t park_bytes = park_lds_bytes(B, kParkLanesMax);
    if (tid < kTeamSlots) { slots[tid].state = kSlotEmpty; slots[tid].pollers = 0; slots[tid].hint = 1; }
    if (tid == 0) dead_slots = 0;
    __syncthreads();                                            // the only barrier of the kernel

    for (unsigned idle = 0;;) {
        // ------------------------------------------------------------------ what is there to do?  (lane 0 decides)
        // An idle wavefront must cost next to nothing (a wavefront waiting at a barrier issues nothing; one that polls
        // takes issue cycles from the marching ones): it reads one hint word per slot and sleeps with a growing
        // interval.  A hint is raised by whoever makes work appear (after the work is visible) and lowered by a
        // wavefront that looked and found nothing -- which then looks once more, so that a hint raised in between is
        // not lost.
        int kind = kTeamNone, si = 0, arg = 0;
        if (lane == 0) {
            for (int i = 0; i < kTeamSlots && kind == kTeamNone; ++i) {
                const int q = (wid + i) % kTeamSlots;
                TeamSlot& s = slots[q];
                if (__hip_atomic_load(&s.hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) continue;
              for (int pass = 0; pass < 2 && kind == kTeamNone; ++pass) {
                if (pass == 1) lds_st(&s.hint, 0);
                // A wavefront that takes a slot over (scan -> busy) waits until nobody else is inside this look at
                // it: what another wavefront read here (counters of the scan that has just ended) must not be acted
                // on after the slot has been given its next scan or its next arena.
                lds_add(&s.pollers, 1);
                const int state = lds_ld(&s.state);
                if (state == kSlotScan) {
                    if (lds_ld(&s.next_chunk) < n_chunks) {
                        const int ch = lds_add(&s.next_chunk, 1);
                        if (ch < n_chunks) { kind = kTeamChunk; si = q; arg = ch; }
                    }
                    if (kind == kTeamNone && lds_ld(&s.done_chunks) == n_chunks) {   // its parked rays are all there
                        const int total = lds_ld(&s.park_count);
                        if (lds_ld(&s.park_next) < total) {
                            const int g = lds_add(&s.park_next, 64);
                            if (g < total) { kind = kTeamParked; si = q; arg = g; }
                        }
                        if (kind == kTeamNone && lds_ld(&s.groups_done) == (total + 63) / 64 &&
                            lds_cas(&s.state, kSlotScan, kSlotBusy)) {
                            kind = kTeamFinish; si = q;
                            for (unsigned w = 0; lds_ld(&s.pollers) > 1 && w < (1u << 22); ++w) __builtin_amdgcn_s_sleep(1);
                        }
                    }
                } else if (state == kSlotEmpty) {
                    if (lds_cas(&s.state, kSlotEmpty, kSlotBusy)) { kind = kTeamSetup; si = q; }
                }
                lds_add(&s.pollers, -1);
              }
              if (kind != kTeamNone) lds_st(&s.hint, 1);          // there may be more of it for the next wavefront
            }
            if (kind == kTeamNone && lds_ld(&dead_slots) == kTeamSlots) kind = kTeamExit;
        }
        kind = __builtin_amdgcn_readfirstlane(kind);
        si = __builtin_amdgcn_readfirstlane(si);
        arg = __builtin_amdgcn_readfirstlane(arg);
        if (kind == kTeamExit) break;
        if (kind == kTeamNone) {
            if (idle < 2) __builtin_amdgcn_s_sleep(16); else if (idle < 4) __builtin_amdgcn_s_sleep(48); else __builtin_amdgcn_s_sleep(127);
            if (++idle > (1u << 20)) break;                     // never reached: a lost hand-off must not hang the GPU
            continue;
        }
        idle = 0;
        team_acquire();
        TeamSlot& s = slots[si];
        float4* park = (float4*)(dyn_lds + (size_t)si * park_bytes);

        if (kind == kTeamChunk) {
            team_march<Field, RULE, RECT>(c, st, io, s, park, arg, -1, lane);
        } else if (kind == kTeamParked) {
            team_march<Field, RULE, RECT>(c, st, io, s, park, 0, arg, lane);
        } else if (kind == kTeamSetup) {
            // -------------------------------------------------------------- phases 0-2 of the per-arena kernel
            if (lane == 0) {
                const unsigned pos = atomicAdd(&g_team_counter[0], 1u);
                if (pos >= (unsigned)E) {
                    lds_st(&s.state, kSlotDead);
                    lds_add(&dead_slots, 1);
                } else {
                    const int e = st.launch_order ? st.launch_order[pos] : (int)pos;
                    const uint64_t genv = (uint64_t)(c.env_index_base + e);
                    const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, c.map_h, c.map_w);
                    s.e = e;
                    s.t_begin = __builtin_amdgcn_s_memrealtime();
                    const double* rp_g = st.robot_pose + 3 * (size_t)e;
                    double a0 = io.action[2 * e], a1 = io.action[2 * e + 1];
                    st.steps[e] += 1;                          // env.py:592
                    if (c.min_turning_radius > 0.0) {          // env.py:595-600
                        double lim = fabs(a1) * c.min_turning_radius;
                        if (a0 >= 0.0) a0 = (a0 > lim) ? a0 : lim;
                        else           a0 = (a0 < -lim) ? a0 : -lim;
                    }
                    s.act[0] = a0; s.act[1] = a1;
                    double p[3] = {rp_g[0], rp_g[1], rp_g[2]};
                    nv::set_vel(p, a0, a1, c.time_step, c.axle_offset, nullptr);      // env.py:664
                    s.rp[0] = p[0]; s.rp[1] = p[1]; s.rp[2] = p[2];
                    team_lidar_pose<RULE>(c, field, s);
                    s.step_key = (unsigned long long)st.episode[e] * 0x100000000ULL + (unsigned long long)st.steps[e] * 2ULL;
                    s.noise_std = (c.add_scan_noise && st.scan_noise_std) ? st.scan_noise_std[e] : 0.0f;
                    s.nkey = (s.noise_std > 0.0f) ? nv::noise_stream(c.seed, genv, s.step_key) : 0;
                    s.n_hist = st.n_hist[e];
                    s.done_chunks = 0; s.park_count = 0; s.park_next = 0; s.groups_done = 0;
                    s.flags = 0; s.stage = 0; s.respawn = 0;
                    team_release();
                    lds_st(&s.next_chunk, 0);                  // the gates last: chunks can be taken from here on
                    lds_st(&s.state, kSlotScan);
                    lds_st(&s.hint, 1);
                }
            }
        } else {                                                // kTeamFinish
            // -------------------------------------------------------------- phases 4-6 of the per-arena kernel
            const int e = uni(s.e);
            const uint64_t genv = (uint64_t)(c.env_index_base + e);
            const Field field(st.field, st.field_overflow, c.shared_field ? 0 : e, c.map_h, c.map_w);
            float* obs_row = io.obs + (size_t)e * D;
            const float* obs_prev = io.obs_prev ? io.obs_prev + (size_t)e * D : nullptr;
            double* rp_g = st.robot_pose + 3 * (size_t)e;
            double* goal_g = st.robot_goal + 2 * (size_t)e;
            double* pa_g = st.prev_action + 2 * (size_t)e;
            double* pv_g = st.prev_pose + 3 * (size_t)e;
            int rescan = 0;
            if (uni(s.stage) == 0) {
                const int flags = uni(lds_ld(&s.flags));
                const int crash = flags & 1, discomfort = (flags >> 1) & 1;
                double rmin = 1.0e300;
                if (discomfort && !crash) {                     // env.py:563-569
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // rows written by other wavefronts
                    for (int k = lane; k < B; k += 64) {
                        double ratio = nv::discomfort_ratio((double)obs_row[(size_t)(S - 1) * B + k],
                                                            st.scan_threshold[k], st.scan_discomfort[k]);
                        rmin = ratio < rmin ? ratio : rmin;
                    }
                    rmin = wave_min_f64(rmin);
                }
                if (lane == 0) {
                    double prev_xy[2] = {pv_g[0], pv_g[1]};
                    double pose[2] = {s.rp[0], s.rp[1]};
                    double vel[2] = {pa_g[0], pa_g[1]};         // env.py:453: the PREVIOUS action
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
                        s.rp[0] = sp[0]; s.rp[1] = sp[1]; s.rp[2] = sp[2];
                        goal_g[0] = sg[0]; goal_g[1] = sg[1];
                        st.episode[e] += 1;
                        st.steps[e] = 0;
                        s.respawn = 1; rescan = 1;
                    } else if (o.crash != 0.0f) {               // env.py:707-717
                        s.rp[0] = pv_g[0]; s.rp[1] = pv_g[1]; s.rp[2] = pv_g[2];
                        rescan = 1;
                    }
                    if (rescan) {                               // scan B (env.py:718-723), then this block again
                        team_lidar_pose<RULE>(c, field, s);
                        if (s.respawn) s.n_hist = 0;
                        s.nkey = (s.noise_std > 0.0f) ? nv::noise_stream(c.seed, genv, s.step_key + 1) : 0;
                        s.done_chunks = 0; s.park_count = 0; s.park_next = 0; s.groups_done = 0;
                        s.stage = 1;
                        team_release();
                        lds_st(&s.next_chunk, 0);
                        lds_st(&s.state, kSlotScan);
                        lds_st(&s.hint, 1);
                    }
                }
                rescan = __builtin_amdgcn_readfirstlane(rescan);
            }
            if (!rescan) {
                // ---------------------------------------------------------- phase 6: pack the observation
                const bool fresh = uni(s.respawn) != 0;
                const int n_hist = uni(s.n_hist);
                if (!fresh && obs_prev) {                       // env.py:267-274: shift the stack
                    for (int j = 0; j < S - 1; ++j)
                        if (S - 1 - j <= n_hist)
                            for (int k = lane; k < B; k += 64) obs_row[(size_t)j * B + k] = obs_prev[(size_t)(j + 1) * B + k];
                }
                if (lane == 0) {
                    float* tail = obs_row + (size_t)S * B;
                    double yaw = nv::wrap_pi(s.rp[2]);          // env.py:454
                    double pxy0 = fresh ? s.rp[0] : pv_g[0];    // env.py:449-452
                    double pxy1 = fresh ? s.rp[1] : pv_g[1];
                    double v0 = fresh ? 0.0 : pa_g[0], v1 = fresh ? 0.0 : pa_g[1];
                    tail[0] = (float)pxy0; tail[1] = (float)pxy1;
                    tail[2] = (float)s.rp[0]; tail[3] = (float)s.rp[1];
                    tail[4] = (float)v0; tail[5] = (float)v1;
                    tail[6] = (float)yaw;
                    if (io.achieved_goal) { io.achieved_goal[2 * e] = (float)s.rp[0]; io.achieved_goal[2 * e + 1] = (float)s.rp[1]; }
                    if (io.desired_goal) { io.desired_goal[2 * e] = (float)goal_g[0]; io.desired_goal[2 * e + 1] = (float)goal_g[1]; }
                    // state for the next step (env.py:725-727)
                    rp_g[0] = s.rp[0]; rp_g[1] = s.rp[1]; rp_g[2] = s.rp[2];
                    if (fresh) { pa_g[0] = 0.0; pa_g[1] = 0.0; st.n_hist[e] = (S - 1 < 1) ? S - 1 : 1; }
                    else       { pa_g[0] = s.act[0]; pa_g[1] = s.act[1]; st.n_hist[e] = (n_hist + 1 < S - 1) ? n_hist + 1 : S - 1; }
                    pv_g[0] = s.rp[0]; pv_g[1] = s.rp[1]; pv_g[2] = yaw;
                    if (st.arena_cost) st.arena_cost[e] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - s.t_begin);
                    team_release();
                    lds_st(&s.state, kSlotEmpty);
                    lds_st(&s.hint, 1);
                }
            }
        }
    }
    // the last wavefront of the launch to leave resets the arena counter for the next launch
    if (lane == 0) {
        const unsigned left = atomicAdd(&g_team_counter[1], 1u);
        if (left == (unsigned)n_wavefronts_total - 1u) { g_team_counter[0] = 0u; __threadfence(); g_team_counter[1] = 0u; }
    }
}
