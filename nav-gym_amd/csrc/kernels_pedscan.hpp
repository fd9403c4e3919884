// kernels_pedscan.hpp -- pedestrian scans, beam table, device-math test hook, gather microbenchmark.
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).

// ============================================================================================
// env.py:685-693: the 512-beam half-plane scan of every pedestrian (what the reference feeds to
// HumanPolicy).  One workgroup per (pedestrian, arena): rectangles of the other agents in LDS,
// march from the pedestrian's integer cell, bearing-culled polygon merge, clip to 6 m.
// ============================================================================================
template <typename Field, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ped_scan_kernel(navsim_config c, navsim_state st, float* __restrict__ out) {
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
