// kernels_policy.hpp -- pedestrian control block with the HumanPolicy actor (row a10).
// Part of the single translation unit navsim_kernels.hip (included inside its anonymous namespace;
// not a standalone header).

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

// conv1 + ReLU, conv2 + ReLU of ONE pedestrian by its 256-thread workgroup.  x[1 + i] = network input i (x[0] = left
// padding, written here); o1 = [kConvCh][258] floats of LDS scratch; f = the pedestrian's 4096 features.  The caller has
// written x[1 .. 512]; no barrier is needed before the call.
// Round 4: conv1's 32 output channels pass through LDS kConvCh at a time -- conv2 accumulates over its input channels in
// ascending order anyway, so it takes them as they come (same chains, same order: bit-identical).  The whole conv1 output
// (33 KB) held a CU at four workgroups; 16.5 KB allow eight: 309 -> 267 us per 32 768 pedestrians (8 channels at a time:
// 1030 us -- profiles/r04_fuse/ab_conv.txt).
#ifndef NAVSIM_CONV_CH
#define NAVSIM_CONV_CH 16
#endif
constexpr int kConvCh = NAVSIM_CONV_CH;
constexpr size_t kConvLdsBytes = (size_t)kConvCh * 258 * sizeof(float);
__device__ __forceinline__ void policy_conv(float* x, float (*o1)[258],
                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                            const float* __restrict__ w2t, const float* __restrict__ b2,
                                            float* __restrict__ f) {
    const int tid = threadIdx.x;
    if (tid == 0) x[0] = 0.0f;
    if (tid < kConvCh) { o1[tid][0] = 0.0f; o1[tid][256] = 0.0f; o1[tid][257] = 0.0f; }
    __syncthreads();
    float xv[5];                                                // conv1: thread = output position
#pragma unroll
    for (int k = 0; k < 5; ++k) xv[k] = (tid < 255) ? x[2 * tid + k] : 0.0f;     // input index 2t + k - 1
    const int t = tid & 127;                                    // conv2: thread = (position, half of the channels)
    const int og = __builtin_amdgcn_readfirstlane((tid >> 7) * 16);   // wave-uniform: weights come by s_load
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    for (int c0 = 0; c0 < 32; c0 += kConvCh) {
        if (c0 > 0) __syncthreads();                            // conv2 has read the previous channels
        if (tid < 255) {
            for (int o = c0; o < c0 + kConvCh; ++o) {
                float a = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int k = 0; k < 5; ++k) a = __builtin_fmaf(w1[(o * 3 + ch) * 5 + k], xv[k], a);
                a = a + b1[o];
                o1[o - c0][1 + tid] = a > 0.0f ? a : 0.0f;
            }
        }
        __syncthreads();
        for (int c = 0; c < kConvCh; ++c) {
            const float i0 = o1[c][2 * t], i1 = o1[c][2 * t + 1], i2 = o1[c][2 * t + 2];   // index 2t + k - 1
            // w2t[c][k][o]: the 16 channels of this half are contiguous -> one scalar s_load_dwordx16 per k
            const float* ww = w2t + ((c0 + c) * 3) * 32 + og;           // wave-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc[j] = __builtin_fmaf(ww[j], i0, acc[j]);
                acc[j] = __builtin_fmaf(ww[32 + j], i1, acc[j]);
                acc[j] = __builtin_fmaf(ww[64 + j], i2, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v = acc[j] + b2[og + j];
        f[(og + j) * 128 + t] = v > 0.0f ? v : 0.0f;
    }
}
// network input of a clipped range (env.py:629-630)
__device__ __forceinline__ float policy_input(float r) {
    double v = (double)r;
    v = v < 0.0 ? 0.0 : (v > 6.0 ? 6.0 : v);
    return (float)(v / 6.0 - 0.5);
}

__global__ __launch_bounds__(256) void policy_features_kernel(const float* __restrict__ scans, int p0, int n_ped,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2t, const float* __restrict__ b2,
                                                              float* __restrict__ feat) {
    __shared__ float x[520];                 // x[1 + i] = input i, x[0] = left padding
    __shared__ float o1[kConvCh][258];       // o1[c][1 + t], zero padding at both ends
    const int tid = threadIdx.x;
    const int p = blockIdx.x;
    if (p >= n_ped) return;
    const float* scan = scans + (size_t)(p0 + p) * 512;
    for (int k = tid; k < 512; k += 256) x[1 + k] = policy_input(scan[k]);
    policy_conv(x, o1, w1, b1, w2t, b2, feat + (size_t)p * kPolFeat);
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

// robot-frame angle of a pedestrian's beam k (env.py:685-693: linspace over the half plane, the end point exact)
__device__ __forceinline__ double ped_beam_angle(const navsim_config& c, int k) {
    const int PB = c.ped_n_beams;
    const double step = (PB > 1) ? (c.ped_angle_last - c.ped_angle_min) / (double)(PB - 1) : 0.0;
    return (PB == 1) ? c.ped_angle_min : ((k == PB - 1) ? c.ped_angle_last : (double)k * step + c.ped_angle_min);
}

// W2t[k][j] = W2[j][k]: coalesced rows for the head kernel; cv2t[c][k][o] = cv2_w[o][c][k]: the conv2 weights
// of 16 adjacent output channels contiguous, so that the features kernel fetches them with wide scalar loads
// tab (optional): cos / sin of the pedestrians' robot-frame beam angles [ped_n_beams][2], the table of beam_dir_fast
__global__ __launch_bounds__(256) void policy_transpose_kernel(const float* __restrict__ w2, float* __restrict__ w2t,
                                                               const float* __restrict__ cv2, float* __restrict__ cv2t,
                                                               navsim_config c, double* __restrict__ tab) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (tab && idx < c.ped_n_beams) {
        double sn, cs;
        nv::sincos(ped_beam_angle(c, idx), sn, cs);
        tab[2 * idx] = cs; tab[2 * idx + 1] = sn;
    }
    if (idx < kPolH2 * kPolIn2) {
        int j = idx / kPolIn2, k = idx - j * kPolIn2;
        w2t[k * kPolH2 + j] = w2[idx];
    }
    if (idx < 32 * 32 * 3) {
        int o = idx / 96, c = (idx - o * 96) / 3, k = idx % 3;
        cv2t[(c * 3 + k) * 32 + o] = cv2[idx];
    }
}

// Eight pedestrians per workgroup: the 133 KB second layer is streamed once per eight of them (one pedestrian
// per workgroup made this kernel L2-bandwidth bound at 22 TB/s), thread j owns hidden unit j of all eight.
#ifndef NAVSIM_HEAD_PEDS
#define NAVSIM_HEAD_PEDS 8          // measured 8 / 16 / 32 on c3: policy 2.349 / 2.351 / 2.52 ms (profiles/r04_fuse/ab_head.txt)
#endif
constexpr int kHeadPeds = NAVSIM_HEAD_PEDS;
__global__ __launch_bounds__(128) void policy_head_kernel(navsim_config c, navsim_state st, int p0, int n_ped,
                                                          const float* __restrict__ h1, const float* __restrict__ w2t,
                                                          navsim_policy_weights w, float* __restrict__ prev_actions,
                                                          double* __restrict__ ped_cmd) {
    __shared__ float z[kHeadPeds][kPolIn2];
    __shared__ float h2[kHeadPeds][kPolH2];
    __shared__ float head[kHeadPeds][2];
    __shared__ int live_s[kHeadPeds];
    const int tid = threadIdx.x, N = c.max_peds, P = c.max_waypoints;
    const int pb = blockIdx.x * kHeadPeds;
    for (int idx = tid; idx < kHeadPeds * kPolH1; idx += 128) {
        const int s = idx / kPolH1, k = idx - s * kPolH1;
        z[s][k] = (pb + s < n_ped) ? h1[(size_t)(pb + s) * kPolH1 + k] : 0.0f;
    }
    if (tid < kHeadPeds) {                                     // one lane per pedestrian: the glue of env.py:633-645
        const int s = tid;
        int live = 0;
        z[s][256] = 0.0f; z[s][257] = 0.0f; z[s][258] = 0.0f; z[s][259] = 0.0f;
        if (pb + s < n_ped) {
            const size_t q = (size_t)(p0 + pb + s);
            const int e = (int)(q / N), i = (int)(q - (size_t)e * N);
            int n = st.n_peds[e];
            n = n > N ? N : n;
            if (i >= n) { ped_cmd[2 * q] = 0.0; ped_cmd[2 * q + 1] = 0.0; }
            else {
                live = 1;
                const double* pp = st.ped_pose + q * 3;
                const double* wp = st.ped_waypoints + (q * P) * 2;
                const double p3[3] = {pp[0], pp[1], pp[2]};
                const int head = ped_pop_waypoints(wp, st.ped_wp_head[q], st.ped_n_waypoints[q], p3);   // env.py:633-640
                st.ped_wp_head[q] = head;
                double sn, cs;
                nv::sincos(pp[2], sn, cs);                     // env.py:644-645
                double gx = wp[2 * head] - pp[0], gy = wp[2 * head + 1] - pp[1];
                z[s][256] = (float)(gx * cs + gy * sn);
                z[s][257] = (float)(-gx * sn + gy * cs);
                z[s][258] = prev_actions[2 * q];
                z[s][259] = prev_actions[2 * q + 1];
            }
        }
        live_s[s] = live;
    }
    __syncthreads();
    {
        float acc[kHeadPeds];
#pragma unroll
        for (int s = 0; s < kHeadPeds; ++s) acc[s] = 0.0f;
        for (int k = 0; k < kPolIn2; ++k) {
            const float wv = w2t[k * kPolH2 + tid];
#pragma unroll
            for (int s = 0; s < kHeadPeds; ++s) acc[s] = __builtin_fmaf(wv, z[s][k], acc[s]);
        }
        const float bj = w.fc2_b[tid];
#pragma unroll
        for (int s = 0; s < kHeadPeds; ++s) { float v = acc[s] + bj; h2[s][tid] = v > 0.0f ? v : 0.0f; }
    }
    __syncthreads();
    if (tid < 2 * kHeadPeds) {                                 // lane = (pedestrian, head)
        const int s = tid >> 1, hd = tid & 1;
        const float* aw = hd ? w.a2_w : w.a1_w;
        float acc = 0.0f;
        for (int k = 0; k < kPolH2; ++k) acc = __builtin_fmaf(aw[k], h2[s][k], acc);
        head[s][hd] = acc + (hd ? w.a2_b[0] : w.a1_b[0]);
    }
    __syncthreads();
    if (tid < kHeadPeds && live_s[tid]) {
        const int s = tid;
        const size_t q = (size_t)(p0 + pb + s);
        double x1 = (double)head[s][0], m1;                    // sigmoid, tanh in float64 on the shared exp
        if (x1 >= 0.0) m1 = 1.0 / (1.0 + nv::exp_neg(-x1));
        else { double ex = nv::exp_neg(x1); m1 = ex / (1.0 + ex); }
        double a2 = fabs((double)head[s][1]);
        double ex2 = nv::exp_neg(-2.0 * a2);
        double t2 = (1.0 - ex2) / (1.0 + ex2);
        float mean0 = (float)m1, mean1 = (float)(head[s][1] < 0.0f ? -t2 : t2);
        mean0 = mean0 < 0.0f ? 0.0f : (mean0 > 1.0f ? 1.0f : mean0);          // env.py:656-657
        mean1 = mean1 < -1.0f ? -1.0f : (mean1 > 1.0f ? 1.0f : mean1);
        prev_actions[2 * q] = mean0; prev_actions[2 * q + 1] = mean1;
        const double vp = st.ped_v_pref[q];
        ped_cmd[2 * q] = (double)mean0 * vp;                   // env.py:659-662
        ped_cmd[2 * q + 1] = (double)mean1 * vp;
    }
}
