// Issue cost of single vector instructions on gfx950 with the SIMD full (8 wavefronts per SIMD) -- what a probe round of the
// march is made of, and what it could be made of instead.  Fixed physical registers (sources in distinct banks), independent
// instructions back to back; ns per instruction of one wavefront, and the same in cycles of the SIMD (x clock / 8 waves).
//   hipcc --offload-arch=gfx950 -O3 -w -o inst_rate inst_rate.hip && ./inst_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "vcc", "s20", "s21"

#define KERNEL(NAME, TEXT) \
__global__ __launch_bounds__(256) void NAME(int n, float* out) { \
    asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 2.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 2.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n" \
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 4.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n" \
                 "v_mov_b32 v22, 1.0\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n" \
                 "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n s_mov_b64 s[20:21], -1\n" ::: CLOB); \
    for (int i = 0; i < n; ++i) asm volatile(REP16(TEXT "\n") ::: CLOB); \
    float r; \
    asm volatile("v_mov_b32 %0, v28" : "=v"(r) :: CLOB); \
    out[blockIdx.x * 256 + threadIdx.x] = r; \
}

KERNEL(k_add_f32, "v_add_f32 v28, v13, v18")
KERNEL(k_sub_f32, "v_sub_f32 v28, v13, v18")
KERNEL(k_mul_f32, "v_mul_f32 v28, v13, v18")
KERNEL(k_max_f32, "v_max_f32 v28, v13, v18")
KERNEL(k_fma_f32, "v_fma_f32 v28, v13, v18, v23")
KERNEL(k_fmac_f32, "v_fmac_f32 v28, v13, v18")
KERNEL(k_med3_f32, "v_med3_f32 v28, v13, v18, v23")
KERNEL(k_max3_f32, "v_max3_f32 v28, v13, v18, v23")
KERNEL(k_trunc_f32, "v_trunc_f32 v28, v13")
KERNEL(k_floor_f32, "v_floor_f32 v28, v13")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 v28, v13")
KERNEL(k_cvt_u32_f32, "v_cvt_u32_f32 v28, v13")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 v28, v13")
KERNEL(k_rsq_f32, "v_rsq_f32 v28, v13")
KERNEL(k_sqrt_f32, "v_sqrt_f32 v28, v13")
KERNEL(k_rcp_f32, "v_rcp_f32 v28, v13")
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 v[28:29], v[12:13], v[18:19]")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 v[28:29], v[12:13], v[18:19], v[22:23]")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 v28, v13, v18")
KERNEL(k_pk_max_i16, "v_pk_max_i16 v28, v13, v18")
KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 v28, v13, v18, 0")
KERNEL(k_min_i32, "v_min_i32 v28, v13, v18")
KERNEL(k_add_u32, "v_add_u32 v28, v13, v18")
KERNEL(k_lshrrev_b32, "v_lshrrev_b32 v28, 3, v13")
KERNEL(k_and_b32, "v_and_b32 v28, 0xff, v13")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 v28, v13, 3, v18")
KERNEL(k_lshl_or_b32, "v_lshl_or_b32 v28, v13, 16, v18")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 v28, v13, v18, v23")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 v28, v13, v18")
KERNEL(k_cndmask, "v_cndmask_b32_e64 v28, v13, v18, s[20:21]")
KERNEL(k_cmp_lt_f32, "v_cmp_lt_f32_e64 s[20:21], v13, v18")
KERNEL(k_cmp_eq_u32, "v_cmp_eq_u32_e64 s[20:21], v13, v18")
KERNEL(k_mov_b32, "v_mov_b32 v28, v13")
KERNEL(k_add_f64, "v_add_f64 v[28:29], v[12:13], v[18:19]")
KERNEL(k_mul_f64, "v_mul_f64 v[28:29], v[12:13], v[18:19]")
KERNEL(k_fma_f64, "v_fma_f64 v[28:29], v[12:13], v[18:19], v[22:23]")
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 v[28:29], v13")
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 v28, v[12:13]")
KERNEL(k_log_f32, "v_log_f32 v28, v13")
KERNEL(k_cos_f32, "v_cos_f32 v28, v13")
KERNEL(k_bfe_u32, "v_bfe_u32 v28, v13, 3, 8")
KERNEL(k_xor_b32, "v_xor_b32 v28, v13, v18")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 v28, v13, v18")
KERNEL(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 v28, v13")
KERNEL(k_mul_f32_same, "v_mul_f32 v28, v13, v13")
KERNEL(k_fma_f32_2same, "v_fma_f32 v28, v13, v13, v18")
KERNEL(k_fma_f32_bank2, "v_fma_f32 v28, v13, v17, v18")
KERNEL(k_med3_f32_bank2, "v_med3_f32 v28, v13, v17, v18")

typedef void (*kern_t)(int, float*);
void run(const char* name, kern_t k, int blocks, int n, double clock_ghz) {
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<blocks, 256>>>(n, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<blocks, 256>>>(n, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (16.0 * n);
    printf("%-44s %7.3f ns per wave-instruction  = %5.2f SIMD cycles at %.1f GHz\n", name, ns, ns * clock_ghz / 8.0, clock_ghz);
    (void)hipFree(out);
}

int main() {
    const int n = 3000, blocks = 2048;          // eight 256-thread workgroups per CU: 8 wavefronts per SIMD
    int khz = 2400000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
#define R(NAME) run(#NAME, NAME, blocks, n, ghz)
    R(k_add_f32); R(k_sub_f32); R(k_mul_f32); R(k_max_f32); R(k_fma_f32); R(k_fmac_f32); R(k_med3_f32); R(k_max3_f32);
    R(k_mul_f32_same); R(k_fma_f32_2same); R(k_fma_f32_bank2); R(k_med3_f32_bank2);
    R(k_trunc_f32); R(k_floor_f32); R(k_cvt_i32_f32); R(k_cvt_u32_f32); R(k_cvt_f32_i32); R(k_cvt_f32_ubyte0);
    R(k_rsq_f32); R(k_sqrt_f32); R(k_rcp_f32); R(k_log_f32); R(k_cos_f32);
    R(k_pk_mul_f32); R(k_pk_fma_f32); R(k_pk_sub_i16); R(k_pk_max_i16); R(k_dot2_i32_i16);
    R(k_min_i32); R(k_add_u32); R(k_lshrrev_b32); R(k_and_b32); R(k_xor_b32); R(k_bfe_u32); R(k_lshl_add_u32); R(k_lshl_or_b32);
    R(k_mad_u32_u24); R(k_mul_u32_u24); R(k_mul_lo_u32); R(k_cndmask); R(k_cmp_lt_f32); R(k_cmp_eq_u32); R(k_mov_b32);
    R(k_add_f64); R(k_mul_f64); R(k_fma_f64); R(k_cvt_f64_f32); R(k_cvt_f32_f64);
    return 0;
}
