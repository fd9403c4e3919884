// Do source-VGPR bank conflicts cost issue cycles on gfx950?  (round 6: formulations of the step kernel whose probe loop has
// the SAME 60 instructions in other registers differ by 2-4 % on c2.)  Fixed physical registers through inline assembly: the
// same instruction with its sources in different banks (register number mod 4) or in the same bank, issued back to back by
// 1 .. 8 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o vgpr_banks vgpr_banks.hip && ./vgpr_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

template <int MODE>
__global__ __launch_bounds__(256) void banks(int n, float* out, unsigned long long* ticks) {
    asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n"
                 "v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n"
                 "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" ::: CLOB);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        // independent instructions (four destinations in turn), 16 per iteration
        if (MODE == 0) asm volatile(REP16("v_pk_mul_f32 v[28:29], v[12:13], v[18:19]\n") ::: CLOB);            // pairs in banks {0,1} x {2,3}
        if (MODE == 1) asm volatile(REP16("v_pk_mul_f32 v[28:29], v[12:13], v[24:25]\n") ::: CLOB);            // both pairs in banks {0,1}
        if (MODE == 2) asm volatile(REP16("v_lshl_add_u32 v28, v13, 3, v18\n") ::: CLOB);                      // banks 1, 2
        if (MODE == 3) asm volatile(REP16("v_lshl_add_u32 v28, v13, 3, v17\n") ::: CLOB);                      // banks 1, 1
        if (MODE == 4) asm volatile(REP16("v_fma_f32 v28, v13, v18, v23\n") ::: CLOB);                         // banks 1, 2, 3
        if (MODE == 5) asm volatile(REP16("v_fma_f32 v28, v13, v17, v21\n") ::: CLOB);                         // banks 1, 1, 1
        if (MODE == 6) asm volatile(REP16("v_mad_u32_u24 v28, v13, v18, v23\n") ::: CLOB);                     // banks 1, 2, 3
        if (MODE == 7) asm volatile(REP16("v_mad_u32_u24 v28, v13, v17, v21\n") ::: CLOB);                     // banks 1, 1, 1
        if (MODE == 8) asm volatile(REP16("v_pk_add_f32 v[28:29], v[26:27], v[12:13]\n") ::: CLOB);            // {2,3} + {0,1}
        if (MODE == 9) asm volatile(REP16("v_pk_add_f32 v[28:29], v[26:27], v[14:15]\n") ::: CLOB);            // {2,3} + {2,3}
        if (MODE == 10) asm volatile(REP16("v_add_f32 v28, v13, v18\n") ::: CLOB);                             // banks 1, 2
        if (MODE == 11) asm volatile(REP16("v_add_f32 v28, v13, v17\n") ::: CLOB);                             // banks 1, 1
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_mov_b32 %0, v28" : "=v"(r) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int n) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    banks<MODE><<<blocks, 256>>>(n, out, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    banks<MODE><<<blocks, 256>>>(n, out, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-52s blocks %5d  %.1f us  %.3f ns per instruction of a wavefront  (%llu ticks)\n", name, blocks, ms * 1e3, ms * 1e6 / (16.0 * n), t);
    hipFree(out); hipFree(ticks);
}

int main() {
    const int n = 4000;
    for (int blocks : {256, 2048}) {        // one 256-thread workgroup per CU (1 wavefront per SIMD) / eight (8 per SIMD)
        run<0>("v_pk_mul_f32  pairs in different banks", blocks, n);
        run<1>("v_pk_mul_f32  pairs in the same banks", blocks, n);
        run<8>("v_pk_add_f32  pairs in different banks", blocks, n);
        run<9>("v_pk_add_f32  pairs in the same banks", blocks, n);
        run<2>("v_lshl_add_u32  2 sources, different banks", blocks, n);
        run<3>("v_lshl_add_u32  2 sources, same bank", blocks, n);
        run<10>("v_add_f32 (VOP2)  different banks", blocks, n);
        run<11>("v_add_f32 (VOP2)  same bank", blocks, n);
        run<4>("v_fma_f32  3 sources, three banks", blocks, n);
        run<5>("v_fma_f32  3 sources, one bank", blocks, n);
        run<6>("v_mad_u32_u24  3 sources, three banks", blocks, n);
        run<7>("v_mad_u32_u24  3 sources, one bank", blocks, n);
    }
    return 0;
}
