// How fast does ONE wavefront issue dependent / independent integer and float instructions when the chip is nearly idle?
// (round 3: the reset kernels of navsim_regen take ~45 cycles per instruction -- clock, issue or fetch?)
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void chain(int n, int* out, unsigned long long* ticks) {
    int a = threadIdx.x + 1, b = a * 3, c = a + 7, d = a ^ 5;
    float f = (float)a, g = f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { a = a * 3 + b; a = a * 5 + c; a = a * 7 + d; a = a * 9 + b; }                 // dependent v_mad_u32 / mul_lo
        if (MODE == 1) { a = a * 3 + 1; b = b * 5 + 1; c = c * 7 + 1; d = d * 9 + 1; }                 // four independent chains
        if (MODE == 2) { f = f * 1.0001f + g; f = f * 1.0002f + g; f = f * 1.0003f + g; f = f * 1.0004f + g; }   // dependent float
        if (MODE == 3) { a = (a + b) ^ c; a = (a + c) ^ d; a = (a + d) ^ b; a = (a + b) ^ d; }          // dependent add / xor
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (int)f;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int n) {
    int* out; unsigned long long* ticks;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<MODE><<<blocks, 256>>>(n, out, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<MODE><<<blocks, 256>>>(n, out, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double instr = 8.0 * n;          // 4 statements x (mul + add) roughly
    printf("%-28s blocks %5d  n %6d  %.1f us  %.2f ns per statement-pair-instruction  s_memtime ticks %llu (%.3f ticks/ns)\n",
           name, blocks, n, ms * 1e3, ms * 1e6 / instr, t, t / (ms * 1e6));
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int blocks : {1, 5, 640, 4096}) {
        run<0>("dependent int mul-add", blocks, 4000);
        run<1>("4 independent int chains", blocks, 4000);
        run<2>("dependent float fma", blocks, 4000);
        run<3>("dependent add / xor", blocks, 4000);
    }
    return 0;
}
