// Development probe (round 6): sustained rate of the int8 matrix instructions the screening kernel is built on, on this box, with
// non-trivial operands: v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8, one and two waves per SIMD, shader cycles per instruction
// (s_memtime) and TOPS.  hipcc --offload-arch=gfx950 -O3 scripts/mfma_i8_peak.hip -o /tmp/mfma_i8_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));
template <int NACC, bool BIG>
__global__ __launch_bounds__(256) void k(int* out, long long* cyc, int iters, int seed) {
    v4i32 a = {seed * (int)(threadIdx.x % 37) + 0x01020304, seed + 0x11223344, (int)threadIdx.x * 0x01010101, 0x7f80ff01};
    v4i32 b = {seed * (int)(threadIdx.x % 29) - 0x0a0b0c0d, seed ^ 0x55aa55aa, (int)threadIdx.x * 0x03050709, 0x40c0e020};
    v16i32 acc[NACC]; v4i32 acs[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0; acs[i] = v4i32{0, 0, 0, 0}; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (BIG) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            else acs[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acs[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acs[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool BIG>
void run(const char* name, int* out, long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 1; wpb <= 2; ++wpb) {
        const int blocks = 256 * wpb, iters = 20000;
        hipLaunchKernelGGL((k<NACC, BIG>), dim3(blocks), dim3(256), 0, 0, out, cyc, 100, 3);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, BIG>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * NACC, ops = (double)blocks * 4 * n * (BIG ? 65536.0 : 32768.0);
        printf("%s acc=%d waves/SIMD=%d  %.3f ms  %.0f TOPS  s_memtime ticks per instruction per wave %.1f  (ns per instr per SIMD %.2f)\n", name, NACC, wpb, ms, ops / ms / 1e9,
               (double)c / n, ms * 1e6 / (n * wpb));
    }
}
int main() {
    int* out; hipMalloc(&out, 4096 * 256 * 4);
    long long* cyc; hipMalloc(&cyc, 8);
    run<6, true>("i32_32x32x32_i8", out, cyc);
    run<4, true>("i32_32x32x32_i8", out, cyc);
    run<6, false>("i32_16x16x64_i8", out, cyc);
    run<12, false>("i32_16x16x64_i8", out, cyc);
    return 0;
}
