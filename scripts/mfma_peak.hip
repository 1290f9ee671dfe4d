// Development probe: sustained v_mfma_f32_32x32x2_f32 rate on this box with non-trivial operands
// (power/clock reality check for the rank kernel's roofline).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 37) + 0.123f, b = seed * (threadIdx.x % 29) - 0.77f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a = -a;   // keep the accumulators bounded
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 1; wpb <= 2; ++wpb)   // blocks per CU (x4 waves): 1 or 2 waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            const int blocks = 256 * wpb, iters = 20000;
            hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.001f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * 4096.0;
            printf("waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", wpb, ms, flop / ms / 1e9);
        }
    return 0;
}
