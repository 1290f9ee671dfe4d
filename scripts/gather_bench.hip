// Development probe: ceiling of the random-row gather the train kernels live on.  Each wave reads ROWS random rows of
// K floats (16-byte loads, PF rows in flight) from a table of N rows; reports TB/s.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int PF>
__global__ __launch_bounds__(256) void gather(const float* tab, const int* ids, int rows_per_wave, int K, float* out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int* my = ids + wave * rows_per_wave;
    const int nq = K / 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r0 = 0; r0 < rows_per_wave; r0 += PF) {
        float4 v[PF][2];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const float* row = tab + (long)__builtin_amdgcn_readfirstlane(my[r0 + f]) * K;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int q = lane + 64 * c;
                v[f][c] = q < nq ? *reinterpret_cast<const float4*>(row + q * 4) : make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < PF; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c) { acc.x += v[f][c].x; acc.y += v[f][c].y; acc.z += v[f][c].z; acc.w += v[f][c].w; }
    }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}
int main() {
    const int K = 400, waves = 10000, rows = 24;
    for (int N : {14505, 123182, 2000000}) {
        float* tab; hipMalloc(&tab, (size_t)N * K * 4); hipMemset(tab, 0, (size_t)N * K * 4);
        std::vector<int> h((size_t)waves * rows);
        srand(1);
        for (auto& x : h) x = (int)(((long)rand() * 32768 + rand()) % N);
        int* ids; hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        float* out; hipMalloc(&out, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](int pf) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (pf == 1) hipLaunchKernelGGL(gather<1>, dim3(waves / 4), dim3(256), 0, 0, tab, ids, rows, K, out);
                if (pf == 4) hipLaunchKernelGGL(gather<4>, dim3(waves / 4), dim3(256), 0, 0, tab, ids, rows, K, out);
                if (pf == 8) hipLaunchKernelGGL(gather<8>, dim3(waves / 4), dim3(256), 0, 0, tab, ids, rows, K, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("N=%8d PF=%d : %.1f us  %.2f TB/s\n", N, pf, ms * 1e3, (double)waves * rows * K * 4 / ms / 1e9);
        };
        run(1); run(4); run(8);
        hipFree(tab); hipFree(ids); hipFree(out);
    }
    return 0;
}
