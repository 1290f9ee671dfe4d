// Development microbenchmark: row-scatter patterns on MI355X (what bounds the gradient scatter?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int K = 400;

template <int MODE>
__global__ __launch_bounds__(256) void scatter(float* G, const int* rows, int nrows, int N) {
    const int lane = threadIdx.x & 63;
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nrows) return;
    int row = rows[w];
    if (MODE == 5) {  // XCD-local rows: remap the row into the slice owned by this XCD
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        const int per = N / 8;
        row = (row % per) + xcc * per;
    }
    float* g = G + (size_t)row * K;
    const float v = 1.0f + lane;
    if (MODE == 1) {
        if (lane < 50) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 4; ++u) unsafeAtomicAdd(g + h * 200 + 4 * lane + u, v);
        }
    } else if (MODE == 2 || MODE == 5) {
#pragma unroll
        for (int t = 0; t < 7; ++t) { const int idx = t * 64 + lane; if (idx < K) unsafeAtomicAdd(g + idx, v); }
    } else if (MODE == 3) {
#pragma unroll
        for (int t = 0; t < 7; ++t) { const int idx = t * 64 + lane; if (idx < K) g[idx] = v; }
    } else if (MODE == 4) {
#pragma unroll
        for (int t = 0; t < 7; ++t) { const int idx = t * 64 + lane; if (idx < K) g[idx] += v; }
    } else if (MODE == 6) {   // float4 non-atomic RMW
        if (lane < 50) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { float4* p = (float4*)(g + h * 200 + 4 * lane); float4 t = *p; t.x += v; t.y += v; t.z += v; t.w += v; *p = t; }
        }
    } else if (MODE == 7) {   // f64 atomics, contiguous (200 doubles per row)
        double* gd = (double*)g;
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int idx = t * 64 + lane; if (idx < 200) unsafeAtomicAdd(gd + idx, (double)v); }
    }
}

template <int MODE>
float run(float* G, const int* rows, int nrows, int N, const char* name) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(scatter<MODE>, dim3((nrows + 3) / 4), dim3(256), 0, 0, G, rows, nrows, N);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scatter<MODE>, dim3((nrows + 3) / 4), dim3(256), 0, 0, G, rows, nrows, N);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-44s N=%8d rows=%d : %8.3f ms  %7.1f G elem/s  %7.1f GB/s\n", name, N, nrows, ms, nrows * (double)K / ms / 1e6, nrows * (double)K * 4 / ms / 1e6);
    return ms;
}

static void flavors() {
    const int nrows = 230000, N = 14505;
    std::vector<int> h(nrows); srand(1);
    for (auto& r : h) r = (int)(((unsigned)rand() * 2654435761u) % (unsigned)N);
    int* rows; CK(hipMalloc(&rows, nrows * 4)); CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
    struct { const char* name; unsigned flag; } fl[] = {{"hipDeviceMallocDefault", hipDeviceMallocDefault},
        {"hipDeviceMallocFinegrained", hipDeviceMallocFinegrained}, {"hipDeviceMallocUncached", hipDeviceMallocUncached}};
    for (auto& f : fl) {
        float* G;
        if (hipExtMallocWithFlags((void**)&G, (size_t)N * K * 4, f.flag) != hipSuccess) { printf("%s: alloc failed\n", f.name); continue; }
        CK(hipMemset(G, 0, (size_t)N * K * 4));
        printf("-- %s\n", f.name);
        run<2>(G, rows, nrows, N, "2 f32 atomics, lane-contiguous");
        run<7>(G, rows, nrows, N, "7 f64 atomics, contiguous");
        run<4>(G, rows, nrows, N, "4 non-atomic RMW dword, contiguous");
        CK(hipFree(G));
    }
    CK(hipFree(rows));
}

int main() {
    flavors();
    const int nrows = 230000;
    for (int N : {14505}) {
        float* G; int* rows;
        CK(hipMalloc(&G, (size_t)N * K * 4)); CK(hipMemset(G, 0, (size_t)N * K * 4));
        std::vector<int> h(nrows); srand(1);
        for (auto& r : h) r = (int)(((unsigned)rand() * 2654435761u) % (unsigned)N);
        CK(hipMalloc(&rows, nrows * 4)); CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
        run<1>(G, rows, nrows, N, "1 f32 atomics, 16B-stride quads (current)");
        run<2>(G, rows, nrows, N, "2 f32 atomics, lane-contiguous");
        run<5>(G, rows, nrows, N, "5 f32 atomics, contiguous, XCD-local rows");
        run<7>(G, rows, nrows, N, "7 f64 atomics, contiguous");
        run<3>(G, rows, nrows, N, "3 plain stores, contiguous");
        run<4>(G, rows, nrows, N, "4 non-atomic RMW dword, contiguous");
        run<6>(G, rows, nrows, N, "6 non-atomic RMW float4");
        CK(hipFree(G)); CK(hipFree(rows));
    }
    return 0;
}
