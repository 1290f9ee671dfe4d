// Development probe (round 4): can the eight private L2s serve the train step's random row gathers?
// A: every wave gathers whole 1.6 KB rows of random ids from the 23 MB table (what F does today: ~0 % L2 hit rate, fabric-bound).
// B: the table is cut into 8 COLUMN slices; block b (observed on XCD b % 8) only ever touches slice b % 8 -- 14 505 rows x 208 B
//    = 3.0 MB, which fits that XCD's 4 MB L2 -- and gathers 4 row slices per wave instruction (16 lanes x 16 B each).
// C: as B, but the slice of a block is (b / 8) % 8: every XCD touches every slice (the control: same instruction stream, no residency).
// Same bytes in all three.  hipcc --offload-arch=gfx950 -O3 scripts/xcd_slice_bench.hip -o scripts/xcd_slice_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int PF>
__global__ __launch_bounds__(256) void gather_rows(const float* tab, const int* ids, int rows_per_wave, int K, float* out) {
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

// sliced layout: tab[slice][row][SQ quads]; a wave = one (positive, slice) item: rows_per_item row slices, 4 per load instruction
template <int PF, bool SCRAMBLE>
__global__ __launch_bounds__(256) void gather_slices(const float* tab, const int* ids, int rows_per_item, int N, int SQ, long n_items, float* out) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    // blocks 8 g + x of a group of 8 all work on positives [4 g, 4 g + 4) -- one slice each
    const long g = blockIdx.x >> 3;
    const int x = SCRAMBLE ? (int)(g & 7) : (int)(blockIdx.x & 7);
    const long item = g * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int* my = ids + item * rows_per_item;
    const float* base = tab + (long)x * N * SQ * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r0 = 0; r0 < rows_per_item; r0 += 4 * PF) {
        float4 v[PF];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int r = min(r0 + 4 * f + sub, rows_per_item - 1);
            const int id = my[r];
            v[f] = l16 < SQ ? *reinterpret_cast<const float4*>(base + ((long)id * SQ + l16) * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int f = 0; f < PF; ++f) { acc.x += v[f].x; acc.y += v[f].y; acc.z += v[f].z; acc.w += v[f].w; }
    }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}

int main() {
    const int K = 400, items = 10000, rows = 24, SQ = 13;
    for (int N : {14505, 40943, 123182}) {
        float* tab; hipMalloc(&tab, (size_t)N * 8 * SQ * 16); hipMemset(tab, 0, (size_t)N * 8 * SQ * 16);
        std::vector<int> h((size_t)items * rows);
        srand(1);
        for (auto& x : h) x = (int)(((long)rand() * 32768 + rand()) % N);
        int* ids; hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        float* out; hipMalloc(&out, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto time = [&](const char* name, auto launch, double bytes) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("N=%7d %-34s %7.1f us  %6.2f TB/s\n", N, name, best * 1e3, bytes / best / 1e9);
        };
        const double bytes_rows = (double)items * rows * K * 4, bytes_sl = (double)items * rows * 8 * SQ * 16;
        time("A whole rows, PF=6", [&] { hipLaunchKernelGGL(gather_rows<6>, dim3(items / 4), dim3(256), 0, 0, tab, ids, rows, K, out); }, bytes_rows);
        time("A whole rows, PF=8", [&] { hipLaunchKernelGGL(gather_rows<8>, dim3(items / 4), dim3(256), 0, 0, tab, ids, rows, K, out); }, bytes_rows);
        const unsigned gs = (unsigned)(items / 4 * 8);
        time("B slice = block % 8, PF=3", [&] { hipLaunchKernelGGL((gather_slices<3, false>), dim3(gs), dim3(256), 0, 0, tab, ids, rows, N, SQ, (long)items, out); }, bytes_sl);
        time("B slice = block % 8, PF=6", [&] { hipLaunchKernelGGL((gather_slices<6, false>), dim3(gs), dim3(256), 0, 0, tab, ids, rows, N, SQ, (long)items, out); }, bytes_sl);
        time("C slice = (block / 8) % 8, PF=6", [&] { hipLaunchKernelGGL((gather_slices<6, true>), dim3(gs), dim3(256), 0, 0, tab, ids, rows, N, SQ, (long)items, out); }, bytes_sl);
        hipFree(tab); hipFree(ids); hipFree(out);
    }
    return 0;
}
