// Development probe (round 6): what a single wave per SIMD can hide between int8 matrix instructions.  A stream of independent
// v_mfma_i32_32x32x32_i8 (inline asm, A in the accumulation file, B / C in the vector file -- the screening kernel's operand classes) with
// a controlled filler pattern in every gap; reports shader cycles per matrix instruction (s_memtime) and ns.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_filler_probe.hip -o build/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

#define MFMA(C, A, B) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(C) : "a"(A), "v"(B))

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(int* out, long long* cyc, int iters, int seed, const char* src) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    v4i32 a = {seed * (int)(threadIdx.x % 37) + 0x01020304, seed + 0x11223344, (int)threadIdx.x * 0x01010101, 0x7f80ff01};
    v4i32 b = {seed * (int)(threadIdx.x % 29) - 0x0a0b0c0d, seed ^ 0x55aa55aa, (int)threadIdx.x * 0x03050709, 0x40c0e020};
    asm volatile("" : "+a"(a));
    v16i32 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    int x0 = threadIdx.x, x1 = seed, x2 = 3, x3 = 5, x4 = 7, y = 0;
    uint32_t m = 0;
    v4i32 ld = {0, 0, 0, 0};
    const uint32_t lane16 = (threadIdx.x & 63) * 16u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            MFMA(acc[i % 6], a, b);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 1 || MODE == 2 || MODE == 3) {   // N independent integer VALU operations per gap (3 / 5 / 8)
                constexpr int N = MODE == 1 ? 3 : (MODE == 2 ? 5 : 8);
                if (N > 0) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x0) : "v"(seed));
                if (N > 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x1) : "v"(seed));
                if (N > 2) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x2) : "v"(seed));
                if (N > 3) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x3) : "v"(seed));
                if (N > 4) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x4) : "v"(seed));
                if (N > 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(seed));
                if (N > 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x1) : "v"(seed));
                if (N > 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x2) : "v"(seed));
            } else if constexpr (MODE == 4) {   // a DEPENDENT chain of 4
                asm volatile("v_sub_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(seed));
            } else if constexpr (MODE == 5) {   // VALU -> VCC -> VALU: compare + add with carry, twice
                asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_lt_i32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(y) : "v"(x0), "v"(seed), "v"(x1) : "vcc");
            } else if constexpr (MODE == 6) {   // VALU -> SGPR pair -> SALU -> VALU: two compares, s_and, cndmask
                asm volatile("v_cmp_ge_i32 s[40:41], %1, %2\n\tv_cmp_lt_i32 s[42:43], %1, %3\n\ts_and_b64 s[40:41], s[40:41], s[42:43]\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, 1, s[40:41]" : "+v"(y) : "v"(x0), "v"(seed), "v"(x1) : "s40", "s41", "s42", "s43");
            } else if constexpr (MODE == 7) {   // the screening kernel's fast path for two outputs: 2 sub, 2 alignbit, 2 xad, and, and_or (8)
                asm volatile("v_sub_u32 %2, %4, %6\n\tv_sub_u32 %3, %5, %6\n\tv_alignbit_b32 %0, %0, %2, 31\n\tv_alignbit_b32 %0, %0, %3, 31\n\t"
                             "v_xad_u32 %1, %4, -1, %7\n\tv_and_b32 %2, %1, %2\n\tv_xad_u32 %1, %5, -1, %7\n\tv_and_or_b32 %1, %1, %3, %2"
                             : "+v"(m), "+v"(y), "+v"(x2), "+v"(x3) : "v"(x0), "v"(x1), "v"(seed), "v"(x4));
            } else if constexpr (MODE == 8) {   // one ds_read_b128 per gap
                asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lds0 + lane16 + (uint32_t)(i * 1024)) : "memory");
            } else if constexpr (MODE == 9 || MODE == 10) {   // LDS-DMA pieces: one per 6 gaps (2 per 12) / one per 12
                if (i == 3 || (MODE == 9 && i == 9))
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane16 + (uint32_t)wv * 1024u), "s"(src), "s"(lds0 + (uint32_t)(wv * 1024 + (it & 7) * 6144)) : "memory", "m0");
                if (i == 11) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if constexpr (MODE == 11) {   // global_load_dwordx4 to registers + ds_write_b128, one pair per 6 gaps
                if (i == 3 || i == 9) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld) : "v"(lane16 + (uint32_t)wv * 1024u), "s"(src) : "memory");
                if (i == 5 || i == 11) asm volatile("s_waitcnt vmcnt(0)\n\tds_write_b128 %0, %1" :: "v"(lds0 + lane16), "v"(ld) : "memory");
            } else if constexpr (MODE == 12) {   // one s_barrier per 12
                if (i == 1) asm volatile("s_barrier" ::: "memory");
            } else if constexpr (MODE == 13) {   // a compare and a not-taken scalar branch per 2 gaps
                if (i & 1) asm volatile("v_cmp_gt_i32 vcc, 0, %0\n\ts_cbranch_vccnz 1f\n\ts_nop 0\n1:" :: "v"(x0 | 0x7fffffff) : "vcc");
            } else if constexpr (MODE == 14) {   // 5 SALU per gap
                asm volatile("s_add_u32 s40, s40, 1\n\ts_and_b32 s41, s40, 7\n\ts_mul_i32 s42, s41, 0x1800\n\ts_add_u32 s43, s42, 4\n\ts_min_u32 s43, s43, s40" ::: "s40", "s41", "s42", "s43");
            } else if constexpr (MODE == 15) {   // 3 v_cvt_f32_i32 + 2 v_fma_f32 (the fp32 fold of one output)
                float f0, f1, f2;
                asm volatile("v_cvt_f32_i32 %0, %3\n\tv_cvt_f32_i32 %1, %4\n\tv_cvt_f32_i32 %2, %5\n\tv_fma_f32 %1, %1, 2.0, %2\n\tv_fma_f32 %0, %0, 4.0, %1" : "=&v"(f0), "=&v"(f1), "=&v"(f2) : "v"(x0), "v"(x1), "v"(x2));
                y += (int)f0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    int s = x0 + x1 + x2 + x3 + x4 + y + (int)m + (ld.x + ld.y + ld.z + ld.w);
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int* out, long long* cyc, const char* src) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int blocks = 256, iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 65536, 0, out, cyc, 50, 3, src);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 65536, 0, out, cyc, iters, 3, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 12;
    printf("mode %2d  %-78s ticks/MFMA %6.1f   ns/MFMA %6.2f\n", MODE, name, (double)c / n, ms * 1e6 / n);
}
int main() {
    int* out; hipMalloc(&out, 4096 * 256 * 4);
    long long* cyc; hipMalloc(&cyc, 8);
    char* src; hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
    run<0>("bare", out, cyc, src);
    run<1>("3 independent v_sub per gap", out, cyc, src);
    run<2>("5 independent VALU per gap", out, cyc, src);
    run<3>("8 independent VALU per gap", out, cyc, src);
    run<4>("dependent chain of 4 VALU per gap", out, cyc, src);
    run<5>("v_cmp -> vcc -> v_addc, twice (4) per gap", out, cyc, src);
    run<6>("2 v_cmp -> SGPR, s_and, s_nop, v_cndmask (5) per gap", out, cyc, src);
    run<7>("screening fast path, two outputs (8 VALU) per gap", out, cyc, src);
    run<8>("1 ds_read_b128 per gap", out, cyc, src);
    run<9>("LDS-DMA dwordx4 piece: 2 per 12 gaps", out, cyc, src);
    run<10>("LDS-DMA dwordx4 piece: 1 per 12 gaps", out, cyc, src);
    run<11>("global_load_dwordx4 + ds_write_b128: 2 pairs per 12 gaps", out, cyc, src);
    run<12>("s_barrier: 1 per 12 gaps", out, cyc, src);
    run<13>("v_cmp + not-taken s_cbranch per 2 gaps", out, cyc, src);
    run<14>("5 SALU per gap", out, cyc, src);
    run<15>("3 v_cvt_f32_i32 + 2 v_fma_f32 per gap", out, cyc, src);
    return 0;
}
