// Round 6: the screening kernel restructured around ONE wave per SIMD with the QUERY operand resident in registers.
// Included by kge_rank.hip behind kge_rank_screen.h, whose limbs / thresholds / recheck / merge kernels, ScreenArgs and error bound it
// shares: the matrix work is rank_screen_kernel_v1's (six limb products in three exact int32 accumulator levels per output), the
// decisions are the same ones taken in integer arithmetic against a tile-wide bound -- the counts are the same integers.
//
// Why.  rank_screen_kernel_v1 (two 256-thread workgroups per CU, 233 registers) re-reads a wave's query fragments from L2 for every
// entity tile -- 12 KB of the 18 KB a workgroup-stage moves -- and a wave-stage lasts ~1 900 cycles for 384 cycles of matrix work
// (MfmaUtil 0.38-0.39 for three rounds; every ablation of one instruction class, the register-staged LDS form of round 5 and the LDS-DMA
// ring of this round left it there).  At the BASELINE widths (ComplEx k = 200, DistMult k = 400: U = 400 int8 units = S = 13 slabs of
// 32) the three limbs of a wave's 32 query rows are 13 x 3 x 4 = 156 registers: with ONE wave per SIMD gfx950's unified file gives a
// wave 512, so they are loaded ONCE per workgroup into the accumulation half of the file and stay.  What a stage then needs is the
// entity slab only (6 KB per workgroup-stage instead of 18), which arrives by LDS-DMA into a ring of eight 6 KB buffers, six positions
// ahead (counted vmcnt; one raw s_barrier per stage; 1.5 DMA instructions per wave and stage), and whose fragments are read from LDS
// one stage ahead of the matrix instructions that consume them.  One wave has nobody to fill its gaps, so the kernel is a software
// pipeline written out: tile t's 156 matrix instructions (one inline-assembly statement each, operand classes pinned) with tile t - 1's
// epilogue sliced between them, and an epilogue cut down to what one wave can hide (the step table of round 6):
// candidates on ONE scale per tile of 64 (rank_limbs_tile_kernel), integer thresholds per (query row, tile) computed once by the
// thread that owns the row, three integer operations to fold an output's accumulator levels, and four subtractions whose SIGN BITS are
// gathered into bit masks -- "not greater" and "equal" per row, "undecided" per tile; no VALU -> SGPR -> VALU round trip and no
// compare + branch anywhere in the chain -- the rare undecided outputs go to the exact recheck as before.
// The tile loop is unrolled over the S slabs (the query registers are indexed statically): one instantiation per S, and written out
// for two consecutive tiles (with an odd S everything that alternates from position to position is a constant of the copy).
//
// What the second half of round 6 measured about ONE wave per SIMD (profiles/r06x*, r06y*):
//   * SQ_VALU_MFMA_COEXEC_CYCLES says how much of the vector work runs in a matrix instruction's shadow: 0.11 of the pipe's busy
//     cycles with the slices in blocks of ten operations (only the first three or four behind a matrix instruction overlap it: the
//     next matrix instruction waits behind the rest), 0.19 with <= 3 per slot;
//   * a v_cmp + s_cbranch per row cost ~75 cycles each (16 per tile): gone, 814 -> 779 us; a vector-memory load in the loop (the tile
//     metas) made the compiler drain the DMA ring with vmcnt(0) once per tile: staged in LDS;
//   * 250 scalar instructions and 40 s_waitcnt fewer per tile (a ring addressed by immediates) bought nothing: scalar work is free;
//   * THE TRAP: gfx90a+ wants two wait states between a VALU write of a register and a matrix instruction that reads it.  The
//     compiler pads its own matrix instructions; it cannot see into these statements.  When a rearrangement made it park a tile's
//     first fragments in the accumulation file and copy them back (v_accvgpr_read) right in front of their first use, that matrix
//     instruction read stale registers -- wrong, run-to-run different ranks in two otherwise correct variants.  The stage-0
//     statements carry their own s_nop 1, and scripts/check_mfma_hazards.py (tests/test_build_hazards.py) scans the generated code.
#pragma once

#ifndef SCRR_ABLATE
#define SCRR_ABLATE 0   // development (scripts/build_variant.sh with EXTRA=-DSCRR_ABLATE=n): 1 no epilogue slices, 2 no matrix instructions, 4 no stage barrier, 8 no fold of the accumulators, 16 no DMA, 32 no fragment reads, 64 no hand-over of the undecided outputs -- wrong counts, timing only
#endif

namespace kge {

#ifndef SCRR_DEPTH
#define SCRR_DEPTH 6
#endif
constexpr int SCRR_D = SCRR_DEPTH;                 // positions in flight ahead of the one being multiplied (even)
constexpr int SCRR_NB = SCRR_DEPTH <= 6 ? 8 : 16;  // ring buffers (>= D + 2; a power of two)
constexpr int SCRR_PIECE = 1024;                   // bytes per DMA instruction: 64 lanes x 16
constexpr int SCRR_STAGE = 6 * SCRR_PIECE;         // entity blocks 0, 1 x 3 limbs, as they lie in memory
constexpr int SCRR_NONE = -(1 << 30);               // "no output here" in the folded accumulators' units (|g| < 2^29.1 for S <= 13; thresholds live in (-2^30, 2^30])
constexpr int SCRR_PEND = 256;                     // undecided pairs a wave parks in LDS before they go to the list
constexpr int SCRR_TMCAP = 1024;                   // tiles per block at most (their metas are staged in LDS; run_screen's schedule holds the run below it)
constexpr int SCRR_UB = 16;                        // tiles whose undecided marks are handed over together
constexpr size_t SCRR_LDS_UND = (size_t)SCRR_NB * SCRR_STAGE + 2 * 128 * 16 + 4 * (size_t)SCRR_PEND * 8 + (size_t)SCRR_TMCAP * 16;   // [SCRR_UB][256] uint32: a thread's marks of the last tiles
constexpr size_t SCRR_LDS_BYTES = SCRR_LDS_UND + (size_t)SCRR_UB * 256 * 4;   // 94 208: ring, thresholds, parked pairs, tile metas, marks

// One LDS-DMA instruction, scalar base + per-lane 32-bit offset: 64 lanes x 16 bytes to LDS bytes [lds, lds + 1 024) (M0 = the
// wave-uniform LDS byte address; the hardware adds lane x 16).  Inline assembly on purpose (as in scripts/experiments/rank_screen_kernel_g_lds_dma_r06.h): the compiler must
// not know an LDS write is pending, the pieces are counted by hand.
__device__ __forceinline__ void scrr_dma16(const char* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

template <int N, typename F>
__device__ __forceinline__ void scrr_static_for(F&& f) {
    if constexpr (N > 0) {
        scrr_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int S>
__global__ __launch_bounds__(SCR_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void rank_screen_kernel_r(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];   // (the ONE LDS object of the kernel)
    int4* const thr_s = reinterpret_cast<int4*>(smem_scr + (size_t)SCRR_NB * SCRR_STAGE);   // [2][128]: (query row, tile)'s integer thresholds, by tile parity

    if (a.wild_mode == 2 && screen_wild(a.b.counter, a.m)) return;   // (rows far below their tile's scale: the per-row-scale path takes the call)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wq = wv * 32;
    int bx, by;   // XCD-aware work order, as rank_screen_kernel_v1
    {
        const int xcd = blockIdx.x & 7;
        const int64_t i = blockIdx.x >> 3;
        const int qlo = (int)(((int64_t)a.qtiles * xcd) / 8), qhi = (int)(((int64_t)a.qtiles * (xcd + 1)) / 8);
        const int nq = qhi - qlo;
        if (i >= (int64_t)nq * a.splits) return;
        const int full = nq / 8;
        const int64_t per_group = (int64_t)8 * a.splits;
        if (i < full * per_group) {
            const int64_t r = i % per_group;
            bx = qlo + (int)(i / per_group) * 8 + (int)(r & 7);
            by = (int)(r >> 3);
        } else {
            const int rem = nq - full * 8;
            const int64_t r = i - full * per_group;
            bx = qlo + full * 8 + (int)(r % rem);
            by = (int)(r / rem);
        }
    }
    const int64_t q0 = (int64_t)bx * SCR_Q;
    const int64_t e_begin = (int64_t)by * a.ent_per_block;
    const int64_t e_end = min(a.m, e_begin + a.ent_per_block);
    const int64_t ntile = (e_end - e_begin + SCR_ET - 1) / SCR_ET;

    // ---- this THREAD's query row (tid & 127) and pair of thresholds (waves 0, 1: greater / smaller; waves 2, 3: the two ends of
    // "equal"), for the whole block: per tile it turns them into integer thresholds in the accumulators' units (below).
    // {c gamma |q|_2,  c A,  c (|q|_1 / 2 + drop A)} inflated by c = 1 + 2^-10 as in rank_screen_kernel_v1; 1 / (2^24 A): a power of two.
    float rq_y, rq_z, rq_w, rq_iA, thrA, thrB;
    {
        const int row = tid & 127;
        const bool okq = q0 + row < a.n;
        const float4 m4 = a.b.qm[okq ? q0 + row : a.n - 1];
        const float c = 1.f + 0x1p-10f;
        rq_y = m4.y * c; rq_z = m4.x * c; rq_w = fmaf(a.drop, m4.x, m4.z) * c;
        rq_iA = 0x1p-24f / m4.x;
        // the relative part of the epilogue's roundings sits in the thresholds (2^-20 |T|, see rank_screen_kernel_v1); non-finite
        // thresholds (nothing can be greater / smaller) stay
        const float2 t2 = a.b.qt[okq ? q0 + row : a.n - 1];
        const float s1 = isfinite(t2.x) ? 0x1p-20f * fabsf(t2.x) : 0.f, s2 = isfinite(t2.y) ? 0x1p-20f * fabsf(t2.y) : 0.f;
        thrA = (tid < 128) ? t2.y + s2 : t2.x + s1;   // G  (greater: S - E >= G)      | EL (equal: S - E >= EL ...
        thrB = (tid < 128) ? t2.x - s1 : t2.y - s2;   // L  (smaller: S + E <  L)      | EH  ... and S + E < EH)
        // (both parities start as "nothing decided": iteration 0 runs the slices on an empty tile)
        thr_s[row] = make_int4(1 << 30, -1073741760, 1 << 30, -1073741760);
        thr_s[128 + row] = make_int4(1 << 30, -1073741760, 1 << 30, -1073741760);
    }
    uint32_t rowmask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rowmask |= (q0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < a.n) ? (3u << (30 - 2 * r)) : 0u;   // (bit 31 - (2 r + ni): the order the slices shift the marks in)

    // ---- this wave's query fragments, all S slabs x 3 limbs: 39 coalesced 1 KB reads, once (rows beyond n: the stale tail of the last
    // block -- finite integers; their outputs are masked)
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const uint32_t blk_stride = (uint32_t)S * SCR_BLK_SLAB;   // bytes between consecutive 32-row blocks
    const uint32_t lane16 = (uint32_t)lane * 16u;
    v4i32 qf[S][3];
    {
        const char* const qsrc = reinterpret_cast<const char*>(a.b.qlimbs) + ((q0 + 32 * wv_s) >> 5) * (int64_t)blk_stride;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int lb = 0; lb < 3; ++lb) {
                const uint4 u = *reinterpret_cast<const uint4*>(qsrc + ((size_t)s * SCR_BLK_SLAB + 1024u * lb + lane16));
                qf[s][lb] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
            }
    }
    float4* const tm_s = reinterpret_cast<float4*>(smem_scr + (size_t)SCRR_NB * SCRR_STAGE + 2 * 128 * 16 + 4 * (size_t)SCRR_PEND * 8);
    {   // {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t} of this block's tiles, staged once (ntile <= SCRR_TMCAP: run_screen's schedule)
        const float4* const src = a.b.tm + (e_begin >> 6);
        for (int i = tid; i < (int)ntile; i += SCR_THREADS) tm_s[i] = src[i];
    }
    __syncthreads();   // (the last ordinary loads of the kernel are behind this barrier: from here on every VM operation is a DMA piece)

    // ---- the DMA schedule.  rank_limbs_tile_kernel lays the candidates' limbs out POSITION-major: [tile of 64][slab][block 0 / 1][limb]
    // [half][row % 32][16 bytes], i.e. the 6 KB image of a (tile, slab) position is contiguous and consecutive positions of a block follow
    // each other -- the loader's address is one scalar add per stage.  Piece p of a position: entity block p / 3, limb p % 3.  Wave w
    // issues piece w of EVERY position and, with every EVEN position p (block-relative), a second one: waves 0, 1 the pieces 4, 5 of p,
    // waves 2, 3 the pieces 4, 5 of p + 1 (one stage early: that buffer was consumed two stages ago) -- three DMA instructions per wave
    // and two stages, the same pattern for every wave, so the counted waits are exact (newer than position g + 1: the 4 positions behind
    // it, two of them even = 6 instructions).  Positions past the block's end re-read its last one.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_scr;
    const int G = (int)ntile * S;   // positions of this block
    const char* const pos_base = reinterpret_cast<const char*>(a.b.elimbs) + (e_begin >> 6) * (int64_t)(S * SCRR_STAGE);
    const uint32_t max_off = (uint32_t)(G - 1) * SCRR_STAGE;
    const uint32_t vo0 = (uint32_t)wv_s * SCRR_PIECE + lane16;             // piece w
    const uint32_t vo1 = (uint32_t)(4 + (wv_s & 1)) * SCRR_PIECE + lane16;   // piece 4 / 5
    const int late = wv_s >> 1;   // (waves 2, 3: their second piece belongs to the NEXT position)
    uint32_t ld_off = 0u;
    int ld_buf = 0;
#ifndef SCRR_FIRST_SLOT
#define SCRR_FIRST_SLOT 6    // the stage's slot that issues the wave's first piece (>= 2: behind the stage barrier of slot 1)
#endif
#ifndef SCRR_SECOND_SLOT
#define SCRR_SECOND_SLOT 10  // ... and its SECOND piece of an even position: two ~60 - 90-cycle DMA issues in ONE slot starve the matrix pipe (both in slot 2: 726 us; 6 / 10: 691 - 714, other splits within noise of it: profiles/r06y7 - r06y9)
#endif
    // this wave's piece(s) of the next position (EVEN: is it an even one?) into ring buffer ld_buf: the first piece, then -- for an even
    // position -- the second (issue_second; it may sit in a later slot of the stage: the counted waits only need both in front of the next
    // stage's), then the advance
    auto issue_first = [&]() __attribute__((always_inline)) {
        scrr_dma16(pos_base + ld_off, vo0, lds0 + (uint32_t)ld_buf * SCRR_STAGE + (uint32_t)wv_s * SCRR_PIECE);
    };
    auto issue_second = [&](auto even_c) __attribute__((always_inline)) {
        if constexpr (decltype(even_c)::value) {
            const uint32_t off1 = late ? min(ld_off + (uint32_t)SCRR_STAGE, max_off) : ld_off;
            scrr_dma16(pos_base + off1, vo1, lds0 + (uint32_t)((ld_buf + late) & (SCRR_NB - 1)) * SCRR_STAGE + (uint32_t)(4 + (wv_s & 1)) * SCRR_PIECE);
        }
        ld_buf = (ld_buf + 1) & (SCRR_NB - 1);
        ld_off = min(ld_off + (uint32_t)SCRR_STAGE, max_off);
    };
    auto issue = [&](auto even_c) __attribute__((always_inline)) { issue_first(); issue_second(even_c); };

    // per accumulator register (= query row of this lane): greater, equal; gmask / emask gather one "not greater" / "equal" bit per
    // output (32 = 16 tiles of two columns) before they are counted
    int cntg[16], cnte[16];
    uint32_t gmask[16], emask[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { cntg[r] = 0; cnte[r] = 0; gmask[r] = 0u; emask[r] = 0u; }
    v16i32 acc[3][2];   // [level][entity block]: level 0 = l0 l0', 1 = l0 l1' + l1 l0', 2 = l0 l2' + l1 l1' + l2 l0'

    int npend = 0;   // pairs parked in this wave's LDS buffer (wave-uniform)
    int2* const pend = reinterpret_cast<int2*>(thr_s + 2 * 128) + wv * SCRR_PEND;
    auto flush = [&]() {   // (inline assembly with its own vmcnt(0): it also drains this wave's DMA pieces -- rare, and only stricter)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int b0 = 0;
        if (lane == 63) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(b0) : "v"(a.b.counter), "v"(npend) : "memory");
        const int64_t base = __shfl(b0, 63, 64);
        for (int i = lane; i < npend; i += 64) {
            if (base + i < a.b.cap) {
                const uint64_t v = *reinterpret_cast<const uint64_t*>(pend + i);
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(a.b.pairs + base + i), "v"(v) : "memory");
            } else {   // the list is full: the call falls back to the exact kernel
                const int one = 1;
                asm volatile("global_store_dword %0, %1, off" :: "v"(a.b.counter + 1), "v"(one) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        npend = 0;
    };
    auto append = [&](uint32_t msk, int64_t et) {   // park the marked outputs (bit 2 r + ni of a lane) of this wave; <= SCRR_PEND of them
        const int mine = __popc(msk);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int total = __shfl(incl, 63, 64);
        if (!total) return;
        if (npend + total > SCRR_PEND) flush();
        int at = npend + incl - mine;
        while (msk) {
            const int bit = __builtin_ctz(msk);
            msk &= msk - 1;
            const int idx = 31 - bit, r = idx >> 1, ni = idx & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(et + ni * 32 + l31));
        }
        npend += total;
    };

    // The usual tile marks one or two outputs per WAVE (0.09 % of them at C2), and handing them over tile by tile -- a prefix sum over the
    // lanes (six dependent cross-lane steps) and a divergent loop, or a scalar walk over the marked lanes: either way a handful of VALU ->
    // SGPR round trips -- cost ~650 cycles per tile with the matrix pipe idle (profiles/r06y12_*: 51 us of the kernel's 712).  So a thread
    // parks its tile mask in LDS (one ds_write) and the wave hands over SCRR_UB tiles at a time: one prefix sum over the lanes' totals,
    // then every lane walks ITS marks (a handful of iterations for the wave).  A batch with more marks than the wave's parking buffer holds
    // (wild rows, non-finite values) goes tile by tile through the compacting path above.
    uint32_t* const und_s = reinterpret_cast<uint32_t*>(smem_scr + SCRR_LDS_UND) + tid;   // [slot][thread]
    auto hand_over = [&](const int t0, const int cnt) {   // the marks of tiles t0 .. t0 + cnt - 1 of this block (slots 0 .. cnt - 1)
        int mine = 0;
        uint32_t nz = 0u;   // which of the slots hold marks of this lane
#pragma unroll
        for (int i = 0; i < SCRR_UB; ++i) {
            const uint32_t v = (i < cnt) ? und_s[i * 256] : 0u;
            mine += __popc(v);
            nz |= v ? (1u << i) : 0u;
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int total = __shfl(incl, 63, 64);
        if (!total) return;
        if (total > SCRR_PEND) {
            for (int i = 0; i < cnt; ++i) {
                const uint32_t msk = und_s[i * 256];
                const int64_t et = e_begin + (int64_t)(t0 + i) * SCR_ET;
                if (__popcll(__ballot(msk != 0u)) <= SCRR_PEND / 32) append(msk, et);   // (<= 32 outputs per lane)
                else for (int ps = 0; ps < 8; ++ps) append(msk & (0xFu << (4 * ps)), et);   // (<= 4 per lane: 256 per wave)
            }
            return;
        }
        if (npend + total > SCRR_PEND) flush();
        int at = npend + incl - mine;
        uint32_t msk = 0u;
        int ti = 0;
        for (int left = mine; left > 0; --left) {   // (divergent: one mark of the lane per iteration, the next marked slot fetched when a mask runs out)
            const bool fresh = msk == 0u;
            const int tn = __builtin_ctz(nz | 0x10000u);
            ti = fresh ? tn : ti;
            nz = fresh ? (nz & (nz - 1u)) : nz;
            const uint32_t ld = und_s[ti * 256];
            msk = fresh ? ld : msk;
            const int bit = __builtin_ctz(msk);
            msk &= msk - 1u;
            const int idx = 31 - bit, r = idx >> 1, ni = idx & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(e_begin + (int64_t)(t0 + ti) * SCR_ET + ni * 32 + l31));
        }
        npend += total;
    };

    // this lane's 16 bytes inside a 1 KB piece: [half][row]
    const char* const frag_ptr = smem_scr + (lh * 512 + l31 * 16);
    v4i32 eb[2][2][3];   // [stage parity][entity block][limb]: the fragments of the NEXT stage are read while this one multiplies
    int rbuf = 0;        // ring buffer of the next fragment read
    auto read_frag = [&](const char* sb, int piece, v4i32& f) __attribute__((always_inline)) {
        const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)piece * SCRR_PIECE);
        f = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
    };

    scrr_static_for<SCRR_D>([&](auto ic) __attribute__((always_inline)) { issue(std::bool_constant<(decltype(ic)::value & 1) == 0>{}); });   // positions 0 .. D - 1 (9 instructions; newer than position 0: 7)
    static_assert(SCRR_D % 2 == 0 && SCRR_D >= 2 && SCRR_D + 2 <= SCRR_NB, "the counted waits are written for an even number of positions in flight");
    // (newer than position 0: positions 1 .. D - 1, D / 2 - 1 of them even)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((SCRR_D - 1) + (SCRR_D / 2 - 1)) : "memory");   // position 0 has landed: this wave's pieces, and everyone's
#pragma unroll
    for (int p = 0; p < 6; ++p) read_frag(frag_ptr, p, eb[0][p / 3][p % 3]);
    rbuf = 1;
    // the query fragments live in the accumulation half of the register file from here on (the matrix instruction reads its A operand
    // there directly): the other half holds the accumulators, the entity fragments and the epilogue
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int lb = 0; lb < 3; ++lb) asm volatile("" : "+a"(qf[s][lb]));

    // ---- the software pipeline.  Iteration t issues the matrix instructions of tile t and, BETWEEN them, the epilogue of tile t - 1
    // in slices of a few instructions (one wave per SIMD: nobody else fills the 32 cycles a matrix instruction occupies the pipe; a
    // wave that issues 156 of them back to back and its 700 epilogue instructions afterwards leaves the pipe idle half the time --
    // measured: MfmaUtil 0.39 with the stages written out but the epilogue behind them).  At the end of a tile its three integer
    // accumulator levels are folded into ONE fp32 value per output (F: 32 registers), which frees the accumulators for the next tile;
    // everything else of the epilogue works on F.  Iteration 0 runs the slices on NaNs (nothing counted, the undecided marks dropped),
    // iteration ntile runs the slices alone.
    // Every matrix instruction is its own inline-assembly statement (operand classes pinned: A in the accumulation file, B / C / D
    // in the vector file -- no copies between the files), separated from the slice behind it by scheduling barriers.
    // The decision arithmetic, per OUTPUT, is integer: g = (L0 << 8) + L1 + (L2 >> 8) (the three levels folded in units of 2^24 A B_t; the
    // floor of L2 / 2^8 is below one unit, inside the 2^27 A B the bound carries for the fold) against four integer thresholds of its
    // (query row, tile) -- ceil((G + E) / sigma), ceil((L - E) / sigma), ceil((EL + E) / sigma), ceil((EH - E) / sigma), sigma = 2^24 A_i
    // B_t, E = E'(row, the tile's maxima) >= E'(row, candidate) -- computed once per row and tile by the thread that owns the row
    // (10 operations) instead of ~21 fp32 operations per output: with one wave per SIMD the epilogue's instruction count, not the
    // matrix pipe, set this kernel's time (profiles/r06e_screen_r_ablations.txt).  An output the tile-wide bound cannot decide is
    // marked and rechecked exactly like one the per-candidate bound could not decide (a few more of them).
    int G0[16], G1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0[r] = SCRR_NONE; G1[r] = SCRR_NONE; }   // (the sentinel of "no output": smaller than everything, counted nowhere)
    int64_t et_prev = e_begin;
    const int row0 = wq + 4 * lh;
    int4 th = thr_s[row0], th_n = th;
    uint32_t undm = 0u;
    float eb_t = 0.f, isig = 0.f;   // (the threshold slices' temporaries)
    int sd4 = 0, sd1 = 0, sd2 = 0, seq = 0;   // (a row's differences in flight between its slots)

    // The tile loop is written out for two consecutive tiles: with an odd number of slabs the parity of a position, of the fragment
    // registers a stage multiplies from and of the threshold buffer alternate from tile to tile -- compile-time constants of the copy
    // (TP), not run-time selects and a branch per stage.
    auto one_tile = [&](auto tp_c, const int t) __attribute__((always_inline)) {
        constexpr int TP = decltype(tp_c)::value;          // t & 1
        constexpr int TPS = (S & 1) ? TP : 0;              // parity of the tile's first position
        // {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t}: wave-uniform.  From LDS: a vector-memory load here would share vmcnt with the DMA
        // pieces, and the compiler's wait for it (vmcnt(0)) drained the ring once per tile
        const float4 tm4 = tm_s[t < (int)ntile ? t : (int)ntile - 1];
        // MM = true: the matrix work of tile t between the slices of tile t - 1; false (iteration ntile): the slices alone
        auto tile_body = [&](auto mm_c) __attribute__((always_inline)) {
        constexpr bool MM = decltype(mm_c)::value;
        scrr_static_for<S>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            constexpr int P = (s + TPS) & 1;
            const char* sb = nullptr;
            scrr_static_for<12>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int k = s * 12 + m;   // slot of the tile: 0 .. 12 S - 1
                // (acc level, entity block, query limb, entity limb) of the stage's m-th matrix instruction: two on the SAME accumulator
                // are never adjacent
                constexpr int LV[12] = {0, 0, 2, 2, 1, 1, 2, 2, 1, 1, 2, 2};
                constexpr int NI[12] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1};
                constexpr int QL[12] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2};
                constexpr int EL[12] = {0, 0, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0};
                if constexpr (MM) {
                    v16i32& C = acc[LV[m]][NI[m]];
                    const v4i32& A = qf[s][QL[m]];
                    const v4i32& Bm = eb[P][NI[m]][EL[m]];
                    if constexpr (SCRR_ABLATE & 2) asm volatile("" : "+v"(C) : "a"(A), "v"(Bm));
                    // (s_nop 1: a tile's first fragments are parked across the tile's end -- in the accumulation file when the vector
                    // file is full -- and copied back right in front of their first use; gfx90a+ wants TWO wait states between a VALU
                    // write of a register and a matrix instruction that reads it, and the compiler, which inserts them for its own matrix
                    // instructions, cannot see into this statement: scripts/check_mfma_hazards.py scans the generated code for the pattern)
                    else if constexpr (s == 0 && m < 6) asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(C) : "a"(A), "v"(Bm));
                    else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(C) : "a"(A), "v"(Bm));
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- the stage's own work, spread over its first slots
                if constexpr (!MM) {
                } else if constexpr (m == 1) {
                    // this wave's pieces of position g + 1 have landed (the D - 2 positions behind it stay in flight) and so have the
                    // other waves'; everyone is past the matrix instructions that consumed position g - 2, whose buffer the DMA reuses
                    // (newer than position g + 1: the D - 2 positions behind it, half of them even)
                    if constexpr (SCRR_ABLATE & 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SCRR_D - 2) + (SCRR_D - 2) / 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((SCRR_D - 2) + (SCRR_D - 2) / 2) : "memory");
                } else if constexpr (m == 2) {
                    sb = frag_ptr + (size_t)rbuf * SCRR_STAGE;   // position g + 1
                    rbuf = (rbuf + 1) & (SCRR_NB - 1);
                } else if constexpr (m >= 3 && m <= 8) {
                    if constexpr (!(SCRR_ABLATE & 32)) read_frag(sb, m - 3, eb[P ^ 1][(m - 3) / 3][(m - 3) % 3]);
                }
                if constexpr (MM && m == SCRR_FIRST_SLOT && !(SCRR_ABLATE & 16)) issue_first();   // position g + D
                if constexpr (MM && m == SCRR_SECOND_SLOT && !(SCRR_ABLATE & 16)) issue_second(std::bool_constant<((TPS + s + SCRR_D) & 1) == 0>{});
                // ---- slice k of the previous tile's epilogue: row k / 8 (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4
                // (lane >> 5)), step k % 8
                // (a tile of S slabs has 12 S slots; the 128 steps -- 16 rows x 8 -- go to its first SLICE_SLOTS of them in order, one per slot
                // at S = 13, several per slot for narrower rows)
                constexpr int SLICE_SLOTS = 12 * S - 28 < 128 ? 12 * S - 28 : 128;
                constexpr int Q_LO = k < SLICE_SLOTS ? (k * 128 + SLICE_SLOTS - 1) / SLICE_SLOTS : 128;
                constexpr int Q_HI = k < SLICE_SLOTS ? ((k + 1) * 128 + SLICE_SLOTS - 1) / SLICE_SLOTS : 128;
                if constexpr (!(SCRR_ABLATE & 1)) scrr_static_for<Q_HI - Q_LO>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = Q_LO + decltype(qc)::value;
                    constexpr int r = q >> 3, j = q & 7;
                    // A row = two outputs (entity blocks 0 / 1), no scalar register and no branch in the chain (a VALU -> SGPR -> VALU round
                    // trip costs ~18 cycles next to matrix instructions, profiles/r06_mfma_filler_probe.txt; a compare + branch per row
                    // ~75, profiles/r06x3_*): d4 = g - Gi, d1 = g - Li, d2 = g - ELi, d3 = g - EHi (thresholds and g inside +-2^30: no
                    // overflow).  sign(d4) = "not greater", sign(~d2 & d3) = "equal" and sign(~d1 & d4 & ~equal) = neither smaller, greater
                    // nor equal -- UNDECIDED -- are shifted into bit masks: two per row, counted every 16 tiles, and one per tile (the 32
                    // outputs of a lane: bit 31 - (2 r + ni)), whose set bits go to the recheck list at the tile's end.  The row's ~20
                    // operations are spread over its slots at <= 3 per slot: what a lone wave issues behind a matrix instruction runs
                    // in its shadow only as long as the next one is not waiting behind it (SQ_VALU_MFMA_COEXEC_CYCLES 0.11 -> 0.19 of
                    // the matrix pipe's busy cycles, profiles/r06x9_*).
                    if constexpr (j == 0) {
                        if constexpr (r < 15) { constexpr int rn = ((r + 1) & 3) + 8 * ((r + 1) >> 2); th_n = thr_s[(TP ^ 1) * 128 + row0 + rn]; }
                        sd4 = G0[r] - th.x; sd1 = G0[r] - th.y;
                        asm volatile("" : "+v"(sd4), "+v"(sd1));
                    } else if constexpr (j == 1) {
                        const int d2 = G0[r] - th.z, d3 = G0[r] - th.w;
                        seq = ~d2 & d3;
                        asm volatile("" : "+v"(seq));
                    } else if constexpr (j == 2) {
                        gmask[r] = __builtin_amdgcn_alignbit(gmask[r], (uint32_t)sd4, 31);
                        emask[r] = __builtin_amdgcn_alignbit(emask[r], (uint32_t)seq, 31);
                        undm = __builtin_amdgcn_alignbit(undm, (uint32_t)(~sd1 & sd4 & ~seq), 31);
                        asm volatile("" : "+v"(gmask[r]), "+v"(emask[r]), "+v"(undm));
                    } else if constexpr (j == 3) {
                        sd4 = G1[r] - th.x; sd1 = G1[r] - th.y; sd2 = G1[r] - th.z;
                        asm volatile("" : "+v"(sd4), "+v"(sd1), "+v"(sd2));
                    } else if constexpr (j == 4) {
                        const int d3 = G1[r] - th.w;
                        seq = ~sd2 & d3;
                        gmask[r] = __builtin_amdgcn_alignbit(gmask[r], (uint32_t)sd4, 31);
                        asm volatile("" : "+v"(seq), "+v"(gmask[r]));
                    } else if constexpr (j == 5) {
                        emask[r] = __builtin_amdgcn_alignbit(emask[r], (uint32_t)seq, 31);
                        undm = __builtin_amdgcn_alignbit(undm, (uint32_t)(~sd1 & sd4 & ~seq), 31);
                        asm volatile("" : "+v"(emask[r]), "+v"(undm));
                    } else if constexpr (j == 7) {
                        th = th_n;
                    }
                });
                // ---- this tile's thresholds, by the thread that owns the query row (free slots behind the slices)
                if constexpr (!MM) {
                } else if constexpr (k == 12 * S - 26) {
                    // E = c (gamma |W q|_2 max |W e|_2 + A max |e|_1 / 2 + B_t (|q|_1 / 2 + drop A)); 1 / sigma = 1 / (2^24 A) * 1 / B_t
                    eb_t = __builtin_fmaf(rq_y, tm4.y, __builtin_fmaf(rq_z, tm4.z, rq_w * tm4.x));
                    isig = rq_iA * tm4.w;
                } else if constexpr (k == 12 * S - 23) {
                    // ceil((T +- E) / sigma) as an int32 (|g| < 2^29.1 for S <= 13: thresholds beyond (-2^30 + 64, 2^30] clamp without changing a
                    // decision); NaN (inf - inf) and scales outside [2^-20, 2^100] (a product could leave fp32's range): nothing decided
                    const bool okscale = (isig >= 0x1p-20f) && (isig <= 0x1p100f);
                    const float xa = __builtin_ceilf((thrA + eb_t) * isig), xb = __builtin_ceilf((thrB - eb_t) * isig);
                    const bool oka = okscale && (xa == xa), okb = okscale && (xb == xb);
                    const int ia = oka ? (int)fminf(fmaxf(xa, -1073741760.f), 1073741824.f) : (1 << 30);      // (never >=)
                    const int ib = okb ? (int)fminf(fmaxf(xb, -1073741760.f), 1073741824.f) : -1073741760;    // (never <, but for the sentinel)
                    int2* const dst = reinterpret_cast<int2*>(thr_s + TP * 128 + (tid & 127)) + (tid >> 7);
                    *dst = make_int2(ia, ib);
                } else if constexpr (k == 12 * S - 16) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the thresholds are in LDS before this wave's next barrier (stage S - 1's)
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        };
        if (t < (int)ntile) tile_body(std::true_type{}); else tile_body(std::false_type{});
        // A VALU read of a matrix instruction's result wants up to 19 wait states behind it (8 / 16 passes), and the compiler cannot see
        // into the statements above.  The accumulators are operands of the pad: nothing that reads them moves in front of it.
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]) :: "memory");
        // ---- the previous tile's undecided outputs go to the list
        if (t >= 1 && !(SCRR_ABLATE & 64)) {
            undm &= rowmask;
            if (et_prev + SCR_ET > e_end) {   // (the range's last tile)
                if (et_prev + l31 >= e_end) undm &= 0x55555555u;        // candidate of block 0 (the odd bits) beyond the range
                if (et_prev + 32 + l31 >= e_end) undm &= 0xAAAAAAAAu;   // candidate of block 1 beyond the range
            }
            const int slot = (t - 1) & (SCRR_UB - 1);
            und_s[slot * 256] = undm;
            if (slot == SCRR_UB - 1 || t == (int)ntile) hand_over(t - 1 - slot, slot + 1);
        }
        undm = 0u;
        if ((t & 15) == 15) {   // 32 "not greater" bits per row gathered: count the others
#pragma unroll
            for (int r = 0; r < 16; ++r) { cntg[r] += 32 - __popc(gmask[r]); gmask[r] = 0u; cnte[r] += __popc(emask[r]); emask[r] = 0u; }
        }
        // ---- this tile's accumulators -> g = (L0 << 8) + L1 + (L2 >> 8); candidates beyond the range: the sentinel
        if (t < (int)ntile) {
            const int64_t et = e_begin + (int64_t)t * SCR_ET;
#pragma unroll
            for (int r = 0; r < ((SCRR_ABLATE & 8) ? 1 : 16); ++r) {
                G0[r] = (int)(((uint32_t)acc[0][0][r] << 8) + (uint32_t)acc[1][0][r] + (uint32_t)(acc[2][0][r] >> 8));
                G1[r] = (int)(((uint32_t)acc[0][1][r] << 8) + (uint32_t)acc[1][1][r] + (uint32_t)(acc[2][1][r] >> 8));
            }
            if (et + SCR_ET > e_end) {   // (the range's last tile)
                const bool v0 = et + l31 < e_end, v1 = et + 32 + l31 < e_end;
#pragma unroll
                for (int r = 0; r < 16; ++r) { G0[r] = v0 ? G0[r] : SCRR_NONE; G1[r] = v1 ? G1[r] : SCRR_NONE; }
            }
            et_prev = et;
            th = thr_s[TP * 128 + row0];
        }
    };
    for (int t = 0; t <= (int)ntile; t += 2) {
        one_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 > (int)ntile) break;
        one_tile(std::integral_constant<int, 1>{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the positions requested past the end: their DMA must not outlive the workgroup's LDS
    if (npend) flush();
    // ---- per query row: sum over the 32 lanes that share it ----
    const int nbits = 2 * (((int)ntile + 1) & 15);   // bits gathered since the last count (the masks start at zero: the upper bits are clear)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int g = cntg[r] + nbits - __popc(gmask[r]), e = cnte[r] + __popc(emask[r]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
        const int64_t qi = q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (l31 == 0 && qi < a.n) {
            if (g) atomicAdd(&a.b.counts[2 * qi + 0], g);
            if (e) atomicAdd(&a.b.counts[2 * qi + 1], e);
        }
    }
}

}  // namespace kge
