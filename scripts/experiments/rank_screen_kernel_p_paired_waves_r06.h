// NOT PART OF THE PRODUCT (round 6): the paired-wave split of the limb products.  Correct (31 identity tests, same ranks) and SLOWER than
// rank_screen_kernel_r: 1 054 us with a barrier per stage-time, 963 with one per window of four, 1 004 with rolling fragment reloads against
// ~830; its bare skeleton (matrix instructions + feed) 674 us against r's ~500 (profiles/r06t_*, r06w_*).  Two waves per SIMD means 128 + 128
// registers: no room to double-buffer the entity fragments, and in-place reloads leave half a stage of LDS latency uncovered.  To build it:
// include behind kge_rank_screen_r.h and launch with 512 threads, SCRP_LDS_BYTES, on the tile-scale path of run_screen.
// Round 6: the screening kernel with the six limb products SPLIT BETWEEN TWO WAVES of a SIMD.
// Included by kge_rank.hip behind kge_rank_screen_r.h, whose tile-wide candidate scales (rank_limbs_tile_kernel), integer thresholds per
// (query row, tile), sign-bit counting and DMA helper it shares: the counts are the same integers.
//
// Why.  rank_screen_kernel_r keeps a wave's query limbs resident (156 registers), which leaves room for ONE wave per SIMD -- and one wave
// cannot hide its own LDS-DMA issue (~90 cycles a piece in the real stream), fragment reads, the fold between tiles and the epilogue
// slices under its matrix instructions (profiles/r06h_*, r06j_*: the parts add up; MfmaUtil 0.43).  Here the 32 query rows x 64 candidates
// of a tile belong to a PAIR of waves on one SIMD (512-thread workgroups, 256 registers per wave):
//   wave A holds query limbs 0 and 2 (104 registers) and issues 8 of a stage's 12 matrix instructions: q0 e0 -> level 0, q0 e1 -> level
//          1, q0 e2 + q2 e0 -> level 2; at a tile's end it folds its three partial levels into one integer per output and hands the 32 of
//          them to its partner through LDS;
//   wave B holds query limb 1 (52 registers) and issues the other 4: q1 e0 -> level 1, q1 e1 -> level 2; it adds its own fold to A's
//          (two floors of L2 / 2^8 instead of one: below two units, inside the eight the bound carries for the fold) and owns the whole
//          epilogue (thresholds, sign-bit slices, counts, the undecided list).  B runs LAG stages behind A, so the two waves' tile ends --
//          where neither issues matrix instructions -- never coincide.
// Both consume the same ring of entity-slab positions (LDS-DMA, 0.75 pieces per wave and stage: three pieces of one position every
// fourth stage-time, `s_waitcnt vmcnt(3)` per stage-time), one raw s_barrier per stage-time for the eight waves.
#pragma once

#ifndef SCRP_ABLATE
#define SCRP_ABLATE 0   // development: 1 no epilogue slices, 2 no tile-end work (fold / hand-over / list), 4 no window barrier, 8 no DMA -- wrong counts, timing only
#endif

namespace kge {

constexpr int SCRP_THREADS = 512;
constexpr int SCRP_D = 8;                           // positions requested ahead of wave A: two windows of four
constexpr int SCRP_LAG = 4;                         // stage-times wave B runs behind wave A
constexpr int SCRP_NB = 16;                         // ring buffers (>= D + LAG + 2; a power of two)
constexpr size_t SCRP_HAND_BYTES = 32 * 64 * 4;     // one pair's hand-over: 32 integers per lane
constexpr int SCRP_QL = 3;                          // trailing slabs of query limb 2 that wave A keeps in LDS instead of registers
constexpr size_t SCRP_LDS_BYTES = (size_t)SCRP_NB * SCRR_STAGE + 2 * 128 * 16 + 4 * SCRP_HAND_BYTES + 4 * (size_t)SCRR_PEND * 8 + 4 * SCRP_QL * 1024;   // 155 648

template <int S>
__global__ __launch_bounds__(SCRP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void rank_screen_kernel_p(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];   // (the ONE LDS object of the kernel)
    int4* const thr_s = reinterpret_cast<int4*>(smem_scr + (size_t)SCRP_NB * SCRR_STAGE);   // [2][128]
    char* const hand_s = reinterpret_cast<char*>(thr_s + 2 * 128);                           // [4 pairs][8][64] int4
    int2* const pend_s = reinterpret_cast<int2*>(hand_s + 4 * SCRP_HAND_BYTES);              // [4][SCRR_PEND]
    uint4* const q2_s = reinterpret_cast<uint4*>(pend_s + 4 * SCRR_PEND);                    // [4 pairs][SCRP_QL][64]: query limb 2, last slabs

    if (a.wild_mode == 2 && screen_wild(a.b.counter, a.m)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 .. 7
    const int role = wv >> 2, w = wv & 3;                      // role 0: wave A, 1: wave B; w: the pair = the 32-query block
    const int l31 = lane & 31, lh = lane >> 5;
    const int wq = w * 32;
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
    const int G = (int)ntile * S;   // positions of this block

    const uint32_t blk_stride = (uint32_t)S * SCR_BLK_SLAB;   // bytes between consecutive 32-row query blocks
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_scr;
    const char* const qsrc = reinterpret_cast<const char*>(a.b.qlimbs) + ((q0 + 32 * w) >> 5) * (int64_t)blk_stride;
    const char* const frag_ptr = smem_scr + (lh * 512 + l31 * 16);   // this lane's 16 bytes inside a 1 KB piece: [half][row]

    // ---- the DMA schedule and the only synchronisation, the same for all eight waves.  Stage-times come in WINDOWS of four; at a
    // window's start (kt % 4 == 0) a wave waits for its pieces of that window (`s_waitcnt vmcnt(3)`: all but its newest three), meets the
    // others at one raw s_barrier and requests its three pieces of the window two ahead: the six pieces of position 4 (j + 2) + wv / 2
    // come from the two waves wv / 2, three each (pieces 3 (wv % 2) + 0 .. 2).  Inside a window nobody waits for anybody: wave A's fold
    // at a tile's end and wave B's epilogue drift against the partner's matrix instructions instead of stopping them (one barrier per
    // stage-time measured 55 % of the wave cycles parked).  Ring: window j (wave A), j - 1 (wave B, LAG = one window), j + 1 and j + 2
    // in flight = 16 positions; window j + 2 overwrites window j - 2, which everyone finished before this barrier.
    static_assert(SCRP_LAG == 4 && SCRP_D == 8 && SCRP_NB == 16, "the window scheme is written for LAG = one window, two windows ahead");
    const char* const pos_base = reinterpret_cast<const char*>(a.b.elimbs) + (e_begin >> 6) * (int64_t)(S * SCRR_STAGE);
    const int my_phase = wv >> 1;
    const uint32_t my_piece0 = (uint32_t)(wv & 1) * 3u;
    int kt = -SCRP_D;   // stage-time
    auto dma_window = [&]() {   // kt is a multiple of 4
        const int P = kt + SCRP_D + my_phase;
        const uint32_t goff = (uint32_t)min(P, G - 1) * SCRR_STAGE + my_piece0 * SCRR_PIECE;
        const uint32_t ldst = lds0 + (uint32_t)(P & (SCRP_NB - 1)) * SCRR_STAGE + my_piece0 * SCRR_PIECE;
#pragma unroll
        for (int i = 0; i < 3; ++i) scrr_dma16(pos_base + goff, lane16 + (uint32_t)i * SCRR_PIECE, ldst + (uint32_t)i * SCRR_PIECE);
    };
    auto stage_sync = [&]() {
        if ((kt & 3) == 0) {
            if constexpr (SCRP_ABLATE & 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
            if constexpr (!(SCRP_ABLATE & 8)) dma_window();
        }
        ++kt;
    };
    auto idle_stage_time = [&]() { stage_sync(); };
    auto read_frag = [&](const char* sb, int piece, v4i32& f) __attribute__((always_inline)) {
        const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)piece * SCRR_PIECE);
        f = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
    };

    if (role == 0) {
        // =================================================== wave A ===================================================
        // Two waves per SIMD: the compiler gives each file half of the 256 registers, so the operands are placed by hand.  Wave A:
        // the accumulation file holds query limb 0 (52), the first S - 3 slabs of limb 2 (40) and accumulator level 2 (32); the vector
        // file levels 0, 1 (64) and the entity fragments (24); the last three slabs of limb 2 wait in LDS (a fragment read in their
        // stages).  The matrix instruction takes A and C / D from either file.
        constexpr int QA2 = S - SCRP_QL;
        v4i32 qa0[S], qa2[QA2];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint4 u = *reinterpret_cast<const uint4*>(qsrc + ((size_t)s * SCR_BLK_SLAB + lane16));
            qa0[s] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
            const uint4 u2 = *reinterpret_cast<const uint4*>(qsrc + ((size_t)s * SCR_BLK_SLAB + 2048u + lane16));
            if (s < QA2) qa2[s] = v4i32{(int)u2.x, (int)u2.y, (int)u2.z, (int)u2.w};
            else q2_s[(w * SCRP_QL + (s - QA2)) * 64 + lane] = u2;
        }
        __syncthreads();   // (the last ordinary loads of the kernel are behind this barrier: from here on every VM operation is a DMA piece)
        dma_window(); kt += 4; dma_window(); kt += 4;   // windows 0 and 1
#pragma unroll
        for (int s = 0; s < S; ++s) {
            asm volatile("" : "+a"(qa0[s]));
            if (s < QA2) asm volatile("" : "+a"(qa2[s]));
        }
        v16i32 acc[3][2];   // [level][entity block]: this wave's partial levels
        v4i32 eb[2][3];     // [entity block][limb]
        int rpos = 0;       // position of the next fragment read
        for (int t = 0; t < (int)ntile; ++t) {
            scrr_static_for<S>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                // The stage's fragments were reloaded in place during the previous stage, each right behind its last use (below) -- unless
                // this stage opens a window: then its position may only be read behind the barrier.
                const bool opens = (kt & 3) == 0;
                stage_sync();
                if (opens) {
                    const char* const sb = frag_ptr + (size_t)(rpos & (SCRP_NB - 1)) * SCRR_STAGE;
#pragma unroll
                    for (int p = 0; p < 6; ++p) read_frag(sb, p, eb[p / 3][p % 3]);
                }
                ++rpos;
                const bool roll = (kt & 3) != 0;   // (the next stage does not open a window: its fragments replace this stage's as they retire)
                const char* const nb = frag_ptr + (size_t)(rpos & (SCRP_NB - 1)) * SCRR_STAGE;
                // (level, entity block, query limb index in qa, entity limb): limb 0 of the candidates first and both of its uses
                // together, so that every fragment retires early; two on the SAME accumulator are never adjacent
                constexpr int LV[8] = {0, 0, 2, 2, 1, 1, 2, 2};
                constexpr int NI[8] = {0, 1, 0, 1, 0, 1, 0, 1};
                constexpr int QJ[8] = {0, 0, 1, 1, 0, 0, 0, 0};
                constexpr int EL[8] = {0, 0, 0, 0, 1, 1, 2, 2};
                constexpr bool FIRST[8] = {true, true, true, true, true, true, false, false};   // (first product of a tile into its accumulator)
                v4i32 q2l = {0, 0, 0, 0};   // (a slab of limb 2 that lives in LDS)
                if constexpr (s >= QA2) { const uint4 u = q2_s[(w * SCRP_QL + (s - QA2)) * 64 + lane]; q2l = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
                scrr_static_for<8>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    v16i32& C = acc[LV[m]][NI[m]];
                    const v4i32& A = QJ[m] == 0 ? qa0[s] : (s < QA2 ? qa2[s < QA2 ? s : 0] : q2l);
                    const v4i32& Bm = eb[NI[m]][EL[m]];
                    if constexpr (LV[m] < 2) {           // levels 0, 1: C / D in the vector file, A = limb 0 in the accumulation file
                        if constexpr (s == 0 && FIRST[m]) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(C) : "a"(A), "v"(Bm));
                        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(C) : "a"(A), "v"(Bm));
                    } else if constexpr (QJ[m] == 0 || s < QA2) {   // level 2 (accumulation file) from limb 0 or the resident part of limb 2
                        if constexpr (s == 0 && FIRST[m]) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&a"(C) : "a"(A), "v"(Bm));
                        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(C) : "a"(A), "v"(Bm));
                    } else {                             // level 2 from the part of limb 2 that lives in the vector file
                        if constexpr (s == 0 && FIRST[m]) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&a"(C) : "v"(A), "v"(Bm));
                        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(Bm));
                    }
                    // the fragments that have just had their last use: the next stage's take their place
                    if constexpr (m == 3) { if (roll) { read_frag(nb, 0, eb[0][0]); read_frag(nb, 3, eb[1][0]); } }
                    else if constexpr (m == 5) { if (roll) { read_frag(nb, 1, eb[0][1]); read_frag(nb, 4, eb[1][1]); } }
                    else if constexpr (m == 7) { if (roll) { read_frag(nb, 2, eb[0][2]); read_frag(nb, 5, eb[1][2]); } }
                });
            });
            // (a VALU read of a matrix instruction's result wants up to 19 wait states behind it; the compiler cannot see into the
            // statements above)
            asm volatile("s_nop 15\n\ts_nop 7"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+a"(acc[2][0]), "+a"(acc[2][1]) :: "memory");
            // this wave's share of g = (L0 << 8) + L1 + (L2 >> 8), 32 integers per lane, to the pair's hand-over block (the partner
            // reads it LAG stage-times later, behind as many barriers; the next tile's values come 13 stage-times later)
            int4* const hand = reinterpret_cast<int4*>(hand_s + (size_t)w * SCRP_HAND_BYTES) + lane;
#pragma unroll
            for (int jj = 0; jj < ((SCRP_ABLATE & 2) ? 1 : 8); ++jj) {
                int v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = 4 * jj + c, r = idx >> 1, ni = idx & 1;
                    v[c] = (int)(((uint32_t)acc[0][ni][r] << 8) + (uint32_t)acc[1][ni][r] + (uint32_t)(acc[2][ni][r] >> 8));
                }
                hand[jj * 64] = make_int4(v[0], v[1], v[2], v[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < SCRP_LAG; ++i) idle_stage_time();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the positions requested past the end: their DMA must not outlive the workgroup's LDS
        return;
    }

    // ======================================================= wave B =======================================================
    // this THREAD's query row and pair of thresholds (see rank_screen_kernel_r): the 256 threads of the four B waves own the 128 rows
    float rq_y, rq_z, rq_w, rq_iA, thrA, thrB;
    const int trow = tid & 127, tpair = (tid >> 7) & 1;
    {
        const bool okq = q0 + trow < a.n;
        const float4 m4 = a.b.qm[okq ? q0 + trow : a.n - 1];
        const float c = 1.f + 0x1p-10f;
        rq_y = m4.y * c; rq_z = m4.x * c; rq_w = fmaf(a.drop, m4.x, m4.z) * c;
        rq_iA = 0x1p-24f / m4.x;
        const float2 t2 = a.b.qt[okq ? q0 + trow : a.n - 1];
        const float s1 = isfinite(t2.x) ? 0x1p-20f * fabsf(t2.x) : 0.f, s2 = isfinite(t2.y) ? 0x1p-20f * fabsf(t2.y) : 0.f;
        thrA = tpair == 0 ? t2.y + s2 : t2.x + s1;   // G | EL
        thrB = tpair == 0 ? t2.x - s1 : t2.y - s2;   // L | EH
        if (tpair == 0) {   // (both parities start as "nothing decided")
            thr_s[trow] = make_int4(1 << 30, -1073741760, 1 << 30, -1073741760);
            thr_s[128 + trow] = make_int4(1 << 30, -1073741760, 1 << 30, -1073741760);
        }
    }
    uint32_t thr_dst = (uint32_t)trow * 16u + (uint32_t)tpair * 8u;   // this thread's pair inside a parity's 2 KB of thresholds
    uint32_t rowmask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rowmask |= (q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh < a.n) ? (3u << (2 * r)) : 0u;
    v4i32 qb[S];   // query limb 1
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint4 u = *reinterpret_cast<const uint4*>(qsrc + ((size_t)s * SCR_BLK_SLAB + 1024u + lane16));
        qb[s] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
    }
    __syncthreads();
    dma_window(); kt += 4; dma_window(); kt += 4;   // windows 0 and 1
#pragma unroll
    for (int s = 0; s < S; ++s) asm volatile("" : "+a"(qb[s]));

    // gem[r] gathers TWO bits per output of the row, "not greater" then "equal" (32 bits = 8 tiles of two columns); every 8 tiles they
    // are counted, summed over the 32 lanes that share the row and added to the call's counts (no per-lane counters, one mask register
    // per row: the vector file is full)
    uint32_t gem[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gem[r] = 0u;
    auto count_masks = [&](int noutputs) {   // outputs gathered per row since the last count
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int g = noutputs - __popc(gem[r] & 0xAAAAAAAAu), e = __popc(gem[r] & 0x55555555u);
            gem[r] = 0u;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
            const int64_t qi = q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (l31 == 0 && qi < a.n) {
                if (g) atomicAdd(&a.b.counts[2 * qi + 0], g);
                if (e) atomicAdd(&a.b.counts[2 * qi + 1], e);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the atomics are out of the way of the counted DMA waits)
    };
    v16i32 bcc[2][2];   // [level 1 / 2][entity block]
    v4i32 eb[2][2];     // [entity block][limb 0 / 1]
    int npend = 0;
    int2* const pend = pend_s + w * SCRR_PEND;
    auto flush = [&]() {   // (its vmcnt(0) also drains this wave's DMA pieces -- rare, and only stricter)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int b0 = 0;
        if (lane == 63) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(b0) : "v"(a.b.counter), "v"(npend) : "memory");
        const int64_t base = __shfl(b0, 63, 64);
        for (int i = lane; i < npend; i += 64) {
            if (base + i < a.b.cap) {
                const uint64_t v = *reinterpret_cast<const uint64_t*>(pend + i);
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(a.b.pairs + base + i), "v"(v) : "memory");
            } else {
                const int one = 1;
                asm volatile("global_store_dword %0, %1, off" :: "v"(a.b.counter + 1), "v"(one) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        npend = 0;
    };
    auto append = [&](uint32_t msk, int64_t et) {
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
            const int r = bit >> 1, ni = bit & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(et + ni * 32 + l31));
        }
        npend += total;
    };

    int G0[16], G1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0[r] = SCRR_NONE; G1[r] = SCRR_NONE; }
    int nearv = 0;
    int64_t et_prev = e_begin;
    const int row0 = wq + 4 * lh;
    int4 th = thr_s[row0];
    uint32_t undm = 0u;
    const float4* const tmeta = a.b.tm + (e_begin >> 6);
    int rpos = 0;

    // the epilogue of one row of the PREVIOUS tile (sign-bit fast path, rank_screen_kernel_r's): both outputs, then the rare slow path
    auto row_slice = [&](auto rc, int t) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int g = ni ? G1[r] : G0[r];
            const int d4 = g - th.x, d1 = g - th.y, d2 = g - th.z, d3 = g - th.w;
            const int eqs = ~d2 & d3;
            gem[r] = __builtin_amdgcn_alignbit(__builtin_amdgcn_alignbit(gem[r], (uint32_t)d4, 31), (uint32_t)eqs, 31);
            const int und = ~d1 & d4 & ~eqs;
            nearv = ni ? (nearv | und) : und;
        }
        asm volatile("" : "+v"(gem[r]), "+v"(nearv));
        if (__ballot(nearv < 0) != 0ull) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int g = ni ? G1[r] : G0[r];
                const bool near = (g >= th.y) && (g < th.x), eq = (g >= th.z) && (g < th.w);
                undm |= (near && !eq) ? (1u << (2 * r + ni)) : 0u;
            }
        }
        if constexpr (r < 15) { constexpr int rn = ((r + 1) & 3) + 8 * ((r + 1) >> 2); th = thr_s[((t + 1) & 1) * 128 + row0 + rn]; }
    };

#pragma unroll
    for (int i = 0; i < SCRP_LAG; ++i) idle_stage_time();
    for (int t = 0; t <= (int)ntile; ++t) {
        if (t < (int)ntile) {
            const float4 tm4 = tmeta[t];   // {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t}: wave-uniform
            scrr_static_for<S>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                const bool opens = (kt & 3) == 0;   // (see wave A: full loads behind a window's barrier, rolling reloads otherwise)
                stage_sync();
                if (opens) {
                    const char* const sb = frag_ptr + (size_t)(rpos & (SCRP_NB - 1)) * SCRR_STAGE;
                    read_frag(sb, 0, eb[0][0]); read_frag(sb, 3, eb[1][0]); read_frag(sb, 1, eb[0][1]); read_frag(sb, 4, eb[1][1]);
                }
                ++rpos;
                const bool roll = (kt & 3) != 0;
                const char* const nb = frag_ptr + (size_t)(rpos & (SCRP_NB - 1)) * SCRR_STAGE;
                // q1 e0 -> level 1, q1 e1 -> level 2
                scrr_static_for<4>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    v16i32& C = bcc[m >> 1][m & 1];
                    const v4i32& A = qb[s];
                    v4i32& Bm = eb[m & 1][m >> 1];
                    // (C / D and A in the accumulation file -- 64 + 52 registers --, the vector file is the epilogue's)
                    if constexpr (s == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&a"(C) : "a"(A), "v"(Bm));
                    else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(C) : "a"(A), "v"(Bm));
                    if (roll) read_frag(nb, 3 * (m & 1) + (m >> 1), Bm);   // (its only use: the next stage's fragment takes its place)
                });
                // the previous tile's epilogue, rows in order (each slice fetches the next row's thresholds): stages 0 .. 9 one row,
                // stages 10 .. 12 two
                static_assert(S == 13, "the row schedule below is written for 13 stages");
                if constexpr (SCRP_ABLATE & 1) {
                } else if constexpr (s < 10) row_slice(std::integral_constant<int, s>{}, t);
                else { row_slice(std::integral_constant<int, 2 * s - 10>{}, t); row_slice(std::integral_constant<int, 2 * s - 9>{}, t); }
                // this tile's integer thresholds, by the thread that owns the query row (see rank_screen_kernel_r)
                if constexpr (s == 5) {
                    const float eb_t = __builtin_fmaf(rq_y, tm4.y, __builtin_fmaf(rq_z, tm4.z, rq_w * tm4.x));
                    const float isig = rq_iA * tm4.w;
                    const bool okscale = (isig >= 0x1p-20f) && (isig <= 0x1p100f);
                    const float xa = __builtin_ceilf((thrA + eb_t) * isig), xb = __builtin_ceilf((thrB - eb_t) * isig);
                    const bool oka = okscale && (xa == xa), okb = okscale && (xb == xb);
                    const int ia = oka ? (int)fminf(fmaxf(xa, -1073741760.f), 1073741824.f) : (1 << 30);
                    const int ib = okb ? (int)fminf(fmaxf(xb, -1073741760.f), 1073741824.f) : -1073741760;
                    *reinterpret_cast<int2*>(reinterpret_cast<char*>(thr_s) + (thr_dst + (uint32_t)(t & 1) * 2048u)) = make_int2(ia, ib);
                } else if constexpr (s == 7) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the thresholds are in LDS before this wave's next barrier
                }
            });
        } else {
            scrr_static_for<16>([&](auto rc) __attribute__((always_inline)) { row_slice(rc, t); });
        }
        // ---- the previous tile's undecided outputs go to the list
        if (t >= 1 && !(SCRP_ABLATE & 2)) {
            undm &= rowmask;
            if (et_prev + l31 >= e_end) undm &= 0xAAAAAAAAu;
            if (et_prev + 32 + l31 >= e_end) undm &= 0x55555555u;
            if (__popcll(__ballot(undm != 0u)) <= SCRR_PEND / 32) append(undm, et_prev);
            else for (int ps = 0; ps < 8; ++ps) append(undm & (0xFu << (4 * ps)), et_prev);
        }
        undm = 0u;
        if ((t & 7) == 7) count_masks(16);
        if (t < (int)ntile) {
            // ---- g = (the partner's share) + L1b + (L2b >> 8); candidates beyond the range: the sentinel
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(bcc[0][0]), "+a"(bcc[0][1]), "+a"(bcc[1][0]), "+a"(bcc[1][1]) :: "memory");
            const int64_t et = e_begin + (int64_t)t * SCR_ET;
            const int4* const hand = reinterpret_cast<const int4*>(hand_s + (size_t)w * SCRP_HAND_BYTES) + lane;
#pragma unroll
            for (int jj = 0; jj < ((SCRP_ABLATE & 2) ? 1 : 8); ++jj) {
                const int4 hv = hand[jj * 64];
                const int v[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int idx = 4 * jj + c, r = idx >> 1, ni = idx & 1;
                    const int g = (int)((uint32_t)v[c] + (uint32_t)bcc[0][ni][r] + (uint32_t)(bcc[1][ni][r] >> 8));
                    if (ni) G1[r] = g; else G0[r] = g;
                }
            }
            if (et + SCR_ET > e_end) {   // (the range's last tile)
                const bool v0 = et + l31 < e_end, v1 = et + 32 + l31 < e_end;
#pragma unroll
                for (int r = 0; r < 16; ++r) { G0[r] = v0 ? G0[r] : SCRR_NONE; G1[r] = v1 ? G1[r] : SCRR_NONE; }
            }
            et_prev = et;
            th = thr_s[(t & 1) * 128 + row0];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (npend) flush();
    count_masks(2 * (((int)ntile + 1) & 7));   // the outputs gathered since the last count (the masks start at zero: the upper bits are clear)
}

}  // namespace kge
