// NOT IN THE PRODUCT LIBRARY (round 6, measured and dropped): 844 us against r's 818 - 838 on the same box -- 280 instructions fewer per
// tile (26 scalar instead of 275, 41 s_waitcnt instead of 68) bought nothing: scalar work and waits are free beside matrix instructions.
// Its recheck list also differed from r's (252 524 against 257 556 pairs): never debugged; probably the unpadded VALU-write ->
// matrix-instruction-read pattern that later explained two other variants (kge_rank_screen_r.h, scripts/check_mfma_hazards.py) -- the copies
// behind its two PAIR bodies are where the allocator would have put them.  Kept for the record of the idea (ring buffer = slab index:
// every LDS address of the loop an immediate); profiles/r06x2_screen_variants.txt.  To build it again: include behind kge_rank_screen_r.h
// and launch rank_screen_kernel_s<13> with ScrsLds<13>::bytes of dynamic LDS (git show 214863e^ has the wiring).
//
// Round 6, second step: rank_screen_kernel_r with its instruction stream cut down to what ONE wave can issue.
// Included by kge_rank.hip behind kge_rank_screen_r.h (whose static_for and constants it uses).  Same matrix work, same decisions, same
// counts as kernels v1 / r.
//
// Why.  One wave per SIMD hides about five single-issue instructions in the 32 cycles a v_mfma_i32_32x32x32_i8 occupies the pipe
// (/opt/skills/guides/MI355X_MICROARCH.md; profiles/r06_mfma_filler_probe.txt).  Kernel r's tile body holds 1 033 instructions for
// 156 matrix instructions -- 6.6 per gap -- and a third of them are not work but bookkeeping the compiler cannot fold: 275 scalar
// operations (ring-buffer index arithmetic modulo 8, clamps, the parity of the second DMA piece), 68 s_waitcnt it must put in front
// of every operand it saw an LDS read for, 36 s_nop, 31 branches.  Arranging the SAME instructions differently does not move the
// kernel (slices spread at three per slot, slow path out of line: 812 - 838 us against 812 - 866, profiles/r06x1_*); only fewer of
// them can.  So:
//   * the LDS ring has S stage buffers, one per slab of a tile: position (tile t, slab s) always lives in buffer s, every LDS
//     address of the loop (DMA target, fragment reads) is an immediate, the global source address is one 64-bit add per stage
//     (positions past the block's end run into the limb array's padding -- two tiles of it -- instead of being clamped);
//   * which wave issues the half piece of a position is decided by the parity of the SLAB (waves 0, 1: even slabs, waves 2, 3: odd
//     slabs) and the tile loop is instantiated once per wave pair -- no run-time parity, the counted vmcnt of every stage a constant;
//   * fragment reads are inline assembly (the compiler does not track them) behind ONE s_waitcnt lgkmcnt(0) per stage, at its last
//     slot, three slots behind the newest read;
//   * the tile metas come from LDS (staged once per block): a vector-memory load in the loop shares vmcnt with the DMA pieces and
//     the compiler's wait for it, vmcnt(0), drained the ring once per tile;
//   * the slices are spread at <= 3 operations per slot and the undecided-output path is out of line.
#pragma once

#ifndef SCRS_ABLATE
#define SCRS_ABLATE 0   // development (scripts/build_variant.sh with EXTRA=-DSCRS_ABLATE=n): 1 no epilogue slices, 2 no matrix instructions, 4 no stage barrier, 8 no fold of the accumulators, 16 no DMA, 32 no fragment reads -- wrong counts, timing only
#endif

namespace kge {

constexpr int SCRS_D = 6;   // positions in flight ahead of the one being multiplied
template <int S> struct ScrsLds {
    static constexpr size_t ring = (size_t)S * SCRR_STAGE;                  // one stage buffer per slab
    static constexpr size_t thr = ring;                                     // [2][128] int4: (query row, tile)'s integer thresholds, by tile parity
    static constexpr size_t pend = thr + 2 * 128 * 16;                      // [4][SCRR_PEND] int2: parked pairs
    static constexpr size_t tm = pend + 4 * (size_t)SCRR_PEND * 8;          // [SCRR_TMCAP] float4: tile metas
    static constexpr size_t bytes = tm + (size_t)SCRR_TMCAP * 16;           // S = 13: 108 544
};

// One LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS bytes [lds_base + OFF, + 1 024).  m0 is written by
// the scalar add itself (the wave-uniform base is loop-invariant, OFF an immediate); the s_nop is the wait state the hardware wants
// between a scalar write of m0 and its use.
template <int OFF>
__device__ __forceinline__ void scrs_dma16(const char* sbase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_base), "n"(OFF) : "memory", "m0", "scc");
}
// One fragment read the compiler does not track (the consumer is behind the stage's own s_waitcnt lgkmcnt(0)).
template <int OFF>
__device__ __forceinline__ void scrs_read_frag(uint32_t addr, v4i32& f) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}

// does wave pair PAIR issue a second piece for a position in slab sp (sp may run past S: the next tile's slabs)?
template <int S, int PAIR> constexpr bool scrs_second(int sp) { return ((sp % S) & 1) == PAIR; }
// a wave's DMA instructions for the positions k0 .. k1 slabs behind slab s
template <int S, int PAIR> constexpr int scrs_behind(int s, int k0, int k1) {
    int n = 0;
    for (int k = k0; k <= k1; ++k) n += 1 + (scrs_second<S, PAIR>(s + k) ? 1 : 0);
    return n;
}

template <int S>
__global__ __launch_bounds__(SCR_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void rank_screen_kernel_s(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];   // (the ONE LDS object of the kernel)
    using L = ScrsLds<S>;
    int4* const thr_s = reinterpret_cast<int4*>(smem_scr + L::thr);
    float4* const tm_s = reinterpret_cast<float4*>(smem_scr + L::tm);

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
    for (int r = 0; r < 16; ++r) rowmask |= (q0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < a.n) ? (3u << (2 * r)) : 0u;

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
    {   // {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t} of this block's tiles (ntile <= SCRR_TMCAP: run_screen's schedule)
        const float4* const src = a.b.tm + (e_begin >> 6);
        for (int i = tid; i < (int)ntile; i += SCR_THREADS) tm_s[i] = src[i];
    }
    __syncthreads();   // (the last ordinary loads of the kernel are behind this barrier: from here on every VM operation is a DMA piece)

    // per accumulator register (= query row of this lane): greater, equal; gmask / emask gather one "not greater" / "equal" bit per
    // output (32 = 16 tiles of two columns) before they are counted
    int cntg[16], cnte[16];
    uint32_t gmask[16], emask[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { cntg[r] = 0; cnte[r] = 0; gmask[r] = 0u; emask[r] = 0u; }
    v16i32 acc[3][2];   // [level][entity block]: level 0 = l0 l0', 1 = l0 l1' + l1 l0', 2 = l0 l2' + l1 l1' + l2 l0'

    int npend = 0;   // pairs parked in this wave's LDS buffer (wave-uniform)
    int2* const pend = reinterpret_cast<int2*>(smem_scr + L::pend) + wv * SCRR_PEND;
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
            const int r = bit >> 1, ni = bit & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(et + ni * 32 + l31));
        }
        npend += total;
    };

    // ---- the DMA schedule.  rank_limbs_tile_kernel lays the candidates' limbs out POSITION-major: [tile of 64][slab][block 0 / 1][limb]
    // [half][row % 32][16 bytes]: the 6 KB image of a (tile, slab) position is contiguous and consecutive positions follow each other.
    // Piece p of a position: entity block p / 3, limb p % 3.  Wave w issues piece w of EVERY position; pieces 4, 5 of a position in an
    // EVEN slab come from waves 0, 1, of one in an odd slab from waves 2, 3.  The loop below is instantiated per wave pair (PAIR = w >> 1),
    // so the number of this wave's instructions behind a position -- the counted vmcnt -- is a constant of the stage.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_scr;
    const uint32_t lds_w0 = lds0 + (uint32_t)wv_s * SCRR_PIECE;               // piece w of buffer 0
    const uint32_t lds_w1 = lds0 + (uint32_t)(4 + (wv_s & 1)) * SCRR_PIECE;   // piece 4 / 5 of buffer 0
    // the source of a piece = a scalar base (this block's first position + the piece: loop-invariant, made provably wave-uniform) + a
    // per-lane offset that advances by one position per stage (one vector add; a block's run stays far below 4 GB)
    auto uniform_ptr = [](const char* p) {
        const uint64_t u = (uint64_t)(uintptr_t)p;
        return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)u)));
    };
    const char* const blk_base = reinterpret_cast<const char*>(a.b.elimbs) + (e_begin >> 6) * (int64_t)(S * SCRR_STAGE);
    const char* const src_w0 = uniform_ptr(blk_base + (size_t)wv_s * SCRR_PIECE);
    const char* const src_w1 = uniform_ptr(blk_base + (size_t)(4 + (wv_s & 1)) * SCRR_PIECE);
    uint32_t pos_off = lane16;   // + the next position to request
    // this lane's 16 bytes inside a 1 KB piece: [half][row]; buffers 10 .. need a second base (ds_read's offset is 16 bits)
    const uint32_t frag_lo = lds0 + (uint32_t)(lh * 512 + l31 * 16);
    const uint32_t frag_hi = frag_lo + 10u * SCRR_STAGE;
    v4i32 eb[2][2][3];   // [stage parity][entity block][limb]: the fragments of the NEXT stage are read while this one multiplies

    int G0[16], G1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0[r] = SCRR_NONE; G1[r] = SCRR_NONE; }   // (the sentinel of "no output": smaller than everything, counted nowhere)
    int nearv = 0;
    int64_t et_prev = e_begin;
    const int row0 = wq + 4 * lh;
    int4 th = thr_s[row0], th_n = th;
    uint32_t undm = 0u;
    float eb_t = 0.f, isig = 0.f;   // (the threshold slices' temporaries)
    int sd4 = 0, sd1 = 0, sd2 = 0, seq = 0, sund = 0;   // (a row's differences in flight between its slots)

    // the query fragments live in the accumulation half of the register file from here on (the matrix instruction reads its A operand
    // there directly): the other half holds the accumulators, the entity fragments and the epilogue
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int lb = 0; lb < 3; ++lb) asm volatile("" : "+a"(qf[s][lb]));

    const int pair = wv_s >> 1;   // (wave-uniform: which copy of the prologue / tile body this wave runs)
    auto issue = [&](auto pair_c, auto sp_c) __attribute__((always_inline)) {   // this wave's piece(s) of the next position, which lives in slab / buffer SP
        constexpr int PAIR = decltype(pair_c)::value, SP = decltype(sp_c)::value;
        scrs_dma16<SP * SCRR_STAGE>(src_w0, pos_off, lds_w0);
        if constexpr (scrs_second<S, PAIR>(SP)) scrs_dma16<SP * SCRR_STAGE>(src_w1, pos_off, lds_w1);
        pos_off += SCRR_STAGE;
    };
    auto read_frags = [&](auto sp_c, auto p_c, v4i32& f) __attribute__((always_inline)) {   // piece P of buffer SP
        constexpr int SP = decltype(sp_c)::value, P = decltype(p_c)::value;
        if constexpr (SP < 10) scrs_read_frag<SP * SCRR_STAGE + P * SCRR_PIECE>(frag_lo, f);
        else scrs_read_frag<(SP - 10) * SCRR_STAGE + P * SCRR_PIECE>(frag_hi, f);
    };

    static_assert(SCRS_D >= 2 && SCRS_D + 2 <= S, "a position is requested into a buffer whose readers are two barriers behind");
    auto prologue = [&](auto pair_c) __attribute__((always_inline)) {
        constexpr int PAIR = decltype(pair_c)::value;
        scrr_static_for<SCRS_D>([&](auto ic) __attribute__((always_inline)) { issue(pair_c, std::integral_constant<int, decltype(ic)::value % S>{}); });   // positions 0 .. D - 1
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(scrs_behind<S, PAIR>(0, 1, SCRS_D - 1)) : "memory");   // position 0 has landed: this wave's pieces, and everyone's
    };
    if (pair) prologue(std::integral_constant<int, 1>{}); else prologue(std::integral_constant<int, 0>{});
    scrr_static_for<6>([&](auto pc) __attribute__((always_inline)) { constexpr int p = decltype(pc)::value; read_frags(std::integral_constant<int, 0>{}, pc, eb[0][p / 3][p % 3]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- the software pipeline (see rank_screen_kernel_r): iteration t issues the matrix instructions of tile t and, BETWEEN them,
    // the epilogue of tile t - 1 in slices; iteration ntile runs the slices alone.
    for (int t = 0; t <= (int)ntile; ++t) {
        const float4 tm4 = tm_s[t < (int)ntile ? t : (int)ntile - 1];   // {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t}: wave-uniform
        auto tile_body = [&](auto mm_c, auto pair_c) __attribute__((always_inline)) {
        constexpr bool MM = decltype(mm_c)::value;
        constexpr int PAIR = decltype(pair_c)::value;
        scrr_static_for<S>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            constexpr int P = s & 1;
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
                    if constexpr (SCRS_ABLATE & 2) asm volatile("" : "+v"(C) : "a"(A), "v"(Bm));
                    else if constexpr (s == 0 && m < 6) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(C) : "a"(A), "v"(Bm));
                    else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(C) : "a"(A), "v"(Bm));
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- the stage's own work, spread over its slots
                if constexpr (!MM) {
                } else if constexpr (m == 1) {
                    // this wave's pieces of the next position (slab s + 1) have landed -- the D - 2 positions behind it stay in flight -- and
                    // so have the other waves'; everyone is past the fragment reads of the buffer the DMA of this stage rewrites
                    if constexpr (SCRS_ABLATE & 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(scrs_behind<S, PAIR>(s, 2, SCRS_D - 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(scrs_behind<S, PAIR>(s, 2, SCRS_D - 1)) : "memory");
                } else if constexpr (m == 2) {
                    if constexpr (!(SCRS_ABLATE & 16)) issue(pair_c, std::integral_constant<int, (s + SCRS_D) % S>{});   // the position D slabs ahead
                } else if constexpr (m >= 3 && m <= 8) {
                    if constexpr (!(SCRS_ABLATE & 32)) read_frags(std::integral_constant<int, (s + 1) % S>{}, std::integral_constant<int, m - 3>{}, eb[P ^ 1][(m - 3) / 3][(m - 3) % 3]);
                } else if constexpr (m == 11) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next stage's fragments (requested >= 3 slots ago) are in their registers
                }
                // ---- slice k of the previous tile's epilogue: row k / 8 (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4
                // (lane >> 5)), step k % 8.  One row = two outputs (entity blocks 0 / 1), no scalar register in the chain: d4 = g - Gi,
                // d1 = g - Li, d2 = g - ELi, d3 = g - EHi (thresholds and g inside +-2^30: no overflow).  sign(d4) = "not greater" and
                // sign(~d2 & d3) = "equal" are shifted into the row's bit masks (counted every 16 tiles); sign(~d1 & d4 & ~equal) =
                // neither smaller, greater nor equal: undecided, marked by the out-of-line path of step 6.
                if constexpr (k < 128 && !(SCRS_ABLATE & 1)) {
                    constexpr int r = k >> 3, j = k & 7;
                    if constexpr (j == 0) {
                        if constexpr (r < 15) { constexpr int rn = ((r + 1) & 3) + 8 * ((r + 1) >> 2); th_n = thr_s[((t + 1) & 1) * 128 + row0 + rn]; }
                        sd4 = G0[r] - th.x; sd1 = G0[r] - th.y;
                        asm volatile("" : "+v"(sd4), "+v"(sd1));
                    } else if constexpr (j == 1) {
                        const int d2 = G0[r] - th.z, d3 = G0[r] - th.w;
                        seq = ~d2 & d3;
                        asm volatile("" : "+v"(seq));
                    } else if constexpr (j == 2) {
                        gmask[r] = __builtin_amdgcn_alignbit(gmask[r], (uint32_t)sd4, 31);
                        emask[r] = __builtin_amdgcn_alignbit(emask[r], (uint32_t)seq, 31);
                        sund = ~sd1 & sd4 & ~seq;
                        asm volatile("" : "+v"(gmask[r]), "+v"(emask[r]), "+v"(sund));
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
                        nearv = (~sd1 & sd4 & ~seq) | sund;
                        asm volatile("" : "+v"(emask[r]), "+v"(nearv));
                    } else if constexpr (j == 6) {
                        if (__builtin_expect(__ballot(nearv < 0) != 0ull, 0)) {   // rare (a fraction of a per cent of the outputs): wave-uniform
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                const int g = ni ? G1[r] : G0[r];
                                const bool near = (g >= th.y) && (g < th.x), eq = (g >= th.z) && (g < th.w);
                                undm |= (near && !eq) ? (1u << (2 * r + ni)) : 0u;
                            }
                        }
                    } else {
                        th = th_n;
                    }
                }
                // ---- this tile's thresholds, by the thread that owns the query row (free slots behind the slices)
                if constexpr (!MM) {
                } else if constexpr (k == 130) {
                    // E = c (gamma |W q|_2 max |W e|_2 + A max |e|_1 / 2 + B_t (|q|_1 / 2 + drop A)); 1 / sigma = 1 / (2^24 A) * 1 / B_t
                    eb_t = __builtin_fmaf(rq_y, tm4.y, __builtin_fmaf(rq_z, tm4.z, rq_w * tm4.x));
                    isig = rq_iA * tm4.w;
                } else if constexpr (k == 133) {
                    // ceil((T +- E) / sigma) as an int32 (|g| < 2^29.1 for S <= 13: thresholds beyond (-2^30 + 64, 2^30] clamp without changing a
                    // decision); NaN (inf - inf) and scales outside [2^-20, 2^100] (a product could leave fp32's range): nothing decided
                    const bool okscale = (isig >= 0x1p-20f) && (isig <= 0x1p100f);
                    const float xa = __builtin_ceilf((thrA + eb_t) * isig), xb = __builtin_ceilf((thrB - eb_t) * isig);
                    const bool oka = okscale && (xa == xa), okb = okscale && (xb == xb);
                    const int ia = oka ? (int)fminf(fmaxf(xa, -1073741760.f), 1073741824.f) : (1 << 30);      // (never >=)
                    const int ib = okb ? (int)fminf(fmaxf(xb, -1073741760.f), 1073741824.f) : -1073741760;    // (never <, but for the sentinel)
                    int2* const dst = reinterpret_cast<int2*>(thr_s + (t & 1) * 128 + (tid & 127)) + (tid >> 7);
                    *dst = make_int2(ia, ib);
                }
                // (the thresholds are in LDS before this wave's next barrier: stage S - 2's lgkmcnt(0) at slot 143 is behind the write)
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        };
        if (t < (int)ntile) {
            if (pair) tile_body(std::true_type{}, std::integral_constant<int, 1>{}); else tile_body(std::true_type{}, std::integral_constant<int, 0>{});
        } else tile_body(std::false_type{}, std::integral_constant<int, 0>{});
        // A VALU read of a matrix instruction's result wants up to 19 wait states behind it (8 / 16 passes), and the compiler cannot see
        // into the statements above.  The accumulators and the fragment set the copies below overwrite are operands of the pad: nothing
        // that reads or rewrites them moves in front of it.
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                       "+v"(eb[0][0][0]), "+v"(eb[0][0][1]), "+v"(eb[0][0][2]), "+v"(eb[0][1][0]), "+v"(eb[0][1][1]), "+v"(eb[0][1][2])
                     :: "memory");
        if constexpr (S & 1) {   // (an odd number of stages: the next tile's first fragments sit in the other parity)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int lb = 0; lb < 3; ++lb) eb[0][ni][lb] = eb[1][ni][lb];
        }
        // ---- the previous tile's undecided outputs go to the list
        if (t >= 1) {
            undm &= rowmask;
            if (et_prev + l31 >= e_end) undm &= 0xAAAAAAAAu;        // candidate of block 0 beyond the range
            if (et_prev + 32 + l31 >= e_end) undm &= 0x55555555u;   // candidate of block 1 beyond the range
            if (__builtin_expect(__ballot(undm != 0u) != 0ull, 0)) {
                if (__popcll(__ballot(undm != 0u)) <= SCRR_PEND / 32) append(undm, et_prev);   // (<= 32 outputs per lane)
                else for (int ps = 0; ps < 8; ++ps) append(undm & (0xFu << (4 * ps)), et_prev);   // (<= 4 per lane: 256 per wave)
            }
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
            for (int r = 0; r < ((SCRS_ABLATE & 8) ? 1 : 16); ++r) {
                G0[r] = (int)(((uint32_t)acc[0][0][r] << 8) + (uint32_t)acc[1][0][r] + (uint32_t)(acc[2][0][r] >> 8));
                G1[r] = (int)(((uint32_t)acc[0][1][r] << 8) + (uint32_t)acc[1][1][r] + (uint32_t)(acc[2][1][r] >> 8));
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
