// NOT PART OF THE PRODUCT (round 6): measured no faster than rank_screen_kernel_v1 (2.119 vs 2.123 ms, MfmaUtil 0.3945 vs 0.3883,
// profiles/r06b_pmc_screen_g.json); was selectable as AMDKGE_SCREEN_KERNEL=3 in commits eba6079 .. a7dff44 (include it behind kge_rank_screen.h).
// Round 6: the screening kernel with its operand feed rebuilt around gfx950's LDS-DMA loads.
// Included by kge_rank.hip behind kge_rank_screen.h, whose limbs / thresholds / recheck / merge kernels, ScreenArgs and bound it shares:
// the matrix work (six limb products in three exact int32 accumulator levels per output) and the epilogue are those of
// rank_screen_kernel_v1 -- the counts are the same integers.
//
// What changed is how a stage's operands reach the matrix instructions.  rank_screen_kernel_v1 takes a wave's QUERY fragments from L2
// into registers one stage ahead and the entity slab through registers into LDS two stages ahead; a wave-stage lasts ~1 820 cycles for
// 384 cycles of matrix work (profiles/r04_pmc_screen.json: MfmaUtil 0.38), and neither fewer instructions of any class nor the
// register-staged "both operands through LDS" form of round 5 (AMDKGE_SCREEN_KERNEL=2: slower) moved it.  Here BOTH operands of a
// position travel global -> LDS by `global_load_lds_dwordx4` (no register staging, no ds_write pass, 36 VGPRs fewer) into a RING of
// three 18 KB stage buffers, two positions ahead; the limb layout is fragment-major, i.e. the LDS image of a stage IS its memory image
// in 1 KB pieces (64 lanes x 16 bytes: one DMA instruction per piece, lane-linear as the hardware requires).  Per stage a wave waits for
// its own five pieces of the position (`s_waitcnt vmcnt(5)`: the five of the next position stay in flight), meets the other three waves
// at ONE raw s_barrier (no vmcnt(0) drain: the compiler's __syncthreads would insert one), issues the five pieces of position g + 2
// into the buffer everyone has just finished reading, and reads its nine fragments.  Every vector-memory operation of the loop is such
// a DMA (the candidates' bound constants ride along as a piece of their own), so the counted wait is exact.
#pragma once

namespace kge {

constexpr int SCG_NB = 3;                          // stage buffers in the ring
constexpr int SCG_PIECE = 1024;                    // bytes per DMA instruction: 64 lanes x 16
constexpr int SCG_STAGE = 18 * SCG_PIECE;          // query blocks 0..3 (3 limbs each), entity blocks 0, 1: 18 pieces, as they lie in memory
constexpr int SCG_PEND = 256;                      // undecided pairs a wave parks in LDS before they go to the list
constexpr int SCG_EMB = 4;                         // candidate-meta buffers (by tile & 3)
constexpr size_t SCG_LDS_BYTES = (size_t)SCG_NB * SCG_STAGE + 2 * 128 * 16 + (size_t)SCG_EMB * SCR_ET * 16 + 4 * (size_t)SCG_PEND * 8;   // 71 680: two workgroups per CU

// One LDS-DMA instruction: 64 lanes x 16 bytes from the per-lane global addresses `g` to LDS bytes [lds, lds + 1 024) (M0 = the
// wave-uniform LDS byte address; the hardware adds lane x 16).  Inline assembly on purpose: issued through
// __builtin_amdgcn_global_load_lds the compiler knows an LDS write is pending and drains the whole queue (s_waitcnt vmcnt(0)) before
// the next ds_read of ANY address -- the ring would never hold a position in flight (seen in the ISA of the first build of this
// kernel).  Unknown to the compiler, the pieces are counted by hand: every wave issues exactly five per stage and waits `vmcnt(5)`.
// (s_nop: an SALU write of M0 needs one wait state before an LDS-DMA instruction reads it.)
__device__ __forceinline__ void scg_dma16(const char* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

__global__ __launch_bounds__(SCR_THREADS, 2) void rank_screen_kernel_g(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];   // (the ONE LDS object of the kernel: a second one makes hipcc drain the DMA queue before every ds_read)
    char* const ring = smem_scr;
    float4* const qm_s = reinterpret_cast<float4*>(smem_scr + (size_t)SCG_NB * SCG_STAGE);
    float4* const qt_s = qm_s + 128;
    float4* const em_s = qt_s + 128;   // [SCG_EMB][SCR_ET]

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
    const int S = a.b.S;
    const int64_t ntile = (e_end - e_begin + SCR_ET - 1) / SCR_ET;

    if (tid < 128) {   // per query row: the bound's constants and the four decision thresholds (see rank_screen_kernel_v1)
        const bool okq = q0 + tid < a.n;
        const float4 m4 = a.b.qm[okq ? q0 + tid : a.n - 1];
        const float c = 1.f + 0x1p-10f;
        qm_s[tid] = make_float4(m4.x * 65536.f, m4.y * c, m4.x * c, fmaf(a.drop, m4.x, m4.z) * c);
        const float2 t2 = a.b.qt[okq ? q0 + tid : a.n - 1];
        const float s1 = isfinite(t2.x) ? 0x1p-20f * fabsf(t2.x) : 0.f, s2 = isfinite(t2.y) ? 0x1p-20f * fabsf(t2.y) : 0.f;
        qt_s[tid] = make_float4(t2.y + s2, t2.x - s1, t2.x + s1, t2.y - s2);
    }
    uint32_t rowmask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rowmask |= (q0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < a.n) ? (3u << (2 * r)) : 0u;
    __syncthreads();   // (the last ordinary loads of the kernel are behind this barrier: from here on every VM operation is a DMA piece)

    // ---- the DMA schedule.  Piece p of a stage image: p < 12: query block p / 3, limb p % 3; p >= 12: entity block (p - 12) / 3, limb
    // (p - 12) % 3.  Wave w issues pieces w, w + 4, w + 8 (query), w + 12 (entity) and a fifth: waves 0, 1 the pieces 16, 17, waves 2, 3
    // the tile's 64 candidate metas (1 KB; both write the same bytes) -- five DMA instructions per wave and stage, all unconditional.
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const uint32_t blk_stride = (uint32_t)S * SCR_BLK_SLAB;   // bytes between consecutive 32-row blocks
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto piece_off = [&](int p) -> uint32_t {   // byte offset of piece p from its operand's (block 0, this slab) address
        const int q = p < 12 ? p : p - 12;
        return (uint32_t)(q / 3) * blk_stride + (uint32_t)(q % 3) * 1024u;
    };
    const uint32_t qo0 = piece_off(wv_s), qo1 = piece_off(wv_s + 4), qo2 = piece_off(wv_s + 8);
    const uint32_t eo0 = piece_off(wv_s + 12), eo1 = piece_off(16 + (wv_s & 1));
    const char* const qbase = reinterpret_cast<const char*>(a.b.qlimbs) + (q0 >> 5) * (int64_t)blk_stride + lane16;
    const char* ebase = nullptr;   // slab ld_s of the loading tile's first block (+ this lane's 16 bytes)
    const char* embase = nullptr;  // the loading tile's candidate metas
    int ld_s = 0, ld_qs = 0, ld_buf = 0;
    int64_t ld_tile = 0;
    auto set_src = [&](int64_t tile) {
        const int64_t et = e_begin + tile * SCR_ET;
        ebase = reinterpret_cast<const char*>(a.b.elimbs) + (et >> 5) * (int64_t)blk_stride + lane16;
        embase = reinterpret_cast<const char*>(a.b.em + et) + lane16;
    };
    // LDS byte addresses (the dynamic array is the kernel's only LDS object; its address-space-3 pointer IS the byte address)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_scr;
    const uint32_t lds_em = lds0 + (uint32_t)(SCG_NB * SCG_STAGE + 2 * 128 * 16);
    auto issue = [&]() {   // the five pieces of the next position into ring buffer ld_buf
        const uint32_t dst = lds0 + (uint32_t)ld_buf * SCG_STAGE + (uint32_t)wv_s * SCG_PIECE;
        const char* const qs = qbase + (size_t)ld_qs * SCR_BLK_SLAB;
        scg_dma16(qs + qo0, dst);
        scg_dma16(qs + qo1, dst + 4 * SCG_PIECE);
        scg_dma16(qs + qo2, dst + 8 * SCG_PIECE);
        scg_dma16(ebase + eo0, dst + 12 * SCG_PIECE);
        // (selects, not a branch: the fifth piece is entity piece 16 / 17 for waves 0, 1 and the tile's candidate metas for waves 2, 3)
        const bool low = wv_s < 2;
        scg_dma16(low ? ebase + eo1 : embase, low ? dst + 16 * SCG_PIECE : lds_em + (uint32_t)(ld_tile & (SCG_EMB - 1)) * (SCR_ET * 16));
        ld_buf = ld_buf + 1 == SCG_NB ? 0 : ld_buf + 1;
        if (++ld_qs == S) ld_qs = 0;
        ebase += SCR_BLK_SLAB;
        if (++ld_s == S) {
            ld_s = 0;
            ld_tile = ld_tile + 1 < ntile ? ld_tile + 1 : ntile - 1;   // (past the end: harmless re-reads of the last tile)
            set_src(ld_tile);
        }
    };

    int cnt[16];   // per accumulator register (= query row of this lane): greater | equal << 16
#pragma unroll
    for (int r = 0; r < 16; ++r) cnt[r] = 0;
    v16i32 acc[3][2];   // [level][entity block]: level 0 = l0 l0', 1 = l0 l1' + l1 l0', 2 = l0 l2' + l1 l1' + l2 l0'
#pragma unroll
    for (int lv = 0; lv < 3; ++lv)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[lv][ni][r] = 0;

    set_src(0);
    issue();   // position 0 -> buffer 0
    issue();   // position 1 -> buffer 1
    int st = 0, t = 0, buf = 0;
    int npend = 0;   // pairs parked in this wave's LDS buffer (wave-uniform)
    int2* const pend = reinterpret_cast<int2*>(em_s + SCG_EMB * SCR_ET) + wv * SCG_PEND;
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
    auto append = [&](uint32_t msk, int64_t et) {   // park the marked outputs (bit 2 r + ni of a lane) of this wave; <= SCG_PEND of them
        const int mine = __popc(msk);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int total = __shfl(incl, 63, 64);
        if (!total) return;
        if (npend + total > SCG_PEND) flush();
        int at = npend + incl - mine;
        while (msk) {
            const int bit = __builtin_ctz(msk);
            msk &= msk - 1;
            const int r = bit >> 1, ni = bit & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(et + ni * 32 + l31));
        }
        npend += total;
    };

    const int G = (int)(ntile * S);   // positions of this block: tiles x slabs (< 2^31)
    const uint32_t frag_off = (uint32_t)(lh * 512 + l31 * 16);   // this lane's 16 bytes inside a 1 KB piece: [half][row]
    for (int g = 0; g < G; ++g) {
        // this wave's five pieces of position g have landed (the five of g + 1 stay in flight) ...
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        // ... and so have the other waves'; everyone has also issued the matrix instructions that consumed buffer (g + 2) % 3 = (g - 1) % 3
        __builtin_amdgcn_s_barrier();
        issue();   // position g + 2
        const char* const sb = ring + (size_t)buf * SCG_STAGE + frag_off;
        const int64_t et = e_begin + (int64_t)t * SCR_ET;
        v4i32 eb[2][3];
        v4i32 qv;
        { const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)(3 * wv_s + 0) * SCG_PIECE); qv = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int lb = 0; lb < 3; ++lb) {
                const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)(12 + 3 * ni + lb) * SCG_PIECE);
                eb[ni][lb] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
            }
        // two matrix instructions on the SAME accumulator are never adjacent (a dependent pair waits out the first one's latency)
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][0], acc[0][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][2], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][2], acc[2][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][1], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][1], acc[1][1], 0, 0, 0);
        v4i32 qw;   // limb 1 of the query block (the entity fragments of limb 2 are finished: their registers are free)
        { const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)(3 * wv_s + 1) * SCG_PIECE); qw = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][1], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][1], acc[2][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][0], acc[1][1], 0, 0, 0);
        { const uint4 u = *reinterpret_cast<const uint4*>(sb + (size_t)(3 * wv_s + 2) * SCG_PIECE); qw = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][0], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][0], acc[2][1], 0, 0, 0);
        buf = buf + 1 == SCG_NB ? 0 : buf + 1;
        if (++st == S) {
            // ---- epilogue (rank_screen_kernel_v1's): C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
            uint32_t undm = 0u;
            {
                // the tile's candidate metas arrived with its first position (a DMA piece of waves 2, 3, behind that stage's barrier);
                // candidates beyond the range get an infinite bound here: never decided, never counted
                float4 E0 = em_s[(t & (SCG_EMB - 1)) * SCR_ET + l31], E1 = em_s[(t & (SCG_EMB - 1)) * SCR_ET + 32 + l31];
                if (et + l31 >= e_end) E0.y = INFINITY;
                if (et + 32 + l31 >= e_end) E1.y = INFINITY;
                const f32x2 B2 = {E0.x, E1.x}, Y2 = {E0.y, E1.y}, Z2 = {E0.z, E1.z};
                const int row0 = wq + 4 * lh;
                float4 qm = qm_s[row0], qt = qt_s[row0];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float4 qm_n = qm, qt_n = qt;
                    if (r < 15) { const int rn = row0 + ((r + 1) & 3) + 8 * ((r + 1) >> 2); qm_n = qm_s[rn]; qt_n = qt_s[rn]; }
                    const f32x2 c0 = {(float)acc[0][0][r], (float)acc[0][1][r]}, c1 = {(float)acc[1][0][r], (float)acc[1][1][r]},
                                c2 = {(float)acc[2][0][r], (float)acc[2][1][r]};
                    const f32x2 f = __builtin_elementwise_fma(c0, f32x2{65536.f, 65536.f}, __builtin_elementwise_fma(c1, f32x2{256.f, 256.f}, c2));
                    const f32x2 s0 = f * (f32x2{qm.x, qm.x} * B2);
                    const f32x2 e = __builtin_elementwise_fma(f32x2{qm.y, qm.y}, Y2, __builtin_elementwise_fma(f32x2{qm.z, qm.z}, Z2, f32x2{qm.w, qm.w} * B2));
                    const f32x2 lo = s0 - e, hi = s0 + e;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const bool gt = lo[ni] >= qt.x, lt = hi[ni] < qt.y, eq = (lo[ni] >= qt.z) && (hi[ni] < qt.w);
                        cnt[r] += gt ? 1 : 0;
                        cnt[r] += eq ? 0x10000 : 0;
                        undm |= !(gt || lt || eq) ? (1u << (2 * r + ni)) : 0u;   // (NaN / infinite bounds compare false everywhere: undecided)
                    }
                    asm volatile("" : "+v"(cnt[r]), "+v"(undm));
                    qm = qm_n; qt = qt_n;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            undm &= rowmask;
            if (et + l31 >= e_end) undm &= 0xAAAAAAAAu;        // candidate of block 0 beyond the range
            if (et + 32 + l31 >= e_end) undm &= 0x55555555u;   // candidate of block 1 beyond the range
            if (__popcll(__ballot(undm != 0u)) <= SCG_PEND / 32) append(undm, et);   // (<= 32 outputs per lane)
            else for (int ps = 0; ps < 8; ++ps) append(undm & (0xFu << (4 * ps)), et);   // (<= 4 per lane: 256 per wave)
#pragma unroll
            for (int lv = 0; lv < 3; ++lv)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[lv][ni][r] = 0;
            st = 0;
            ++t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two positions requested past the end: their DMA must not outlive the workgroup's LDS
    if (npend) flush();
    // ---- per query row: sum over the 32 lanes that share it ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int g = cnt[r] & 0xFFFF, e = cnt[r] >> 16;
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
