// NOT PART OF THE PRODUCT (moved out of ampligraph_amd/csrc/kge_rank_screen.h in round 6): round 5's second form of the
// int8 screening kernel -- both operands REGISTER-staged through LDS, two stages ahead.  Measured slower than the shipped
// rank_screen_kernel_v1 (C2 both sides 2.53 vs 2.33 ms, C3 1.15 vs 1.08; SQ_WAVE_CYCLES 980 M vs 862 M per launch for the same 22.7 M
// matrix instructions: profiles/r05d_*).  Kept as a record of the experiment; to build it again, paste it back behind
// rank_screen_kernel_v1 (it uses that header's ScreenArgs, SCR_* constants and epilogue conventions) and launch it with
// SCR_LDS_BYTES_Q of dynamic LDS.  Its successor is the LDS-DMA form, ampligraph_amd/csrc/kge_rank_screen_g.h.

// Round 5, the variant AMDKGE_SCREEN_KERNEL=2 selects -- NOT the default: it measured 8 % slower (see run_screen in
// kge_rank.hip).  BOTH operands through LDS, both two stages ahead.  The hypothesis: in rank_screen_kernel_v1 above a wave's query
// fragments come straight from L2 into registers ONE stage ahead -- a third register set does not fit the 256 a wave has at two waves
// per SIMD -- and a stage (384 cycles of matrix work) lasts ~2 000 cycles: as long as that round trip?  The measurement says no:
// with the query loads off the critical path a wave-stage takes 2 070 cycles instead of 1 820 (SQ_WAVE_CYCLES per launch 980 M vs
// 862 M for the same 22.7 M matrix instructions): the three extra LDS reads and stores per stage cost more than the L2 wait they
// replace.  Here the
// workgroup's four query blocks of a slab (12 KB) travel like the entity slab: requested two stages ahead into one register
// set per parity (the 24 registers the fragment double-buffer held), parked in the other LDS buffer at the end of the next
// stage, and a wave reads its three limb fragments from LDS right before the matrix instructions that use them (the second and
// third into the registers the finished entity fragments leave).  No vector memory instruction sits on a stage's critical path.
constexpr size_t SCR_LDS_BYTES_Q = SCR_LDS_BYTES + (size_t)2 * 4 * 3 * 2 * 32 * 16;   // + the query slabs, double-buffered (24 KB)
__global__ __launch_bounds__(SCR_THREADS, 2) void rank_screen_kernel(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];
    typedef uint4 (*slab_t)[2][3][2][32];   // [buffer][entity block][limb][half][row]: a buffer is the two blocks' slabs back to back
    slab_t Es = reinterpret_cast<slab_t>(smem_scr);
    float4* qm_s = reinterpret_cast<float4*>(smem_scr + (size_t)2 * 3 * 2 * SCR_ET * 16);
    float4* qt_s = qm_s + 128;
    float4* em_s = qt_s + 128;
    typedef uint4 (*qslab_t)[4][3][2][32];   // [buffer][query block][limb][half][row]: the four blocks' slabs back to back, as in memory
    qslab_t Qs = reinterpret_cast<qslab_t>(smem_scr + SCR_LDS_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wq = wv * 32;
    int bx, by;   // XCD-aware work order, as rank_count_mfma_kernel
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

    if (tid < 128) {
        // per query row, the constants of the error bound pre-combined and inflated by c = 1 + 2^-10 (covers the epilogue's own
        // roundings of the bound):  {2^16 A,  c gamma |q|_2,  c A,  c (|q|_1 / 2 + drop A)}
        const bool okq = q0 + tid < a.n;
        const float4 m4 = a.b.qm[okq ? q0 + tid : a.n - 1];
        const float c = 1.f + 0x1p-10f;
        qm_s[tid] = make_float4(m4.x * 65536.f, m4.y * c, m4.x * c, fmaf(a.drop, m4.x, m4.z) * c);
        // The epilogue's own fp32 roundings.  Rebuilding f = L0 2^16 + L1 2^8 + L2 (an integer below 2^42) in fp32 costs at most
        // 2^11 absolutely (conversions of L1, L2 beyond 2^24 and the inner sum; part of `drop`, see run_screen) and 2^-24 |f| in the outer; the scales are powers of two;
        // S~ -+ E' rounds by 2^-24 (|S~| + E').  The part relative to |S~| (eps = 2^-23, taken as 2^-22) is moved into the
        // thresholds:  S~ - E' >= T + 2 eps |T|  implies  S~ - E' - eps |S~| >= T  (|S~| <= 2 (|T| + E'): directly, the 2^-10
        // inflation of E' taking the E' part; larger |S~|: its sign decides) -- 2^-20 |T| here, which also covers these sums' own
        // rounding.  {G: greater, L: smaller, EL / EH: equal}; non-finite thresholds (nothing can be greater / smaller) stay.
        const float2 t2 = a.b.qt[okq ? q0 + tid : a.n - 1];
        const float s1 = isfinite(t2.x) ? 0x1p-20f * fabsf(t2.x) : 0.f, s2 = isfinite(t2.y) ? 0x1p-20f * fabsf(t2.y) : 0.f;
        qt_s[tid] = make_float4(t2.y + s2, t2.x - s1, t2.x + s1, t2.y - s2);
    }
    // outputs of rows beyond n (per lane: bits 2 r, 2 r + 1 of its 16 rows) and, per tile, of candidates beyond the range are
    // cleared from the undecided mask; they cannot be counted either (see the -inf bias below)
    uint32_t rowmask = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rowmask |= (q0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < a.n) ? (3u << (2 * r)) : 0u;
    // this wave's query fragments: block (q0 + wq) / 32, one coalesced 1 KB read per limb and slab (rows beyond n: the stale tail
    // of the last block -- finite integers; their outputs are masked by the thresholds above)
    // Addresses are a wave-uniform base (scalar registers, advanced once per stage) plus a per-lane byte offset fixed for the
    // whole kernel: no 64-bit vector address arithmetic inside the stage loop.
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    // query slab loader: the workgroup's four 32-row blocks of slab s are four 3 072-byte runs, S slabs apart; thread tid copies the
    // 16-byte pieces tid, tid + 256, tid + 512 of their concatenation (12 288 bytes) -- the LDS image is that concatenation
    const char* const qbase = reinterpret_cast<const char*>(a.b.qlimbs) + (q0 >> 5) * (int64_t)S * SCR_BLK_SLAB;
    // entity slab loader: a stage's LDS image is the two 32-row blocks' 3 072-byte slabs back to back (6 144 bytes), exactly as
    // they lie in memory.  Every thread copies 16 bytes at 16 tid (the first 4 096) and 8 bytes at 4 096 + 8 tid: both loads of
    // ALL threads are unconditional (a predicated load leaves the compiler without a vmcnt it can count on -- it then waited for
    // loads it had just issued) and every instruction reads and writes one contiguous run.
    const uint32_t blk_stride = (uint32_t)S * SCR_BLK_SLAB;   // bytes between consecutive 32-row blocks (S <= 64 slabs: < 2^18)
    const uint32_t qo0 = (uint32_t)(tid / 192) * blk_stride + (uint32_t)(tid % 192) * 16u;
    const uint32_t qo1 = (uint32_t)((tid + 256) / 192) * blk_stride + (uint32_t)((tid + 256) % 192) * 16u;
    const uint32_t qo2 = (uint32_t)((tid + 512) / 192) * blk_stride + (uint32_t)((tid + 512) % 192) * 16u;
    uint4* const ldsQ = &Qs[0][0][0][0][0] + tid;
    const uint32_t offA = tid < 192 ? (uint32_t)tid * 16u : blk_stride + (uint32_t)(tid - 192) * 16u;
    const uint32_t offB = blk_stride + 1024u + (uint32_t)tid * 8u;
    const uint32_t emoff = (uint32_t)(tid & (SCR_ET - 1)) * 16u;
    uint4* const ldsA = &Es[0][0][0][0][0] + tid;
    uint2* const ldsB = reinterpret_cast<uint2*>(reinterpret_cast<char*>(&Es[0][0][0][0][0]) + 4096) + tid;
    const char* ebase = nullptr;   // slab ld_s of the tile's first block (et is a multiple of 64; blocks beyond the table's end are slack rows of the buffer)
    // The (tile, slab) sequence is ONE stream of positions g = 0 .. ntile S - 1, software-pipelined two deep on the entity side:
    // at position g the global loads of position g + 2 are issued (register set g % 2), the set loaded during g - 1 (position
    // g + 1) goes to the other LDS buffer at the end, and the query fragments of g + 1 are requested for the next position --
    // a round trip to L2 / the Infinity Cache (~2 000 cycles) is covered by two stages of matrix work of both resident
    // workgroups instead of stalling every stage (measured: 2 950 cycles per stage with a one-deep pipeline against 385 of MFMA).
    uint4 eA0 = make_uint4(0, 0, 0, 0), eA1 = eA0;   // register sets 0 / 1 x the thread's two pieces (scalars: an array would live in scratch)
    uint2 eB0 = make_uint2(0, 0), eB1 = eB0;
    int ld_s = 0;
    int64_t ld_tile = 0;
    auto set_src = [&](int64_t et) { ebase = reinterpret_cast<const char*>(a.b.elimbs) + (et >> 5) * (int64_t)blk_stride; };
    auto load_e = [&](uint4& pa, uint2& pb) {
        pa = *reinterpret_cast<const uint4*>(ebase + offA);
        pb = *reinterpret_cast<const uint2*>(ebase + offB);
        ebase += SCR_BLK_SLAB;
        if (++ld_s == S) {
            ld_s = 0;
            ld_tile = ld_tile + 1 < ntile ? ld_tile + 1 : ntile - 1;   // (past the end: harmless re-reads of the last tile)
            set_src(e_begin + ld_tile * SCR_ET);
        }
    };
    auto store_e = [&](int buf, const uint4& pa, const uint2& pb) {
        ldsA[(size_t)buf * 384] = pa;
        ldsB[(size_t)buf * 768] = pb;
    };
    uint4 qa0 = make_uint4(0, 0, 0, 0), qb0 = qa0, qc0 = qa0, qa1 = qa0, qb1 = qa0, qc1 = qa0;   // register sets 0 / 1 x the thread's three pieces
    int ld_qs = 0;   // slab of the next query request (the query rows do not change with the entity tile: slabs 0 .. S - 1, round and round)
    auto load_q = [&](uint4& pa, uint4& pb, uint4& pc) {
        const char* const qs = qbase + (size_t)ld_qs * SCR_BLK_SLAB;
        pa = *reinterpret_cast<const uint4*>(qs + qo0);
        pb = *reinterpret_cast<const uint4*>(qs + qo1);
        pc = *reinterpret_cast<const uint4*>(qs + qo2);
        if (++ld_qs == S) ld_qs = 0;
    };
    auto store_q = [&](int buf, const uint4& pa, const uint4& pb, const uint4& pc) {
        ldsQ[(size_t)buf * 768] = pa;
        ldsQ[(size_t)buf * 768 + 256] = pb;
        ldsQ[(size_t)buf * 768 + 512] = pc;
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

    set_src(e_begin);
    load_e(eA0, eB0);        // position 0 -> LDS buffer 0
    load_q(qa0, qb0, qc0);
    load_e(eA1, eB1);        // position 1 -> register set 1
    load_q(qa1, qb1, qc1);
    store_e(0, eA0, eB0);
    store_q(0, qa0, qb0, qc0);
    __syncthreads();
    int st = 0;
    int t = 0;   // tile of the current position
    int npend = 0;   // pairs parked in this wave's LDS buffer (wave-uniform)
    int2* const pend = reinterpret_cast<int2*>(em_s + SCR_ET) + wv * SCR_PEND;
    // The list writes are issued as inline assembly and end with their own vmcnt(0): a store (or returning atomic) the compiler
    // knows about, pending next to the stage loop's prefetch loads, makes it give up counting vmcnt -- the first wait of every
    // stage became a vmcnt(0) on loads issued a moment earlier.  Memory operations it does not know about only make its waits
    // stricter (vmcnt retires in order), never wrong.
    auto flush = [&]() {
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
    auto append = [&](uint32_t msk, int64_t et) {   // park the marked outputs (bit 2 r + ni of a lane) of this wave; <= SCR_PEND of them
        const int mine = __popc(msk);
        int incl = mine;   // inclusive prefix over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int total = __shfl(incl, 63, 64);
        if (!total) return;
        if (npend + total > SCR_PEND) flush();
        int at = npend + incl - mine;
        while (msk) {
            const int bit = __builtin_ctz(msk);
            msk &= msk - 1;
            const int r = bit >> 1, ni = bit & 1;
            pend[at++] = make_int2((int)(q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * lh), (int)(et + ni * 32 + l31));
        }
        npend += total;
    };
    auto stage = [&](auto par_c) __attribute__((always_inline)) {
        constexpr int P = decltype(par_c)::value;   // g % 2: this position's LDS buffer and query set, the register set free for g + 2
        const int64_t et = e_begin + t * SCR_ET;
        // Loads of a stage, in THIS order and all unconditional (vmcnt retires in order and the compiler counts it): the tile's
        // candidate metas (used at stage 0 only), then the entity and query pieces of position g + 2.
        // (the metas of up to 63 candidates beyond the range are read: rows of later candidates or the head of the recheck list)
        float4 m4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.b.em + et) + emoff);
        if constexpr (P == 0) { load_e(eA0, eB0); load_q(qa0, qb0, qc0); } else { load_e(eA1, eB1); load_q(qa1, qb1, qc1); }
        __builtin_amdgcn_sched_barrier(0);
        // both entity blocks' fragments first, then the 12 matrix instructions ordered so that two of them on the SAME accumulator
        // are never adjacent (a dependent pair would wait out the first one's latency: twice its issue time)
        v4i32 eb[2][3];
        v4i32 qv;
        { const uint4 u = Qs[P][wv_s][0][lh][l31]; qv = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int lb = 0; lb < 3; ++lb) { const uint4 u = Es[P][ni][lb][lh][l31]; eb[ni][lb] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][0], acc[0][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][2], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][2], acc[2][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[0][1], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qv, eb[1][1], acc[1][1], 0, 0, 0);
        v4i32 qw;   // limb 1 of the query block (the entity fragments of limb 2 are finished: their registers are free)
        { const uint4 u = Qs[P][wv_s][1][lh][l31]; qw = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][1], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][1], acc[2][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][0], acc[1][1], 0, 0, 0);
        { const uint4 u = Qs[P][wv_s][2][lh][l31]; qw = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[0][0], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qw, eb[1][0], acc[2][1], 0, 0, 0);
        if constexpr (P == 0) { store_e(1, eA1, eB1); store_q(1, qa1, qb1, qc1); } else { store_e(0, eA0, eB0); store_q(0, qa0, qb0, qc0); }
        if (st == 0 && tid < SCR_ET) {   // candidates beyond the range: an infinite error bound -- never decided, never counted
            if (et + tid >= e_end) m4.y = INFINITY;
            em_s[tid] = m4;   // (read in the epilogue, behind this stage's barrier; the previous epilogue ended with one)
        }
        __syncthreads();
        if (++st == S && t < ntile) {
        // ---- epilogue: C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
            // Undecided outputs are only MARKED here (bit 2 r + ni of a per-lane mask); the appends to the recheck list happen once
            // per tile behind the loop: one atomic per wave instead of a ballot, a branch and an atomic per output.
            uint32_t undm = 0u;
            {
                // Fully unrolled over the lane's 16 rows, the two candidate columns of a row as packed fp32 pairs
                // (v_pk_fma / mul / add_f32), the next row's constants requested before this row's arithmetic; a scheduling
                // barrier per row keeps the compare masks (SGPR pairs) of one row from piling up behind those of all sixteen.
                const float4 E0 = em_s[l31], E1 = em_s[32 + l31];   // {B, |W e|_2 (infinite: inf / NaN row, candidate beyond the range), |e|_1 / 2, -}
                const f32x2 B2 = {E0.x, E1.x}, Y2 = {E0.y, E1.y}, Z2 = {E0.z, E1.z};
                const int row0 = wq + 4 * lh;
                float4 qm = qm_s[row0], qt = qt_s[row0];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float4 qm_n = qm, qt_n = qt;
                    if (r < 15) { const int rn = row0 + ((r + 1) & 3) + 8 * ((r + 1) >> 2); qm_n = qm_s[rn]; qt_n = qt_s[rn]; }
                    // S~ = (L0 2^16 + L1 2^8 + L2) 2^16 A B
                    const f32x2 c0 = {(float)acc[0][0][r], (float)acc[0][1][r]}, c1 = {(float)acc[1][0][r], (float)acc[1][1][r]},
                                c2 = {(float)acc[2][0][r], (float)acc[2][1][r]};
                    const f32x2 f = __builtin_elementwise_fma(c0, f32x2{65536.f, 65536.f}, __builtin_elementwise_fma(c1, f32x2{256.f, 256.f}, c2));
                    const f32x2 s0 = f * (f32x2{qm.x, qm.x} * B2);
                    // E' = c (gamma |W q|_2 |W e|_2 + A |e|_1 / 2 + B (|q|_1 / 2 + drop A)); the term relative to |S~| sits in the thresholds
                    const f32x2 e = __builtin_elementwise_fma(f32x2{qm.y, qm.y}, Y2, __builtin_elementwise_fma(f32x2{qm.z, qm.z}, Z2, f32x2{qm.w, qm.w} * B2));
                    const f32x2 lo = s0 - e, hi = s0 + e;
                    // greater: lo >= G;  smaller: hi < L;  equal after quantisation: lo >= EL and hi < EH (the quantisation bins are
                    // wide enough for that to settle a third of the near-ties)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const bool gt = lo[ni] >= qt.x, lt = hi[ni] < qt.y, eq = (lo[ni] >= qt.z) && (hi[ni] < qt.w);
                        cnt[r] += gt ? 1 : 0;
                        cnt[r] += eq ? 0x10000 : 0;
                        undm |= !(gt || lt || eq) ? (1u << (2 * r + ni)) : 0u;   // (NaN / infinite bounds compare false everywhere: undecided)
                    }
                    asm volatile("" : "+v"(cnt[r]), "+v"(undm));   // (the counts are formed HERE: left to itself the compiler keeps all 64 compare masks for later)
                    qm = qm_n; qt = qt_n;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            undm &= rowmask;
            if (et + l31 >= e_end) undm &= 0xAAAAAAAAu;        // candidate of block 0 beyond the range
            if (et + 32 + l31 >= e_end) undm &= 0x55555555u;   // candidate of block 1 beyond the range
            // Undecided pairs are parked in this wave's LDS buffer and go to the list when it is full (one returning atomic and
            // coalesced stores per flush): one atomic per wave and tile on the single counter -- 145 000 of them at C2 --
            // serialised at the L2 and cost more than the matrix work.
            if (__popcll(__ballot(undm != 0u)) <= SCR_PEND / 32) append(undm, et);   // (<= 32 outputs per lane)
            else for (int ps = 0; ps < 4; ++ps) append(undm & (0xFFu << (8 * ps)), et);   // (<= 8 per lane: 512 per wave)
#pragma unroll
            for (int lv = 0; lv < 3; ++lv)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[lv][ni][r] = 0;
            st = 0;
            ++t;
            __syncthreads();   // em_s is rewritten by the next tile
        }
    };
    // Always whole pairs of stages: with an odd number of positions the last pair's second stage multiplies re-read rows of the
    // last tile into accumulators nobody reads (its epilogue is guarded by t < ntile).  A conditional second stage gives the
    // loop a path on which the first stage's loads are still pending, and the compiler then waits for them on EVERY path.
    const int G = (int)(ntile * S);   // (< 2^31: a block's tiles x slabs)
    for (int g = 0; g < G; g += 2) {
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
    }
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

