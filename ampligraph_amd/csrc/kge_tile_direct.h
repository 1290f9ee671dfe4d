// Owner-computes tile pass for LONG rows (more than 128 quads per half: K > 2 KB, the C5 row width), "row direct" form.
// Included by kge_train_tiled.hip behind tile_backward_kernel, whose arguments (TileArgs), bucket layout and bookkeeping it shares.
//
// Why a second form.  tile_backward_kernel keeps a tile's gradient rows in LDS, adds every bucket entry to them and then
// flushes the rows through the optimizer.  With 8 KB rows a tile holds 18 rows, a CU runs ONE tile at a time (150 KB of LDS),
// and at one GPU's C5 shard (6.25 M rows, 65 536 positives x 66 entries: 0.7 entries per row) its phases -- zero the
// accumulators, a dozen entries, 18 row flushes one after the other per wave group -- leave the memory system idle most of the
// time: measured 3.4 TB/s (touched-rows mode) / 4.4 TB/s (dense) of L2<->fabric traffic against the 6.3 TB/s the same chip
// streams (profiles/r03a_c5_shard_kernels_and_pmc.json).  It also reads a touched row's x twice: once as the operand of its
// RotatE / TransE corruption entries, once more in the flush.
//
// Here a workgroup is ONE wave group (gw waves, every lane one quad of each half of the row) and owns one tile's bucket -- and a
// tile is LARGE (make_plan: ~150 rows instead of the 18 the LDS accumulators allowed; the pass keeps only an entry list in LDS):
// with 18-row tiles one step launched 347 000 workgroups whose start-up (bucket load, sort, bookkeeping) cost more than their
// rows did -- measured on the shard: tile budget 150 KB 61.0 / 83.1 ms per step (touched-rows / dense), 600 KB 41.2 / 65.9,
// 1 200 KB 40.4 / 65.3, 2 400 KB 41.3 / 67.8 (profiles/r03_c5_direct_variants.txt).
//   1. the bucket (+ this tile's share of the overflow list) is copied to LDS and counting-sorted by local row (a few dozen
//      16-byte entries: microseconds, and other workgroups of the CU cover it);
//   2. rows are taken one at a time: x, m, v of the row are requested, the row's entries are folded into a REGISTER
//      accumulator (operand rows of up to DIRECT_UN entries in flight), the update rule is applied and x, m, v are stored --
//      x is read once, nothing but the sorted entry list lives in LDS, so 5-6 workgroups share a CU and their row pipelines
//      overlap; rows without entries are skipped outright in touched-rows mode and get the plain (g = 0) update otherwise.
// Same arithmetic per entry and per element as tile_backward_kernel (add_entry / opt_elem): the sums differ only in the order
// in which a row's entries are added (sorted by local row, bucket order within a row), i.e. by fp32 summation order.
// Not used in deterministic mode (that one sorts by content in tile_backward_kernel) or when a bucket could outgrow the LDS list.
#pragma once

namespace kge {

constexpr int DIRECT_UN = 2;          // entries whose operand rows are in flight together (rows with more than DIRECT_PRE entries)
#ifndef KGE_DIRECT_PRE
#define KGE_DIRECT_PRE 0   // (measured at one GPU's C5 shard: 0 and 2 within noise once the tiles are large -- 40.4 / 65.3 ms vs 41.2 / 65.9)
#endif
constexpr int DIRECT_PRE = KGE_DIRECT_PRE;   // entries of the NEXT row whose operand rows are requested ahead, with its x / m / v
constexpr int DIRECT_PRE_N = DIRECT_PRE > 0 ? DIRECT_PRE : 1;   // (array extent)
constexpr int DIRECT_OVF_SLACK = 256; // overflow-list entries of this tile the LDS list has room for beyond the bucket capacity

__host__ __device__ inline size_t direct_lds_bytes(int cap, int tile_rows) {
    const size_t n = (size_t)cap + DIRECT_OVF_SLACK;
    return n * 16 + n * 2 + (size_t)(tile_rows + 2) * 4 * 2 + (size_t)tile_rows * 2 + 64;
}

template <int MODEL, int GW>
#ifdef KGE_DIRECT_WAVES   // development builds: force an occupancy
__attribute__((amdgpu_waves_per_eu(KGE_DIRECT_WAVES, KGE_DIRECT_WAVES)))
#endif
__global__ __launch_bounds__(GW * 64) void tile_direct_kernel(TileArgs a) {
    using T = ModelTraits<MODEL>;
    constexpr int NC = T::NC;
    constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
    constexpr int THREADS = GW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last, s_total, s_nact;

    const int tid = threadIdx.x, lane = tid & 63, wg = tid >> 6;
    const int tile = blockIdx.x;
    if (tile >= a.n_tiles) {   // relation table: ordinary dense sweep (its gradient was completed by the forward kernel's atomics)
        const int64_t first = (int64_t)(tile - a.n_tiles) * THREADS + tid, stride = (int64_t)a.rel_blocks * THREADS;
        float racc;
#define KGE_REL_SWEEP(KIND) racc = opt_sweep<KIND>(a.rel_opt, first, stride)
        KGE_OPT_DISPATCH(a.rel_opt.kind, KGE_REL_SWEEP)
#undef KGE_REL_SWEEP
        if (a.rel_opt.reg_loss && a.rel_opt.lam != 0.f) {
            const float w = wave_sum(racc);
            if (lane == 0) atomicAdd(a.rel_opt.reg_loss, (double)a.rel_opt.lam * (double)w);
        }
        return;
    }
    const int cnt = min(a.counters[tile * 32], a.cap);
    const int on = min(a.counters[a.n_tiles * 32], a.ovf_cap);
    if (tid == 0)   // the forward kernel's loss partials: folded by the first LOSS_PARTS tiles (see tile_backward_kernel)
        for (int sl = tile; sl < LOSS_PARTS; sl += a.n_tiles) {
            const unsigned long long old = atomicExch(reinterpret_cast<unsigned long long*>(a.loss_parts + (size_t)sl * LOSS_PART_STRIDE), 0ull);
            const double v = __longlong_as_double((long long)old);
            if (v != 0.0) atomicAdd(a.loss_sum, v);
        }
    const uint32_t NT = (uint32_t)a.n_tiles, RB = (uint32_t)a.rb;
    const int nrow = a.tile_rows;
    auto row_of = [&](int r) KGE_TILE_INLINE -> int64_t { return row_of_tile((uint32_t)tile, (uint32_t)r, NT, RB); };

    // ---- 1. bucket -> LDS, counting sort by local row ---------------------------------------------------------------
    const int lcap = a.cap + DIRECT_OVF_SLACK;
    uint4* ents = reinterpret_cast<uint4*>(smem);                                            // [lcap]
    uint16_t* order = reinterpret_cast<uint16_t*>(smem + (size_t)lcap * 16);                 // [lcap] entry indices, by row
    int* rstart = reinterpret_cast<int*>(smem + (((size_t)lcap * 18 + 15) & ~(size_t)15));   // [nrow + 1]
    int* rfill = rstart + nrow + 2;                                                          // [nrow + 1]
    uint16_t* active = reinterpret_cast<uint16_t*>(rfill + nrow + 2);                        // [nrow] rows this pass visits
    const StageEntry* list = a.lists + (size_t)tile * a.cap;
    if (tid == 0) { s_total = cnt; s_nact = 0; }
    for (int r = tid; r <= nrow; r += THREADS) { rstart[r] = 0; rfill[r] = 0; }
    for (int i = tid; i < cnt; i += THREADS) ents[i] = reinterpret_cast<const uint4*>(list)[i];
    __syncthreads();
    // Entries of buckets that were full (the shared overflow list): this tile's share joins the LDS list.  Should it not fit
    // (a pathologically hot tile: thousands of positives on one row without the hot-row replicas) the list keeps the bucket
    // only and every row also scans the overflow list in memory (ovf_slow below): slow, but complete.
    for (int base = 0; base < on; base += THREADS) {
        uint4 e = make_uint4(0, 0, 0, 0xFFFFFFFFu);
        if (base + tid < on) e = reinterpret_cast<const uint4*>(a.ovf)[base + tid];
        if (e.w != 0xFFFFFFFFu && (e.w / RB) % NT == (uint32_t)tile) {
            const int at = atomicAdd(&s_total, 1);
            if (at < lcap) ents[at] = e;
        }
    }
    __syncthreads();
    int total = s_total;
    const bool ovf_slow = total > lcap;
    if (ovf_slow) total = cnt;
    for (int i = tid; i < total; i += THREADS) atomicAdd(&rstart[entry_local(ents[i].y) + 1], 1);
    __syncthreads();
    if (wg == 0) {   // exclusive prefix over the rows (nrow is small: one wave, 64 rows per step)
        int carry = 0;
        for (int b = 0; b <= nrow; b += 64) {
            const int r = b + lane;
            int v = (r <= nrow) ? rstart[r] : 0, s = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
            if (r <= nrow) rstart[r] = carry + s;   // inclusive sum of the counts shifted by one == exclusive start
            carry += __shfl(s, 63, 64);
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += THREADS) {
        const int lr = (int)entry_local(ents[i].y);
        order[rstart[lr] + atomicAdd(&rfill[lr], 1)] = (uint16_t)i;
    }
    // the rows this pass visits, in row order: all of the tile's rows, or (touched-rows mode, in place) those with an entry, a
    // mark of the forward kernel's atomics or hot-row replicas
    if (wg == 0) {
        int n = 0;
        for (int b = 0; b < nrow; b += 64) {
            const int r = b + lane;
            bool take = false;
            if (r < nrow) {
                const int64_t row = row_of(r);
                take = row < a.n_rows;
                if (take && a.lazy && a.apply_update && !ovf_slow)
                    take = rstart[r + 1] > rstart[r] || (a.hot_map && a.hot_map[row]) || (a.touched && a.touched[row]);
            }
            const unsigned long long mk = __ballot(take);
            if (take) active[n + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0))] = (uint16_t)r;
            n += __popcll(mk);
        }
        if (lane == 0) s_nact = n;
    }
    __syncthreads();
    const int nact = s_nact;
    // (within a row the order of its entries is their arrival order in the scatter above: fp32 summation order only)

    // ---- 2. rows through registers, software-pipelined over the visited rows ------------------------------------------------
    // x, m, v of row i + 1 and the operand rows of its first DIRECT_PRE entries are REQUESTED before row i is computed and stored:
    // vmcnt retires in issue order, so a wave that loaded, computed and stored one row at a time waited for its own stores
    // before every load (measured: 13 us per row).  Issued ahead of the stores, the next row's loads are what the wave waits
    // for, with the stores still in flight behind them.
    const int q = lane + 64 * wg;
    const bool qok = q < a.nq;
    const int qoff = (qok ? q : 0) * 4;
    float reg_acc = 0.f;
    struct RowRegs {
        float4 x[NC], m[NC], v[NC];
        float4 sv[DIRECT_PRE_N][NC], pv[DIRECT_PRE_N];
        uint32_t meta[DIRECT_PRE_N];
        float gg[DIRECT_PRE_N];
        int e0, e1, hot, marked;
        int64_t off;
    };
    auto entry_loads = [&](int idx, uint32_t& meta, float& gg, float4 (&sv)[NC], float4& pv) KGE_TILE_INLINE {
        const uint4 e = ents[order[idx]];   // same address in every lane: LDS broadcast, then scalars
        const uint32_t pos = __builtin_amdgcn_readfirstlane(e.x);
        meta = __builtin_amdgcn_readfirstlane(e.y);
        gg = __uint_as_float(__builtin_amdgcn_readfirstlane(e.z));
        const int role = meta & 3;   // 0 / 1: corruption with object / subject replaced; 2 / 3: the positive's own s / o row
        const int which = (role == 0) ? 2 : (role == 1) ? 3 : (role == 2) ? 0 : 1;
        const float* src = a.stage_rows + ((int64_t)pos * a.ns + which) * a.K + qoff;
#pragma unroll
        for (int h = 0; h < NC; ++h) sv[h] = KGE_LD4(src + h * a.k);
        if constexpr (MODEL == AMDKGE_TRANSE) {
            if (role < 2) pv = KGE_LD4(a.rel + (int64_t)a.triples[3 * (int64_t)pos + 1] * a.K + qoff);
        }
    };
    auto request = [&](int i, RowRegs& R) KGE_TILE_INLINE {
        const int r = active[i];
        const int64_t row = row_of(r);
        R.e0 = rstart[r]; R.e1 = rstart[r + 1];
        R.hot = a.hot_map ? a.hot_map[row] : 0;
        R.marked = (a.touched && a.touched[row]) ? 1 : 0;
        R.off = row * a.K + qoff;
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            R.m[h] = make_float4(0.f, 0.f, 0.f, 0.f); R.v[h] = R.m[h];
            R.x[h] = KGE_LD4(a.x + R.off + h * a.k);   // the live row: operand of its corruption entries AND of the update
            if (a.apply_update) {   // (the slot pointers an update rule does not use are NULL)
                if (a.s0) R.m[h] = KGE_LD4(a.s0 + R.off + h * a.k);
                if (a.s1) R.v[h] = KGE_LD4(a.s1 + R.off + h * a.k);
            }
        }
#pragma unroll
        for (int u = 0; u < DIRECT_PRE; ++u)
            if (R.e0 + u < R.e1) entry_loads(R.e0 + u, R.meta[u], R.gg[u], R.sv[u], R.pv[u]);
    };
    auto fold = [&](uint32_t meta, float ge, const float4 (&sv)[NC], const float4& pv, const float4 (&x)[NC], float4 (&g)[NC]) KGE_TILE_INLINE {
        const int role = meta & 3;
        if (TRILINEAR || role >= 2) {
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                g[h].x += ge * sv[h].x; g[h].y += ge * sv[h].y; g[h].z += ge * sv[h].z; g[h].w += ge * sv[h].w;
            }
        } else if constexpr (MODEL == AMDKGE_ROTATE) {
            // g (e - S) / |e - S|, S the staged side row (A = s o r, or B = o o conj(r)), e this row (see add_entry)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float dr = (&x[0].x)[c] - (&sv[0].x)[c], di = (&x[NC - 1].x)[c] - (&sv[NC - 1].x)[c];
                const float mm = KGE_SQRT(dr * dr + di * di) + ((qoff + c >= a.k_live) ? 1.f : 0.f);
                const float gm = KGE_DIV(ge, mm);
                (&g[0].x)[c] += gm * dr;
                (&g[NC - 1].x)[c] += gm * di;
            }
        } else if constexpr (MODEL == AMDKGE_TRANSE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float p[1] = {(&pv.x)[c]}, e[1] = {(&x[0].x)[c]}, sd[1] = {(&sv[0].x)[c]}, ds[1], dp[1], dd[1];
                if (role == 0) grad_unit<AMDKGE_TRANSE>(sd, p, e, ge, ds, dp, dd);
                else grad_unit<AMDKGE_TRANSE>(e, p, sd, ge, ds, dp, dd);
                (&g[0].x)[c] += (role == 0) ? dd[0] : ds[0];
            }
        }
    };
    RowRegs cur, nxt;
    if (nact > 0) request(0, cur);
    for (int i = 0; i < nact; ++i) {
        if (i + 1 < nact) request(i + 1, nxt);
        float4 g[NC];
#pragma unroll
        for (int h = 0; h < NC; ++h) g[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < DIRECT_PRE; ++u)
            if (cur.e0 + u < cur.e1) fold(cur.meta[u], cur.gg[u], cur.sv[u], cur.pv[u], cur.x, g);
        for (int i0 = cur.e0 + DIRECT_PRE; i0 < cur.e1; i0 += DIRECT_UN) {   // rows with more entries than were requested ahead
            uint32_t meta[DIRECT_UN];
            float gg[DIRECT_UN];
            float4 sv[DIRECT_UN][NC], pv[DIRECT_UN];
#pragma unroll
            for (int u = 0; u < DIRECT_UN; ++u)
                if (i0 + u < cur.e1) entry_loads(i0 + u, meta[u], gg[u], sv[u], pv[u]);
#pragma unroll
            for (int u = 0; u < DIRECT_UN; ++u)
                if (i0 + u < cur.e1) fold(meta[u], gg[u], sv[u], pv[u], cur.x, g);
        }
        bool row_touched = cur.e1 > cur.e0 || cur.hot || cur.marked;
        if (ovf_slow) {   // the overflow list did not fit the LDS list: this row's entries are picked out of it in memory
            const uint32_t row32 = (uint32_t)((cur.off - qoff) / a.K);
            for (int base = 0; base < on; base += 64) {
                uint4 e = make_uint4(0, 0, 0, 0xFFFFFFFFu);
                if (base + lane < on) e = reinterpret_cast<const uint4*>(a.ovf)[base + lane];
                unsigned long long mk = __ballot(e.w == row32);
                row_touched = row_touched || mk != 0ull;
                while (mk) {
                    const int tt = __builtin_ctzll(mk);
                    mk &= mk - 1;
                    const uint32_t pos = __builtin_amdgcn_readlane(e.x, tt), meta = __builtin_amdgcn_readlane(e.y, tt);
                    const float ge = __uint_as_float(__builtin_amdgcn_readlane(e.z, tt));
                    const int role = meta & 3, which = (role == 0) ? 2 : (role == 1) ? 3 : (role == 2) ? 0 : 1;
                    const float* src = a.stage_rows + ((int64_t)pos * a.ns + which) * a.K + qoff;
                    float4 sv[NC], pv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int h = 0; h < NC; ++h) sv[h] = KGE_LD4(src + h * a.k);
                    if constexpr (MODEL == AMDKGE_TRANSE) {
                        if (role < 2) pv = KGE_LD4(a.rel + (int64_t)a.triples[3 * (int64_t)pos + 1] * a.K + qoff);
                    }
                    fold(meta, ge, sv, pv, cur.x, g);
                }
            }
        }
        // (ovf_slow visits every row of the tile: in touched-rows mode one without any entry keeps its bits)
        if (qok && !(a.lazy && a.apply_update && !row_touched)) {
            const int64_t off = cur.off;
            const int hot = cur.hot;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                float4* gp4 = (a.pos_atomic || !a.apply_update) ? reinterpret_cast<float4*>(a.g_ent + off + h * a.k) : nullptr;
                if (hot) {   // sum the replicas in fixed order and leave them zero for the next step
                    float4* hp = reinterpret_cast<float4*>(a.hot_buf + (int64_t)(hot - 1) * HOT_REPL * a.K + qoff + h * a.k);
#pragma unroll 4
                    for (int rp = 0; rp < HOT_REPL; ++rp) {
                        const float4 t = hp[(size_t)rp * (a.K >> 2)];
                        g[h].x += t.x; g[h].y += t.y; g[h].z += t.z; g[h].w += t.w;
                        hp[(size_t)rp * (a.K >> 2)] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (a.pos_atomic) {   // rows of the positives' own s / o, added by the forward kernel's atomics
                    const float4 gd = *gp4;
                    g[h].x += gd.x; g[h].y += gd.y; g[h].z += gd.z; g[h].w += gd.w;
                    if (a.apply_update) *gp4 = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (!a.apply_update) {
                    *gp4 = g[h];
                    continue;
                }
                // (the update rule is dispatched per row half: a wave-uniform switch, one compiled rule body each)
#define KGE_UPD(KIND) do { \
                    opt_elem<KIND>(a.opt, cur.x[h].x, g[h].x, cur.m[h].x, cur.v[h].x, reg_acc); opt_elem<KIND>(a.opt, cur.x[h].y, g[h].y, cur.m[h].y, cur.v[h].y, reg_acc); \
                    opt_elem<KIND>(a.opt, cur.x[h].z, g[h].z, cur.m[h].z, cur.v[h].z, reg_acc); opt_elem<KIND>(a.opt, cur.x[h].w, g[h].w, cur.m[h].w, cur.v[h].w, reg_acc); \
                    if constexpr (opt_nslots(KIND) >= 1) *reinterpret_cast<float4*>(a.s0 + off + h * a.k) = cur.m[h]; \
                    if constexpr (opt_nslots(KIND) == 2) *reinterpret_cast<float4*>(a.s1 + off + h * a.k) = cur.v[h]; } while (0)
                KGE_OPT_DISPATCH(a.opt.kind, KGE_UPD)
#undef KGE_UPD
                *reinterpret_cast<float4*>(a.x + off + h * a.k) = cur.x[h];
            }
        }
        cur = nxt;
    }
    if (a.apply_update && a.reg_loss && a.opt.lam != 0.f) {
        const float w = wave_sum(reg_acc);
        if (lane == 0) atomicAdd(a.loss_parts + (size_t)((tile * GW + wg) & (LOSS_PARTS - 1)) * LOSS_PART_STRIDE + 1, (double)a.opt.lam * (double)w);
    }
    // ---- 3. bookkeeping left zeroed for the next step (as tile_backward_kernel) -----------------------------------------
    __syncthreads();
    if (a.touched)
        for (int r = tid; r < nrow; r += THREADS)
            if (row_of(r) < a.n_rows) a.touched[row_of(r)] = 0;
    if (tid == 0) {
        a.counters[tile * 32] = 0;
        __threadfence();
        s_last = 0;
        if (atomicAdd(a.counters + (size_t)(a.n_tiles + 1) * 32, 1) == a.n_tiles - 1) {
            a.counters[a.n_tiles * 32] = 0;
            a.counters[(a.n_tiles + 1) * 32] = 0;
            s_last = 1;
        }
    }
    __syncthreads();
    if (s_last && wg == 0 && a.apply_update && a.reg_loss && a.opt.lam != 0.f) fold_loss_parts(a.loss_parts, a.reg_loss, lane, 1);
}

}  // namespace kge
