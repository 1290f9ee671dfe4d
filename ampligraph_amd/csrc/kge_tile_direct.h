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
// Here a workgroup is ONE wave group (gw waves, every lane one quad of each half of the row) and owns one tile's bucket:
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

constexpr int DIRECT_UN = 4;          // entries whose operand rows are in flight together
constexpr int DIRECT_OVF_SLACK = 256; // overflow-list entries of this tile the LDS list has room for beyond the bucket capacity

__host__ __device__ inline size_t direct_lds_bytes(int cap, int tile_rows) {
    const size_t n = (size_t)cap + DIRECT_OVF_SLACK;
    return n * 16 + n * 2 + (size_t)(tile_rows + 2) * 4 * 2 + 64;
}

template <int MODEL, int GW>
__global__ __launch_bounds__(GW * 64) void tile_direct_kernel(TileArgs a) {
    using T = ModelTraits<MODEL>;
    constexpr int NC = T::NC;
    constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
    constexpr int THREADS = GW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last, s_total;

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
    const StageEntry* list = a.lists + (size_t)tile * a.cap;
    if (tid == 0) s_total = cnt;
    for (int r = tid; r <= nrow; r += THREADS) { rstart[r] = 0; rfill[r] = 0; }
    for (int i = tid; i < cnt; i += THREADS) ents[i] = reinterpret_cast<const uint4*>(list)[i];
    __syncthreads();
    for (int base = 0; base < on; base += THREADS) {   // entries of buckets that were full: every tile filters the shared list
        uint4 e = make_uint4(0, 0, 0, 0xFFFFFFFFu);
        if (base + tid < on) e = reinterpret_cast<const uint4*>(a.ovf)[base + tid];
        if (e.w != 0xFFFFFFFFu && (e.w / RB) % NT == (uint32_t)tile) {
            const int at = atomicAdd(&s_total, 1);
            if (at < lcap) ents[at] = e;
        }
    }
    __syncthreads();
    int total = s_total;
    if (total > lcap) {   // more entries than the LDS list holds (a pathologically hot tile): flagged, the host raises
        if (tid == 0) atomicExch(a.status_flag, 2);
        total = lcap;
    }
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
    __syncthreads();
    // (within a row the order of its entries is their arrival order in the scatter above: fp32 summation order only)

    // ---- 2. rows, one at a time, through registers --------------------------------------------------------------------
    const int q = lane + 64 * wg;
    const bool qok = q < a.nq;
    const int qoff = (qok ? q : 0) * 4;
    float reg_acc = 0.f;
    {
        for (int r = 0; r < nrow; ++r) {
            const int64_t row = row_of(r);
            if (row >= a.n_rows) continue;
            const int e0 = rstart[r], e1 = rstart[r + 1];
            const int hot = a.hot_map ? a.hot_map[row] : 0;
            const bool marked = a.touched && a.touched[row];
            if (a.lazy && a.apply_update && !hot && !marked && e0 == e1) continue;   // untouched row: keeps its bits
            const int64_t off = row * a.K + qoff;
            float4 x[NC], m[NC], v[NC], g[NC];
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                g[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                m[h] = g[h]; v[h] = g[h];
                x[h] = KGE_LD4(a.x + off + h * a.k);   // the live row: operand of its corruption entries AND of the update
                if (a.apply_update) {   // (the slot pointers an update rule does not use are NULL)
                    if (a.s0) m[h] = KGE_LD4(a.s0 + off + h * a.k);
                    if (a.s1) v[h] = KGE_LD4(a.s1 + off + h * a.k);
                }
            }
            for (int i0 = e0; i0 < e1; i0 += DIRECT_UN) {
                uint32_t meta[DIRECT_UN];
                float gg[DIRECT_UN];
                float4 sv[DIRECT_UN][NC], pv[DIRECT_UN][NC];
#pragma unroll
                for (int u = 0; u < DIRECT_UN; ++u) {
                    if (i0 + u < e1) {
                        const uint4 e = ents[order[i0 + u]];   // same address in every lane: LDS broadcast, then scalars
                        const uint32_t pos = __builtin_amdgcn_readfirstlane(e.x);
                        meta[u] = __builtin_amdgcn_readfirstlane(e.y);
                        gg[u] = __uint_as_float(__builtin_amdgcn_readfirstlane(e.z));
                        const int role = meta[u] & 3;   // 0 / 1: corruption with object / subject replaced; 2 / 3: the positive's own s / o row
                        const int which = (role == 0) ? 2 : (role == 1) ? 3 : (role == 2) ? 0 : 1;
                        const float* src = a.stage_rows + ((int64_t)pos * a.ns + which) * a.K + qoff;
#pragma unroll
                        for (int h = 0; h < NC; ++h) sv[u][h] = KGE_LD4(src + h * a.k);
                        if constexpr (MODEL == AMDKGE_TRANSE) {
                            if (role < 2) pv[u][0] = KGE_LD4(a.rel + (int64_t)a.triples[3 * (int64_t)pos + 1] * a.K + qoff);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < DIRECT_UN; ++u) {
                    if (i0 + u >= e1) continue;
                    const int role = meta[u] & 3;
                    const float ge = gg[u];
                    if (TRILINEAR || role >= 2) {
#pragma unroll
                        for (int h = 0; h < NC; ++h) {
                            g[h].x += ge * sv[u][h].x; g[h].y += ge * sv[u][h].y; g[h].z += ge * sv[u][h].z; g[h].w += ge * sv[u][h].w;
                        }
                    } else if constexpr (MODEL == AMDKGE_ROTATE) {
                        // g (e - S) / |e - S|, S the staged side row (A = s o r, or B = o o conj(r)), e this row (see add_entry)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float dr = (&x[0].x)[c] - (&sv[u][0].x)[c], di = (&x[NC - 1].x)[c] - (&sv[u][NC - 1].x)[c];
                            const float mm = KGE_SQRT(dr * dr + di * di) + ((qoff + c >= a.k_live) ? 1.f : 0.f);
                            const float gm = KGE_DIV(ge, mm);
                            (&g[0].x)[c] += gm * dr;
                            (&g[NC - 1].x)[c] += gm * di;
                        }
                    } else if constexpr (MODEL == AMDKGE_TRANSE) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float p[1] = {(&pv[u][0].x)[c]}, e[1] = {(&x[0].x)[c]}, sd[1] = {(&sv[u][0].x)[c]}, ds[1], dp[1], dd[1];
                            if (role == 0) grad_unit<AMDKGE_TRANSE>(sd, p, e, ge, ds, dp, dd);
                            else grad_unit<AMDKGE_TRANSE>(e, p, sd, ge, ds, dp, dd);
                            (&g[0].x)[c] += (role == 0) ? dd[0] : ds[0];
                        }
                    }
                }
            }
            if (!qok) continue;
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
                    opt_elem<KIND>(a.opt, x[h].x, g[h].x, m[h].x, v[h].x, reg_acc); opt_elem<KIND>(a.opt, x[h].y, g[h].y, m[h].y, v[h].y, reg_acc); \
                    opt_elem<KIND>(a.opt, x[h].z, g[h].z, m[h].z, v[h].z, reg_acc); opt_elem<KIND>(a.opt, x[h].w, g[h].w, m[h].w, v[h].w, reg_acc); \
                    if constexpr (opt_nslots(KIND) >= 1) *reinterpret_cast<float4*>(a.s0 + off + h * a.k) = m[h]; \
                    if constexpr (opt_nslots(KIND) == 2) *reinterpret_cast<float4*>(a.s1 + off + h * a.k) = v[h]; } while (0)
                KGE_OPT_DISPATCH(a.opt.kind, KGE_UPD)
#undef KGE_UPD
                *reinterpret_cast<float4*>(a.x + off + h * a.k) = x[h];
            }
        }
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
