// Owner-computes training step for gfx950 (no global atomics on the entity table).
//
// Why: on MI355X `global_atomic_add_f32` retires ~10.4 G cache-line operations/s chip-wide whatever the
// line's occupancy (scripts/atomic_bench.hip), which floors the atomic gradient scatter of
// kge_train.hip at 230 000 rows x 12.5 lines = 0.28 ms per C2 step, 3x the HBM time of the same bytes.
//
// How: two kernels per step.
//   F  train_fwdbwd_kernel<.., STAGE=true> (kge_train_kernel.h): the fused forward + loss + per-positive
//      backward.  Instead of scattering replacement-row gradients it APPENDS one 16-byte entry
//      {positive, role|local row, dL/dscore, row id} per corruption (and one per s / o row of the positive) to
//      the bucket of the table TILE that owns the destination row (one returning atomic per entry on a
//      128-byte-strided counter; full buckets spill to one shared overflow list), and stores per positive
//      four K-float rows with plain 16-byte coalesced stores: the complete gradient rows of its own s and o,
//      and two "side" rows A, B (trilinear models: A = d score/d o (s,p), B = d score/d s (p,o), which do not
//      depend on the replaced row; TransE: copies of the s and o rows as read by the forward pass, plus one packed
//      byte per unit and corruption with the sign of s + p - o; RotatE: A = s o r, B = o o conj(r)).
//      The relation-row gradient (237 hot rows at C2) keeps the atomic row-add into the dense relation
//      gradient buffer, which the ordinary sweep (kge_opt.hip) consumes.
//   T  tile_backward_kernel (this file): one workgroup OWNS a tile of entity rows and keeps their gradient
//      accumulators in LDS (<= 150 KB).  It walks its bucket (+ the overflow list), adds g * A|B (trilinear),
//      -/+ g with the stored sign bits (TransE; grad_unit on three rows where a unit is (near) zero) or
//      g (e - S) / |e - S| on the side row S and its own live row e (RotatE) -- every wave into the rows it
//      owns, plain LDS read-modify-writes --, and finally applies the optimizer + regulariser to its rows straight from LDS: the
//      entity gradient never exists in HBM and the entity table needs no separate optimizer sweep.
//      The side rows are staged copies because other tiles update their rows in place while this one
//      is still reading.
//
// Replaces the same reference code as kge_train.hip + kge_opt.hip: ScoringBasedEmbeddingModel.train_step
// (/root/reference/ampligraph/latent_features/models/ScoringBasedEmbeddingModel.py:370-429) including
// optimizer.minimize (optimizers.py:136-168) and the LP regulariser (regularizers.py:35-37).
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

// RotatE's modulus and its reciprocal in the TRAINING kernels use the hardware v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the
// correctly rounded libm sequences: the fused kernels are bound by exactly these on RotatE (measured 1.26x on the step);
// loss and gradients stay far inside the 1e-5 relative tolerance of the parity tests.  predict() (kge_score.hip) keeps the
// exact forms; the rank kernels have their own rank_sqrt.  Device functions are inlined per kernel, so the two variants of
// score_unit / grad_unit never meet at link time.
#define KGE_FAST_ROTATE 1

#include "kge_opt.h"
#include <type_traits>
#include "kge_train_kernel.h"

// the lambdas of the tile kernel capture its argument struct by reference: one of them left out of line puts the whole struct
// (and the operand arrays passed to it) into scratch memory -- measured 6x on the TransE instantiation
#ifndef KGE_TILE_INLINE
#define KGE_TILE_INLINE __attribute__((always_inline))
#endif

// 16-byte operand load of the entry loop, as a VALUE.  Written as plain `x = *reinterpret_cast<const float4*>(p)` the RotatE
// tile pass measured 130 us instead of 113: assigned through the reference, the compiler orders the loads of a batch against
// the operand arrays of the previous one (more s_waitcnt, fewer loads in flight).  Found by bisection.
#ifdef KGE_LD4_FN
namespace kge { __device__ __forceinline__ float4 ld4_value(const float* p) { return *reinterpret_cast<const float4*>(p); } }
#define KGE_LD4(p) kge::ld4_value(p)
#else
#define KGE_LD4(p) (false ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(p))
#endif

namespace kge {

constexpr int TILE_THREADS = 1024;
constexpr int TILE_WAVES = TILE_THREADS / 64;
constexpr int TILE_QCAP = 128;                                              // entries of a wave's LDS queue (tile_backward_kernel)
constexpr size_t TILE_QUEUE_BYTES = (size_t)TILE_WAVES * TILE_QCAP * 16;    // 32 KB behind the accumulators
// which instantiations of tile_backward_kernel collect their entries in the LDS queue first (see the kernel); CH = quads per lane
// (rows beyond 2 KB are bandwidth-bound chunk by chunk, and the queue's 32 KB of LDS would shrink their tiles: C5 5 % slower)
__host__ __device__ constexpr bool tile_queued(int model, int CH, int K) {
    return ((model == AMDKGE_TRANSE || model == AMDKGE_ROTATE) && K <= 512) || (model == AMDKGE_DISTMULT && CH == 1);
}
static inline int tile_ch_of(int nq) { return (nq <= 64 || nq > 128) ? 1 : 2; }   // the CH run_tiled picks for the tile kernel

struct TileArgs {
    float* x;                 // entity table (updated in place when g_out == NULL)
    float* s0;                // optimizer slots
    float* s1;
    float* g_ent;             // dense entity gradient buffer
    int apply_update;         // 1: optimizer applied from LDS; 0: g_ent receives the entity gradient (data parallel)
    int pos_atomic;           // g_ent holds the s / o rows of the positives (forward kernel's atomics): fold them in
    int ns;                   // staged rows per positive (4; 5 in deterministic mode)
    int det;                  // deterministic mode: the tile's entries are sorted into a canonical order before they are added
    int sort_cap;             // det: entries the LDS sort buffer holds (a multiple of 64, <= 8192)
    int pos_bits;             // det: bits the positives' indices need (the radix passes of the index sort)
    int own_cache;            // RotatE, queued form: the tile's own live rows are copied into LDS behind the accumulators (see make_plan)
    int lazy;                 // touched-rows optimizer mode (amdkge_opt.lazy): rows without an entry keep their bits
    const uint8_t* hot_map;   // AMDKGE_TILED_HOT_ROWS (see HOT_MAX in kge_train_kernel.h); NULL = off
    float* hot_buf;
    uint8_t* touched;         // lazy + pos_atomic: rows the forward kernel's atomics touched (read, then cleared here)
    const uint32_t* sign_codes;   // TransE: [B][eta][nq] packed sign bytes written by the forward kernel (see ENTRY_J_SHIFT in kge_train_kernel.h); NULL = off
    int eta;
    const float* rel;         // live relation table (TransE / RotatE side of the gradient)
    const float* rel_cs;      // RotatE: [R][cos(phase) || sin(phase)] of this step's relation table (rel_phase_kernel)
    const int32_t* triples;
    const float* stage_rows;  // [B][4][K]
    const StageEntry* lists;  // [n_tiles][cap]
    const StageEntry* ovf;    // overflow entries
    int* counters;            // [(n_tiles + 2) * 32]: bucket fills, overflow count, finished-tiles ticket
    int* status_flag;         // det-sort overflow flag: at a workspace offset that does not depend on the plan (sticky until queried)
    double* loss_parts;       // the forward kernel's per-block loss partials, folded into loss_sum by the last tile
    double* loss_sum;
    double* reg_loss;
    OptArgs rel_opt;          // fused relation-table sweep (rel_blocks > 0): blocks [n_tiles, n_tiles + rel_blocks)
    int rel_blocks;
    int64_t n_rows;
    int64_t n_rels;
    int k, K, nq;             // stored half width, floats per stored row, quads per half
    int k_live;               // the model's k (RotatE: units behind it are zero padding, see grad_unit)
    int tile_rows, n_tiles, cap, ovf_cap;
    int rb;                   // rows per ownership block (block-interleaved tiles)
    int direct;               // launched as tile_direct_kernel (kge_tile_direct.h)
    int gw;                   // waves that share one row (1: a wave covers the row; 4 / 8: long rows are split over a group of
                              // waves, each lane one quad), rows are owned by wave GROUPS: TILE_WAVES / gw owners per tile
    ModelConst mc;
    OptArgs opt;
#ifdef KGE_ABLATE
    int dbg;                  // development ablation build only: 1024 no flush, 2048 no accumulator zeroing, 4096 no bucket scan, 16384 no sort (det)
                              // (per-load switches in the entry loop were tried: they push its operand arrays to scratch)
#endif
};

// LDS accumulators: [tile_rows][K] in table layout.  ds_add_f32 turned out to be far too slow for this
// (measured 0.49 ms for the C2 tile pass: ~3.4 clocks per LDS float atomic per CU), so rows are PARTITIONED
// over the waves of the workgroup instead (local row % TILE_WAVES) and every wave updates its own rows with
// plain 16-byte LDS read-modify-writes; all waves scan the tile's whole bucket and pick their entries with a
// ballot.
// DET (RotatE in deterministic mode only): IEEE square root and division in the entry arithmetic (kge_device.h kge_sqrt_t)
template <int MODEL, int CH, int UNROLL, bool DET = false>
__global__ __launch_bounds__(TILE_THREADS) void tile_backward_kernel(TileArgs a) {
    using T = ModelTraits<MODEL>;
    constexpr int NC = T::NC;
    constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = reinterpret_cast<float*>(smem);

    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x;
    if (tile >= a.n_tiles) {
        // relation table: ordinary dense sweep (its gradient was completed by the forward kernel's atomics)
        const int64_t first = (int64_t)(tile - a.n_tiles) * TILE_THREADS + tid, stride = (int64_t)a.rel_blocks * TILE_THREADS;
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
    // bucket fill + overflow count are read by every wave up front: the counters are reset behind the
    // workgroup barrier at the end of the kernel (the library keeps them zero between steps)
    const int cnt = KGE_DBG(a, 4096) ? 0 : min(a.counters[tile * 32], a.cap);
    const int on = KGE_DBG(a, 4096) ? 0 : min(a.counters[a.n_tiles * 32], a.ovf_cap);
    // BLOCK-INTERLEAVED ownership: the table is cut into blocks of TILE_RB consecutive rows and block b belongs to tile
    // b % n_tiles (local row r <-> table row row_of_tile(tile, r)).  Real graphs number their hubs first (ids are handed out
    // first-seen), so contiguous row ranges give the first tiles several times the entries of the others and the tile pass is
    // as slow as its busiest tile (zipf graph); dealing the blocks round the tiles spreads every popularity class.  Blocks
    // rather than single rows keep the optimizer's streams (x, m, v of a tile) in runs of TILE_RB rows: with single rows a
    // large table (C4: 123 k rows) lost 12 % to page locality.  Rows beyond the table in a tile's last block are skipped.
    // The forward kernel's per-block loss partials (complete: it finished before this launch started) are folded into the
    // caller's accumulator by the first LOSS_PARTS tiles, one slot each, while they start up.  A fold by the LAST tile -- a
    // returning exchange, a butterfly and an add behind everyone else's work -- sat on the critical path of every step.
    if (tid == 0)
        for (int sl = tile; sl < LOSS_PARTS; sl += a.n_tiles) {
            const unsigned long long old = atomicExch(reinterpret_cast<unsigned long long*>(a.loss_parts + (size_t)sl * LOSS_PART_STRIDE), 0ull);
            const double v = __longlong_as_double((long long)old);
            if (v != 0.0) atomicAdd(a.loss_sum, v);
        }
    const uint32_t NT = (uint32_t)a.n_tiles;
    const int nrow = a.tile_rows;
    const uint32_t RB = (uint32_t)a.rb;
    auto row_of = [&](int r) KGE_TILE_INLINE -> int64_t { return row_of_tile((uint32_t)tile, (uint32_t)r, NT, RB); };
    auto row_ok = [&](int r) KGE_TILE_INLINE -> bool { return row_of(r) < a.n_rows; };

    // Row ownership.  Short rows: wave wv owns local rows r with r % 16 == wv and covers the whole row (CH quads per lane).
    // Long rows (gw > 1): a GROUP of gw waves owns the row and each wave covers its 64-quad slice, so that the few entries
    // of a tile with few, long rows (C5: 18 rows of 8 KB) are spread over all lanes instead of over at most 16 owners.
    const int gw = a.gw, grp = wv / gw, wg = wv % gw, G = TILE_WAVES / gw;
    // an owner zeroes the rows it owns: no workgroup barrier is needed anywhere in this kernel
    bool qok[CH];
    int qoff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c + 64 * CH * wg;
        qok[c] = q < a.nq;
        qoff[c] = (qok[c] ? q : 0) * 4;
    }
    // RotatE with many entries per row: every corruption entry needs the tile's own live row e (g (e - S) / |e - S|); read
    // from the table that is one L2 round trip and 4K bytes per entry (half of the pass's traffic at the C2 shape), so the
    // owner keeps a copy of its rows in LDS behind the accumulators (written by the wave that owns the row, like the zeroes)
    float* own = acc + (size_t)a.tile_rows * a.K;
    const size_t acc_floats = (size_t)a.tile_rows * a.K * (a.own_cache ? 2 : 1);
    for (int r = grp; r < (KGE_DBG(a, 2048) ? 0 : nrow); r += G)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (qok[c]) {
                    *reinterpret_cast<float4*>(acc + (size_t)r * a.K + qoff[c] + h * a.k) = make_float4(0, 0, 0, 0);
                    if constexpr (MODEL == AMDKGE_ROTATE) {
                        if (a.own_cache && row_ok(r))
                            *reinterpret_cast<float4*>(own + (size_t)r * a.K + qoff[c] + h * a.k) = KGE_LD4(a.x + row_of(r) * a.K + qoff[c] + h * a.k);
                    }
                }
    // touched-rows mode: one flag byte per (row, wave of the owning group) behind the accumulators.  Every wave of a group
    // sees the same entries, so each keeps its own copy: written and read by the same wave, no synchronisation needed.
    uint8_t* tflag = reinterpret_cast<uint8_t*>(acc + acc_floats);
    if (a.lazy)
        for (int r = grp + G * lane; r < nrow; r += G * 64)
            tflag[r * gw + wg] = (a.touched && row_ok(r) && a.touched[row_of(r)]) ? 1 : 0;

    // Operand loads of one staged entry (all arguments wave-uniform): the staged side row and, for TransE / RotatE
    // corruption entries, the relation row (RotatE: its cos / sin from the per-step table) and the tile's own live row.
    // Kept apart from the arithmetic so that the loads of UNROLL entries are in flight together.
    constexpr int NX = TRILINEAR ? 1 : NC;   // trilinear models need no relation / own-row operands
    auto load_ops = [&](uint32_t pos, uint32_t meta, int pp, float4 (&v)[CH][NC], float4 (&pv)[CH][NX], float4 (&ev)[CH][NX], auto own_c) KGE_TILE_INLINE {
        constexpr bool OWN = decltype(own_c)::value;   // RotatE: the own row comes from the LDS copy (add_entry reads it)
        const int role = meta & 3;   // 0: corruption, object replaced; 1: corruption, subject replaced; 2: own s row; 3: own o row
        const int which = (role == 0) ? 2 : (role == 1) ? 3 : (role == 2) ? 0 : 1;
        const float* src = a.stage_rows + ((int64_t)pos * a.ns + which) * a.K;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int h = 0; h < NC; ++h) v[c][h] = *reinterpret_cast<const float4*>(src + qoff[c] + h * a.k);
        if constexpr (!TRILINEAR) {
            if (role < 2) {
                if constexpr (OWN) {
                    // (read from LDS by add_entry itself: nothing of the own row is held across the batch)
                } else {
                    const float* re = a.x + row_of((int)entry_local(meta)) * a.K;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int h = 0; h < NC; ++h) ev[c][h] = KGE_LD4(re + qoff[c] + h * a.k);
                }
                if constexpr (MODEL != AMDKGE_ROTATE) {   // (RotatE: the staged side row already carries the rotation)
                    const float* rp = a.rel + (int64_t)pp * a.K;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int h = 0; h < NC; ++h) pv[c][h] = KGE_LD4(rp + qoff[c] + h * a.k);
                }
            }
        }
    };
    auto add_entry = [&](uint32_t meta, float g, const float4 (&v)[CH][NC], const float4 (&pv)[CH][NX], const float4 (&ev)[CH][NX], auto own_c) KGE_TILE_INLINE {
        constexpr bool OWN = decltype(own_c)::value;
        const int role = meta & 3;
        const int lr = (int)entry_local(meta);
        float* arow = acc + (size_t)lr * a.K;
        if (a.lazy) tflag[lr * gw + wg] = 1;
        float4 out[CH][NC];
        if (TRILINEAR || role >= 2) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < NC; ++h) out[c][h] = make_float4(g * v[c][h].x, g * v[c][h].y, g * v[c][h].z, g * v[c][h].w);
        } else if constexpr (MODEL == AMDKGE_ROTATE) {
            // both sides: g (e - S) / |e - S| with S the staged side row (A = s o r, or B = o o conj(r)) and e the tile's own
            // row -- for object-side entries the very operations of grad_unit's dd
            float4 eo[CH][NC];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < NC; ++h)
                    if constexpr (OWN) eo[c][h] = *reinterpret_cast<const float4*>(own + (size_t)lr * a.K + qoff[c] + h * a.k);
                    else eo[c][h] = ev[c][h < NX ? h : 0];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dr = (&eo[c][0].x)[u] - (&v[c][0].x)[u], di = (&eo[c][1].x)[u] - (&v[c][1].x)[u];
                    const float m = kge_sqrt_t<DET>(dr * dr + di * di) + ((qoff[c] + u >= a.k_live) ? 1.f : 0.f);   // (padding units: 0 / 1)
                    const float gm = kge_div_t<DET>(g, m);
                    (&out[c][0].x)[u] = gm * dr;
                    (&out[c][1].x)[u] = gm * di;
                }
        } else {
            // TransE: the gradient w.r.t. the replaced row depends on that row -> same grad_unit arithmetic as the atomic
            // path, on (side row copy, live relation row, own live row)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float p[NC], e[NC], sd[NC], ds[NC], dp[NC], dd[NC];
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        p[h] = (&pv[c][h < NX ? h : 0].x)[u]; e[h] = (&ev[c][h < NX ? h : 0].x)[u]; sd[h] = (&v[c][h].x)[u];
                    }
                    if constexpr (MODEL != AMDKGE_ROTATE) prep_rel<MODEL>(a.mc, p);
                    const float pad1 = (MODEL == AMDKGE_ROTATE && qoff[c] + u >= a.k_live) ? 1.f : 0.f;
                    if (role == 0) grad_unit<MODEL>(sd, p, e, g, ds, dp, dd, pad1);
                    else grad_unit<MODEL>(e, p, sd, g, ds, dp, dd, pad1);
#pragma unroll
                    for (int h = 0; h < NC; ++h) (&out[c][h].x)[u] = (role == 0) ? dd[h] : ds[h];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!qok[c]) continue;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                float4* d = reinterpret_cast<float4*>(arow + qoff[c] + h * a.k);
                float4 t = *d;
                t.x += out[c][h].x; t.y += out[c][h].y; t.z += out[c][h].z; t.w += out[c][h].w;
                *d = t;
            }
        }
    };
    // entries of `mine` selected by `mask`, UNROLL at a time so that the operand loads of several entries are in flight
    auto process = [&](const StageEntry& mine, unsigned long long mask) KGE_TILE_INLINE {
        int mine_pp = 0;   // relation id of the lane's entry (TransE / RotatE): one gather per 64 entries, not one per entry
        if constexpr (MODEL == AMDKGE_TRANSE) {
            if (mask) mine_pp = a.triples[3 * (int64_t)mine.pos + 1];   // mine.pos is a valid positive index (0 for padding lanes)
        }
        while (mask) {
            uint32_t meta[UNROLL];
            float g[UNROLL];
            float4 v[UNROLL][CH][NC], pv[UNROLL][CH][NX], ev[UNROLL][CH][NX];
            int m = 0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (mask) {
                    const int tt = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const uint32_t pos = __builtin_amdgcn_readlane(mine.pos, tt);
                    meta[u] = __builtin_amdgcn_readlane(mine.meta, tt);
                    g[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.g), tt));
                    load_ops(pos, meta[u], __builtin_amdgcn_readlane(mine_pp, tt), v[u], pv[u], ev[u], std::false_type{});
                    m = u + 1;
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (u < m) add_entry(meta[u], g[u], v[u], pv[u], ev[u], std::false_type{});
        }
    };

    // ---- this tile's bucket: every wave walks all of it, 64 entries per coalesced 16-byte load ----
    const StageEntry* list = a.lists + (size_t)tile * a.cap;
    if (a.det) {
        // Deterministic mode.  The position of an entry in its bucket is decided by a returning atomic, i.e. by arrival
        // order, and fp32 addition is not associative.  Here the tile's entries (bucket + its share of the overflow list) are
        // first sorted in LDS by their full 128-bit content -- a canonical order that depends on the batch only (entries
        // that compare equal ARE equal) -- and then added in that order.
        uint4* sbuf = reinterpret_cast<uint4*>(smem + (((size_t)a.tile_rows * a.K * 4 + (size_t)a.tile_rows * gw + 15) & ~(size_t)15));
        __shared__ int s_total;
        if (tid == 0) s_total = cnt;
        for (int i = tid; i < cnt && i < a.sort_cap; i += TILE_THREADS) sbuf[i] = reinterpret_cast<const uint4*>(list)[i];
        __syncthreads();
        for (int base = 0; base < on; base += TILE_THREADS) {
            uint4 e = make_uint4(0, 0, 0, 0xFFFFFFFFu);
            if (base + tid < on) e = reinterpret_cast<const uint4*>(a.ovf)[base + tid];
            if (e.w != 0xFFFFFFFFu && (e.w / RB) % NT == (uint32_t)tile) {   // .w = dest
                const int at = atomicAdd(&s_total, 1);
                if (at < a.sort_cap) sbuf[at] = e;
            }
        }
        __syncthreads();
        int total = s_total;
        if (total > a.sort_cap) {   // more entries than the sort buffer holds (a very hot tile): flagged, the host raises
            if (tid == 0) atomicExch(a.status_flag, 1);
            total = a.sort_cap;
        }
        // Canonical order = ascending (pos, meta, bits of g, dest).  What has to be canonical is the order in which the entries of ONE
        // ROW are added, and a row belongs to one owner (a wave, or a group of waves that all walk the same entries): so every wave
        // first collects the INDICES of its own entries (the same ballot as the accumulation loop uses) into a private LDS queue and
        // sorts that queue by itself -- a bitonic network over <= 256 16-bit indices, keys read through them, no workgroup barrier
        // (a wave's LDS operations complete in order).  ~57 entries per wave at C2: 21 wave-local stages instead of a sort of the
        // whole bucket (the 55 - 66 block-wide stages of rounds 2 - 4 cost 31 of the pass's 116 us, profiles/r05g_det_sort_ablation.txt;
        // a block-wide LSD radix sort of indices, the fall-back below, 27).  The order of a row's entries is the one the whole-bucket
        // sort gave: same sums, same bits.  A wave with more entries than its queue holds (a hub's tile) sends the TILE to the fall-back.
        {
            __shared__ int s_qovf;
            if (tid == 0) s_qovf = 0;
            __syncthreads();
            // indices per wave: the 4 sort_cap bytes behind the entries, dealt to 16 waves (stride sort_cap / 8 uint16).  The bitonic
            // network below pads a queue to a power of two >= 64, so the queue's capacity is the largest power of two inside the
            // stride (sort_cap is a multiple of 64, not a power of two: with QC = the stride itself a wave holding 129 .. 208 of
            // 208 slots padded into its neighbour's queue); a stride below 64 cannot hold a network at all: fall-back.
            const int qstride = a.sort_cap / 8;
            int QC = 64;
            while (QC * 2 <= min(256, qstride)) QC *= 2;
            uint16_t* qix = reinterpret_cast<uint16_t*>(sbuf + a.sort_cap) + (size_t)wv * qstride;
            int qn = 0;
            bool over = qstride < 64;
            for (int base = 0; base < total && !over; base += 64) {
                const bool in = base + lane < total;
                const uint32_t meta = in ? sbuf[base + lane].y : 0u;
                const bool mineq = in && (int)(entry_local(meta) % G) == grp;
                const unsigned long long m = __ballot(mineq);
                const int c = __popcll(m);
                if (qn + c > QC) { over = true; break; }
                if (mineq) qix[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (uint16_t)(base + lane);
                qn += c;
            }
            if (over && lane == 0) s_qovf = 1;
            __syncthreads();
            if (!s_qovf && !KGE_DBG(a, 16384)) {
                int n2w = 64;
                while (n2w < qn) n2w <<= 1;
                for (int i = qn + lane; i < n2w; i += 64) qix[i] = 0xFFFFu;   // padding: sorts behind every entry
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                auto key_less = [&](uint16_t ix, uint16_t iy) KGE_TILE_INLINE -> bool {   // entry ix before entry iy?
                    if (iy == 0xFFFFu) return ix != 0xFFFFu;
                    if (ix == 0xFFFFu) return false;
                    const uint4 x = sbuf[ix], y = sbuf[iy];
                    if (x.x != y.x) return x.x < y.x;
                    if (x.y != y.y) return x.y < y.y;
                    if (x.z != y.z) return x.z < y.z;
                    if (x.w != y.w) return x.w < y.w;
                    return ix < iy;   // identical entries: any fixed order
                };
                for (int kk = 2; kk <= n2w; kk <<= 1)
                    for (int j = kk >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < n2w; i += 64) {
                            const int pi = i ^ j;
                            if (pi > i) {
                                const uint16_t x = qix[i], y = qix[pi];
                                const bool up = (i & kk) == 0;
                                if (up ? key_less(y, x) : key_less(x, y)) { qix[i] = y; qix[pi] = x; }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    }
                for (int base = 0; base < qn; base += 64) {
                    StageEntry mine{0u, 0u, 0.f, 0u};
                    const bool in = base + lane < qn;
                    if (in) { const uint4 e = sbuf[qix[base + lane]]; mine = StageEntry{e.x, e.y, __uint_as_float(e.z), e.w}; }
                    process(mine, __ballot(in));
                }
            } else {
        // ---- fall-back (a wave's queue overflowed): the whole bucket, block-wide.  An LSD radix sort on `pos` of 16-bit INDICES (the
        // bits B needs, one stable 1-bit split per pass: ballot + popcount within a 64-entry chunk, one wave's scan over the chunk
        // totals), then one pass that orders the entries of equal `pos` -- short runs -- by the remaining 96 bits.
        uint16_t* idxA = reinterpret_cast<uint16_t*>(sbuf + a.sort_cap);
        uint16_t* idxB = idxA + a.sort_cap;
        int* ccnt = reinterpret_cast<int*>(idxB + a.sort_cap);   // [sort_cap / 64] ones per chunk, then their exclusive prefix
        __shared__ int s_ones;
        const int nchunk = (total + 63) >> 6;
        for (int i = tid; i < total; i += TILE_THREADS) idxA[i] = (uint16_t)i;
        __syncthreads();
        constexpr int SORT_IT = 8;   // sort_cap <= 8192 = 8 entries per thread
        for (int bit = 0; bit < (KGE_DBG(a, 16384) ? 0 : a.pos_bits); ++bit) {   // (ablation 16384: entries stay in arrival order)
            int myid[SORT_IT], mybit[SORT_IT], myin[SORT_IT];
#pragma unroll
            for (int it = 0; it < SORT_IT; ++it) {
                const int e = tid + it * TILE_THREADS;
                myid[it] = 0; mybit[it] = 0; myin[it] = 0;
                if ((e & ~63) < total) {   // (whole chunks take part in the ballot; entries beyond `total` count as zeros and are not scattered)
                    const bool valid = e < total;
                    myid[it] = valid ? idxA[e] : 0;
                    mybit[it] = valid ? (int)((sbuf[myid[it]].x >> bit) & 1u) : 0;
                    const unsigned long long m = __ballot(mybit[it] != 0);
                    myin[it] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    if (lane == 0) ccnt[e >> 6] = __popcll(m);
                }
            }
            __syncthreads();
            if (wv == 0) {   // exclusive prefix over the chunk totals (<= 128 chunks: two per lane)
                int carry = 0;
                for (int c0 = 0; c0 < nchunk; c0 += 64) {
                    const int c = c0 + lane;
                    const int v = c < nchunk ? ccnt[c] : 0;
                    int incl = v;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                    if (c < nchunk) ccnt[c] = carry + incl - v;
                    carry += __shfl(incl, 63, 64);
                }
                if (lane == 0) s_ones = carry;
            }
            __syncthreads();
            const int zeros = total - s_ones;
#pragma unroll
            for (int it = 0; it < SORT_IT; ++it) {
                const int e = tid + it * TILE_THREADS;
                if (e < total) {
                    const int ones_before = ccnt[e >> 6] + myin[it];
                    idxB[mybit[it] ? zeros + ones_before : e - ones_before] = (uint16_t)myid[it];
                }
            }
            __syncthreads();
            uint16_t* t = idxA; idxA = idxB; idxB = t;
        }
        // entries of equal pos are adjacent now (in arrival order): rank each one inside its run by (meta, g, dest), position as tie-break
        if (!KGE_DBG(a, 16384)) {
            for (int i = tid; i < total; i += TILE_THREADS) {
                const uint4 me = sbuf[idxA[i]];
                int gs = i;
                while (gs > 0 && sbuf[idxA[gs - 1]].x == me.x) --gs;
                int r = 0;
                for (int j = gs; j < total; ++j) {
                    if (j == i) continue;
                    const uint4 y = sbuf[idxA[j]];
                    if (y.x != me.x) break;
                    const bool lt = (y.y != me.y) ? (y.y < me.y) : ((y.z != me.z) ? (y.z < me.z) : ((y.w != me.w) ? (y.w < me.w) : (j < i)));
                    r += lt ? 1 : 0;
                }
                idxB[gs + r] = idxA[i];
            }
            __syncthreads();
            uint16_t* t = idxA; idxA = idxB; idxB = t;
        }
        for (int base = 0; base < total; base += 64) {
            StageEntry mine{0u, 0u, 0.f, 0u};
            const bool in = base + lane < total;
            if (in) { const uint4 e = sbuf[idxA[base + lane]]; mine = StageEntry{e.x, e.y, __uint_as_float(e.z), e.w}; }
            process(mine, __ballot(in && (int)(entry_local(mine.meta) % G) == grp));
        }
            }   // fall-back
        }
    } else if (!tile_queued(MODEL, CH, a.K)) {
        // Trilinear models with rows beyond 1 KB (ComplEx, DistMult k > 256) keep the chunk-by-chunk form: they are bandwidth-
        // bound as they are (5.9 TB/s at C2), the queued form's operand sets do not fit the 128 VGPRs a 1024-thread workgroup
        // leaves per lane (ComplEx: arrays went to scratch, 6x slower), and its 32 KB of LDS would shrink their tiles
        // (measured: C3 and C4 7-8 % slower).
        for (int base = 0; base < cnt; base += 64) {
            StageEntry mine{0u, 0u, 0.f, 0u};
            const bool in = base + lane < cnt;
            if (in) mine = list[base + lane];
            process(mine, __ballot(in && (int)(entry_local(mine.meta) % G) == grp));
        }
        for (int base = 0; base < on; base += 64) {   // overflow list (entries of buckets that were full): every tile filters all of it
            StageEntry mine{0u, 0u, 0.f, 0xFFFFFFFFu};
            if (base + lane < on) mine = a.ovf[base + lane];
            const bool hit = mine.dest != 0xFFFFFFFFu && (mine.dest / RB) % NT == (uint32_t)tile;
            process(mine, __ballot(hit && (int)(entry_local(mine.meta) % G) == grp));
        }
    } else {
    // Two phases per wave.  Scanning a 64-entry chunk yields only ~64/16 entries for this wave: processed chunk by chunk, a
    // wave had 4 operand rows in flight and the chunk loads were serialised behind them (narrow rows -- TransE, DistMult
    // k <= 256, C1 -- ran at a fraction of the fabric bandwidth: the pass took the same ~66 us whatever the row width).
    // Now the wave first COLLECTS its entries into a private LDS queue (chunk loads software-pipelined, no row traffic),
    // then drains the queue UNROLL entries at a time with all operand loads of a batch in flight.
    uint4* queue = reinterpret_cast<uint4*>(smem + ((acc_floats * 4 + (a.lazy ? (size_t)a.tile_rows * gw : 0) + 15) & ~(size_t)15)) +
                   (size_t)wv * TILE_QCAP;
    int qn = 0;
    auto drain_n = [&](auto n_c, auto own_c, const uint4* q) KGE_TILE_INLINE {
        constexpr int UN = decltype(n_c)::value;
        for (int i0 = 0; i0 < qn; i0 += UN) {
            uint32_t meta[UN];
            float g[UN];
            float4 v[UN][CH][NC], pv[UN][CH][NX], ev[UN][CH][NX];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (i0 + u < qn) {
                    const uint4 e = q[i0 + u];   // same address in every lane: one LDS broadcast read, then scalars
                    meta[u] = __builtin_amdgcn_readfirstlane(e.y);
                    g[u] = __uint_as_float(__builtin_amdgcn_readfirstlane(e.z));
                    load_ops(__builtin_amdgcn_readfirstlane(e.x), meta[u], (int)__builtin_amdgcn_readfirstlane(e.w), v[u], pv[u], ev[u], own_c);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (i0 + u < qn) add_entry(meta[u], g[u], v[u], pv[u], ev[u], own_c);
        }
        qn = 0;
    };
    auto drain = [&](const uint4* q) KGE_TILE_INLINE {
        if constexpr (MODEL == AMDKGE_ROTATE) {
            // (with the own rows in LDS an entry holds one operand row; twice the entries in flight measured no different)
            if (a.own_cache) { drain_n(std::integral_constant<int, UNROLL>{}, std::true_type{}, q); return; }
        }
        drain_n(std::integral_constant<int, UNROLL>{}, std::false_type{}, q);
    };
    if constexpr (MODEL != AMDKGE_TRANSE) {
        // (this form is kept as it was measured: restating it as the single loop below cost the RotatE instantiation 17 % of
        // the pass -- the compiler's schedule of the drain, not its work, changed)
        auto collect = [&](const StageEntry& mine, bool sel) KGE_TILE_INLINE {
            const unsigned long long mask = __ballot(sel);
            if (!mask) return;
            if (qn + 64 > TILE_QCAP) drain(queue);
            if (sel) {
                uint32_t pp = 0;   // relation id of the entry's positive (TransE / RotatE), carried in place of `dest`
                if constexpr (MODEL == AMDKGE_TRANSE) pp = (uint32_t)a.triples[3 * (int64_t)mine.pos + 1];
                queue[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0))] =
                    make_uint4(mine.pos, mine.meta, __float_as_uint(mine.g), pp);
            }
            qn += __popcll(mask);
        };
        {
            StageEntry next{0u, 0u, 0.f, 0u};
            if (lane < cnt) next = list[lane];
            for (int base = 0; base < cnt; base += 64) {
                const StageEntry mine = next;
                const bool in = base + lane < cnt;
                if (base + 64 + lane < cnt) next = list[base + 64 + lane];   // the next chunk is in flight while this one is filed
                collect(mine, in && (int)(entry_local(mine.meta) % G) == grp);
            }
        }
        // ---- overflow list (entries of buckets that were full): every tile filters all of it ----
        for (int base = 0; base < on; base += 64) {
            StageEntry mine{0u, 0u, 0.f, 0xFFFFFFFFu};
            if (base + lane < on) mine = a.ovf[base + lane];
            const bool hit = mine.dest != 0xFFFFFFFFu && (mine.dest / RB) % NT == (uint32_t)tile;
            collect(mine, hit && (int)(entry_local(mine.meta) % G) == grp);
        }
        drain(queue);
    } else {
    // TransE with sign codes: an entry's whole operand is ONE dword per lane, so such entries get a queue of their own
    // (the first TILE_QFAST slots) and are drained FAST_U at a time -- the drain is a chain of load round trips, one per batch,
    // and the 4 entries per batch the three-row form allows left the pass latency-bound (47 of its 69 us at k = 200).
    // Own-row entries keep the general queue (the remaining slots) and the general drain; an entry whose codes show a (near-)zero
    // unit is redone in the three-row form on the spot.
    constexpr int TILE_QFAST = 64, FAST_U = 16;   // (both queues hold a whole 64-entry chunk)
    const bool coded = a.sign_codes != nullptr;
    uint4* const queue_slow = coded ? queue + TILE_QFAST : queue;
    const int slow_cap = coded ? TILE_QCAP - TILE_QFAST : TILE_QCAP;
    int qf = 0;
    uint32_t padfill[CH];   // 0x01 in the bytes of this lane's padding units (units >= k_live), 0 elsewhere
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        padfill[c] = 0u;
#pragma unroll
        for (int x = 0; x < 4; ++x) padfill[c] |= (qoff[c] + x >= a.k_live) ? (1u << (8 * x)) : 0u;
    }
    auto drain_fast = [&]() KGE_TILE_INLINE {
        {
            for (int i0 = 0; i0 < qf; i0 += FAST_U) {
                uint32_t meta[FAST_U], gb[FAST_U], cw[FAST_U][CH];
                uint32_t redo = 0u;
#pragma unroll
                for (int u = 0; u < FAST_U; ++u) {
                    if (i0 + u < qf) {
                        const uint4 e = queue[i0 + u];
                        meta[u] = __builtin_amdgcn_readfirstlane(e.y);
                        gb[u] = __builtin_amdgcn_readfirstlane(e.z);
                        const uint32_t* src = a.sign_codes + ((int64_t)__builtin_amdgcn_readfirstlane(e.x) * a.eta + (meta[u] >> ENTRY_J_SHIFT)) * a.nq;
#pragma unroll
                        for (int c = 0; c < CH; ++c) cw[u][c] = src[qoff[c] >> 2];
                    }
                }
#pragma unroll
                for (int u = 0; u < FAST_U; ++u) {
                    if (i0 + u < qf) {
                        // a unit of the model whose byte shows |d| < 2^-125 (a zero byte once the sign bit is masked; padding
                        // units are made non-zero): sign(0) = 0 must hold exactly -> the entry is redone in the three-row form.
                        // So is one whose byte shows an all-ones exponent top (0x7f: |d| >= 2^127, inf or NaN): sign(NaN) must be NaN,
                        // which grad_unit gives and a sign bit cannot (round 6).
                        bool tiny = false;
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const uint32_t t = (cw[u][c] & 0x7f7f7f7fu) | padfill[c];
                            const uint32_t t7 = t ^ 0x7f7f7f7fu;
                            tiny |= qok[c] && ((((t - 0x01010101u) & ~t) | ((t7 - 0x01010101u) & ~t7)) & 0x80808080u) != 0u;
                        }
                        if (__ballot(tiny)) {   // rare: noted, redone behind the batch (keeps the three-row code out of this unrolled loop)
                            redo |= 1u << u;
                            continue;
                        }
                        const int lr = (int)entry_local(meta[u]);
                        float* arow = acc + (size_t)lr * a.K;
                        if (a.lazy) tflag[lr * gw + wg] = 1;
                        const unsigned g2 = (meta[u] & 3u) == 0u ? gb[u] ^ 0x80000000u : gb[u];   // -/+ g sign(d): role 0 is d/do = -g sign(d) (grad_unit)
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            if (!qok[c]) continue;
                            float4* dst = reinterpret_cast<float4*>(arow + qoff[c]);
                            float4 t = *dst;
#pragma unroll
                            for (int x = 0; x < 4; ++x) {
                                const float y = __uint_as_float(__builtin_amdgcn_bitop3_b32(cw[u][c] << (24 - 8 * x), 0x80000000u, g2, 0x6a));
                                (&t.x)[x] += (qoff[c] + x < a.k_live) ? y : 0.f;
                            }
                            *dst = t;
                        }
                    }
                }
                while (redo) {
                    const int u = __builtin_ctz(redo);
                    redo &= redo - 1;
                    const uint4 e = queue[i0 + u];
                    const uint32_t pos = __builtin_amdgcn_readfirstlane(e.x), m16 = __builtin_amdgcn_readfirstlane(e.y) & 0xFFFFu;
                    float4 v1[CH][NC], pv1[CH][NX], ev1[CH][NX];
                    load_ops(pos, m16, a.triples[3 * (int64_t)pos + 1], v1, pv1, ev1, std::false_type{});
                    add_entry(m16, __uint_as_float(__builtin_amdgcn_readfirstlane(e.z)), v1, pv1, ev1, std::false_type{});
                }
            }
        }
        qf = 0;
    };
    // TransE: ONE loop over the chunks of the bucket and then of the overflow list (entries of buckets that were full: every
    // tile filters all of it), both queues drained at a single place in it.
    const int nb = (cnt + 63) >> 6, nch = nb + ((on + 63) >> 6);
    auto load_chunk = [&](int it) KGE_TILE_INLINE -> StageEntry {
        StageEntry en{0u, 0u, 0.f, 0xFFFFFFFFu};   // dest = ~0: not an entry
        if (it < nb) { if (it * 64 + lane < cnt) en = list[it * 64 + lane]; }
        else if (it < nch) { if ((it - nb) * 64 + lane < on) en = a.ovf[(it - nb) * 64 + lane]; }
        return en;
    };
    StageEntry next = load_chunk(0);
    for (int it = 0; it <= nch; ++it) {
        const StageEntry mine = next;
        next = load_chunk(it + 1);   // the next chunk is in flight while this one is filed
        bool sel = mine.dest != 0xFFFFFFFFu && (int)(entry_local(mine.meta) % G) == grp;
        if (it >= nb) sel = sel && (mine.dest / RB) % NT == (uint32_t)tile;
        const bool fast = coded && sel && (mine.meta & 3u) < 2u;   // sign-coded corruption entries: the fast queue
        sel = sel && !fast;
        const unsigned long long mf = __ballot(fast), ms = __ballot(sel);
        const int nf = __popcll(mf), ns = __popcll(ms);
#ifdef KGE_DRAIN_EARLY
        if (it == nch || qf + 64 > TILE_QFAST || qn + 64 > slow_cap) {
#else
        if (it == nch || qf + nf > TILE_QFAST || qn + ns > slow_cap) {
#endif
            drain_fast();
            drain(queue_slow);
        }
        if (fast) queue[qf + __builtin_amdgcn_mbcnt_hi((unsigned)(mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mf, 0))] =
                      make_uint4(mine.pos, mine.meta, __float_as_uint(mine.g), 0u);
        qf += nf;
        if (sel) {
            uint32_t pp = 0;   // relation id of the entry's positive (TransE / RotatE), carried in place of `dest`
            if constexpr (!TRILINEAR) {
                if (!coded) pp = (uint32_t)a.triples[3 * (int64_t)mine.pos + 1];   // (with sign codes only own-row entries come this way)
            }
            queue_slow[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0))] =
                make_uint4(mine.pos, mine.meta, __float_as_uint(mine.g), pp);
        }
        qn += ns;
    }
    }
    }

    // ---- flush: the tile's rows leave LDS exactly once ------------------------------------------------
    // (the optimizer kind is dispatched once, outside the row loop: one compiled flush loop per update rule)
    float reg_acc = 0.f;
    auto flush = [&](auto kind_c) KGE_TILE_INLINE {
    constexpr int KIND = decltype(kind_c)::value;
    // (Round 4, measured and dropped: software-pipelining this loop -- the next row's x / m / v requested before the current row
    // is updated and stored, two named register sets -- changed nothing at C2 (0.1416 vs 0.1411 ms/step), cost 3 % at C4 and 7 % at
    // the C5 row width, gained 2 % at C3 (profiles/r04c_flush_ab.jsonl): with 16 waves per CU in the flush the round trips of one
    // wave are already covered by the others; the flush is bandwidth, not latency.)
    for (int r = grp; r < (KGE_DBG(a, 1024) ? 0 : nrow); r += G) {
        const float* arow = acc + (size_t)r * a.K;
        if (!row_ok(r)) continue;
        const int hot = a.hot_map ? a.hot_map[row_of(r)] : 0;   // a hot row's own-gradient rows wait in its replicas (always "touched")
        if (a.lazy && a.apply_update && !hot && !tflag[r * gw + wg]) continue;   // untouched row: x, slots, regulariser stay as they are
        // Round 5: the loads of BOTH halves of a row (x, m, v each) are issued before the first element is updated -- twice the bytes
        // in flight per wave (the hypothesis: 16 waves x 256 CUs x 2.4 KB = 10 MB chip-wide caps the flush near 4 TB/s at a ~2.5 us
        // loaded round trip; C4's tile pass moves 4.24 TB/s of counter traffic, profiles/r05a_c4_pmc_traffic.json).  MEASURED: no
        // change -- C4 0.3908 vs 0.3934 ms, C2 0.1408 vs 0.1406, C3 0.250 vs 0.253 (profiles/r05c_benches.jsonl).  Like round 4's
        // cross-row pipeline it shows the flush is not short of loads in flight; kept because it costs nothing.
        float4 gq[CH][NC], xq[CH][NC], mq[CH][NC], vq[CH][NC];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!qok[c]) continue;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                float4 g = *reinterpret_cast<const float4*>(arow + qoff[c] + h * a.k);
                const int64_t off = row_of(r) * a.K + qoff[c] + h * a.k;
                float4* gp4 = reinterpret_cast<float4*>(a.g_ent + off);
                if (hot) {   // sum the replicas in fixed order and leave them zero for the next step
                    float4* hp = reinterpret_cast<float4*>(a.hot_buf + (int64_t)(hot - 1) * HOT_REPL * a.K + qoff[c] + h * a.k);
#pragma unroll 4
                    for (int q = 0; q < HOT_REPL; ++q) {
                        const float4 t = hp[(size_t)q * (a.K >> 2)];
                        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                        hp[(size_t)q * (a.K >> 2)] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (a.pos_atomic) {   // rows of the positives' own s / o
                    const float4 gd = *gp4;
                    g.x += gd.x; g.y += gd.y; g.z += gd.z; g.w += gd.w;
                    if (a.apply_update) *gp4 = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                gq[c][h] = g;
                xq[c][h] = make_float4(0, 0, 0, 0); mq[c][h] = xq[c][h]; vq[c][h] = xq[c][h];
                if (!a.apply_update) {
                    *gp4 = g;
                    continue;
                }
                xq[c][h] = *reinterpret_cast<const float4*>(a.x + off);
                if constexpr (opt_nslots(KIND) >= 1) mq[c][h] = *reinterpret_cast<const float4*>(a.s0 + off);
                if constexpr (opt_nslots(KIND) == 2) vq[c][h] = *reinterpret_cast<const float4*>(a.s1 + off);
            }
        }
        if (!a.apply_update) continue;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!qok[c]) continue;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const int64_t off = row_of(r) * a.K + qoff[c] + h * a.k;
                float4 x = xq[c][h], m = mq[c][h], v = vq[c][h];
                const float4 g = gq[c][h];
                opt_elem<KIND>(a.opt, x.x, g.x, m.x, v.x, reg_acc); opt_elem<KIND>(a.opt, x.y, g.y, m.y, v.y, reg_acc);
                opt_elem<KIND>(a.opt, x.z, g.z, m.z, v.z, reg_acc); opt_elem<KIND>(a.opt, x.w, g.w, m.w, v.w, reg_acc);
                if constexpr (opt_nslots(KIND) >= 1) *reinterpret_cast<float4*>(a.s0 + off) = m;
                if constexpr (opt_nslots(KIND) == 2) *reinterpret_cast<float4*>(a.s1 + off) = v;
                *reinterpret_cast<float4*>(a.x + off) = x;
            }
        }
    }
    };
#define KGE_FLUSH(KIND) flush(std::integral_constant<int, KIND>{})
    KGE_OPT_DISPATCH(a.opt.kind, KGE_FLUSH)
#undef KGE_FLUSH
    if (a.apply_update && a.reg_loss && a.opt.lam != 0.f) {
        // per-wave regulariser terms go to the partial slots as well (second double of a slot), folded by the last tile
        const float w = wave_sum(reg_acc);
        if (lane == 0) atomicAdd(a.loss_parts + (size_t)((tile * TILE_WAVES + wv) & (LOSS_PARTS - 1)) * LOSS_PART_STRIDE + 1, (double)a.opt.lam * (double)w);
    }
    // ---- leave the bookkeeping zeroed for the next step: own bucket now, overflow count by the last tile ----
    __syncthreads();
    if (a.touched)   // every wave has read its flags: clear the forward kernel's marks for the next step
        for (int r = tid; r < nrow; r += TILE_THREADS)
            if (row_ok(r)) a.touched[row_of(r)] = 0;
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
    // the regulariser partials (added by every tile's waves before the ticket) are folded by the last tile to finish; the data
    // loss was folded when the kernel started (below the fill counts)
    if (s_last && wv == 0 && a.apply_update && a.reg_loss && a.opt.lam != 0.f) fold_loss_parts(a.loss_parts, a.reg_loss, lane, 1);
}

}  // namespace kge
#include "kge_tile_direct.h"
#include "kge_train_cols.h"
namespace kge {

// Deterministic mode: the relation-row gradient.  One workgroup per relation collects the positives of its relation IN BATCH ORDER
// and adds their staged fifth rows in that order, every thread owning fixed columns of the row: the same additions in the same
// order on every run (and in oracle/train_ordered.py).
// Round 5: the first version walked the batch 256 positions at a time -- a dependent chain of (triple load, two barriers, row loads)
// per step, 40 steps at B = 10 000: 61.6 us of latency for 42 rows per relation (profiles/r05a_splits.log).  Now the batch is taken in
// segments of 8 192 positions: every wave scans its contiguous quarter of the segment with all 32 triple loads issued up front
// and compacts its hits into its own LDS list (wave-local ballots, no barrier); the four lists, read one after the other, are the
// segment's hits in batch order; the rows are then added with 16 loads in flight per thread and the adds in list order
// (24.7 us with 4 096-position segments and 8 loads in flight, profiles/r05b_c4_direct_short_rows_and_splits.txt).
constexpr int RELDET_SEG = 8192, RELDET_Q = RELDET_SEG / 4, RELDET_UN = 16;   // (32 KB of lists; B = 10 000: two segments)
__global__ __launch_bounds__(256) void rel_backward_det_kernel(const int32_t* __restrict__ triples, int64_t B, const float* __restrict__ stage_rows,
                                                               int ns, int K, float* __restrict__ g_rel) {
    __shared__ int s_list[4][RELDET_Q];
    __shared__ int s_cnt[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float acc[16];   // K <= 4096 floats (k <= 2048 complex units): 16 columns per thread
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int64_t seg = 0; seg < B; seg += RELDET_SEG) {
        const int64_t q0 = seg + (int64_t)wv * RELDET_Q;
        int pv[RELDET_Q / 64];
#pragma unroll
        for (int j = 0; j < RELDET_Q / 64; ++j) {
            const int64_t i = q0 + j * 64 + lane;
            pv[j] = i < B ? triples[3 * i + 1] : -1;
        }
        int n = 0;
#pragma unroll
        for (int j = 0; j < RELDET_Q / 64; ++j) {
            const bool hit = pv[j] == r;
            const unsigned long long m = __ballot(hit);
            if (hit) s_list[wv][n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = wv * RELDET_Q + j * 64 + lane;
            n += __popcll(m);
        }
        if (lane == 0) s_cnt[wv] = n;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = tid + 256 * c;
            if (col >= K) break;
            float a = acc[c];
            for (int w = 0; w < 4; ++w) {
                const int nw = s_cnt[w];
                for (int h0 = 0; h0 < nw; h0 += RELDET_UN) {
                    float v[RELDET_UN];
#pragma unroll
                    for (int u = 0; u < RELDET_UN; ++u)
                        v[u] = (h0 + u < nw) ? stage_rows[((seg + s_list[w][h0 + u]) * (int64_t)ns + 4) * K + col] : 0.f;
#pragma unroll
                    for (int u = 0; u < RELDET_UN; ++u)
                        if (h0 + u < nw) a += v[u];
                }
            }
            acc[c] = a;
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (tid + 256 * c < K) g_rel[(int64_t)r * K + tid + 256 * c] += acc[c];
}

// RotatE: cos / sin of every relation phase, once per step, in the declared correctly rounded form (prep_rel_exact, kge_device.h); the tile pass sees the
// very values the forward kernel used)
__global__ __launch_bounds__(256) void rel_phase_kernel(const float* __restrict__ rel, int64_t n_rels, int k, int K, ModelConst mc,
                                                        float* __restrict__ cs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rels * k) return;
    const int64_t r = idx / k;
    const int c = (int)(idx - r * k);
    float p[2] = {rel[r * K + c], 0.f};
    prep_rel_exact<AMDKGE_ROTATE>(mc, p);
    cs[r * K + c] = p[0];
    cs[r * K + k + c] = p[1];
}

// ---- plan: tile size, bucket capacity and workspace layout (shared by the two entry points) ------------------
struct TiledPlan {
    int tile_rows, n_tiles, cap, ovf_cap, rb;
    int ns, sort_cap;   // deterministic mode: 5 staged rows per positive, LDS sort buffer entries (0 otherwise)
    size_t off_cnt, off_lists, off_ovf, off_rows, off_cs, off_touch, off_loss, off_flag, off_hot_map, off_hot_buf, off_codes, total;
    bool own_cache;     // RotatE, queued form, >= 4 corruption entries per table row and step: own rows cached in LDS
    bool codes;         // TransE with one wave per positive: the forward kernel hands the signs of d_j to the tile pass (ENTRY_EXACT)
    bool direct;        // long rows (> 128 quads per half): the row-direct tile pass (kge_tile_direct.h)
};

static int g_tile_direct = 1;   // amdkge_set_tile_direct (A/B measurements, tests): 0 keeps long rows on tile_backward_kernel
// (Round 5 also offered the row-direct pass to SHORTER rows -- 32 .. 128 quads per half, BASELINE configs[3]'s 1.6 KB rows, one or two
// waves per row.  MEASURED SLOWER than the LDS-accumulator tiles where it was meant to help: C4 tile pass 339 us vs 476 us, whole step
// 0.393 vs 0.551 ms, profiles/r05b_c4_direct_short_rows_and_splits.txt -- one wave per row is a serial chain of row round trips, the
// LDS tiles keep 16 waves of a CU on 16 different rows.  Removed from the product in round 6: `git show db3a690` has it.)

static bool make_plan(const amdkge_model* m, int64_t B, int32_t eta, TiledPlan& p, bool det = false, bool det_wide = false) {
    const int ks = stored_k(m), K = row_floats(m);
    if (ks % 4 != 0 || ks > 2048) return false;   // 16-byte layout; one wave (k <= 512) or one workgroup (k <= 2048) per positive
    if ((ks <= 512 ? (size_t)4 * slot_lds_bytes(eta, 1) + 32 + 16 * (size_t)K : slot_lds_bytes(eta, 4) + 8 + 4 * (size_t)K) +
            sign_stash_bytes(m->scoring_type, eta, 2) > 150 * 1024) return false;
    const int64_t entries = B * (eta + 2);
    if (entries >= (1ll << 31)) return false;
    p.ns = det ? 5 : 4;
    p.sort_cap = 0;
    // deterministic mode shares the LDS between the accumulators and the sort buffer: shrink the tiles until a bucket
    // (+ slack for overflow entries) fits the buffer next to them
    const int model_t = m->scoring_type == AMDKGE_HOLE ? AMDKGE_COMPLEX : m->scoring_type;
    const size_t queue_bytes = tile_queued(model_t, tile_ch_of(ks / 4), K) ? TILE_QUEUE_BYTES : 0;
    // (few entries per row -- one GPU's C5 shard sees 0.7 -- would only halve the tiles)
    p.own_cache = !det && m->scoring_type == AMDKGE_ROTATE && queue_bytes != 0 && B * (int64_t)eta >= 4 * m->n_ents;
    const size_t row_bytes = (size_t)K * 4 * (p.own_cache ? 2 : 1);
#ifndef KGE_DIRECT_BUDGET_KB
#define KGE_DIRECT_BUDGET_KB 1200
#endif
    // (long rows on the row-direct pass keep nothing but their entry list in LDS: the tile size is not an LDS question there)
    // The row-direct pass pays off while a row sees few entries per step (one GPU's C5 shard: 0.7; a 1 M-row table at the same
    // batch: 4.3 -- 21.8 vs 26.2 ms); where rows collect many (ComplEx k = 1000 on 14 505 entities: 15 -- 0.88 vs 0.72 ms) the
    // LDS accumulators win, so the form is chosen by the batch's mean entries per row.
    const bool direct_long = g_tile_direct && !det && ks / 4 > 128 && B * (int64_t)(eta + 2) <= 8 * m->n_ents;
    const bool direct_shape = direct_long;
    const size_t nodet_budget = direct_shape ? (size_t)KGE_DIRECT_BUDGET_KB * 1024 : 150 * 1024 - queue_bytes;
    for (size_t budget = det ? 96 * 1024 : nodet_budget;; budget = budget * 3 / 4) {
        // Whole ownership blocks per tile (block-interleaved ownership, see tile_backward_kernel).  The block size is the largest
        // power of two <= TILE_RB that still lets the tiles fill the 256 CUs evenly: a tile's rows come in multiples of the
        // block, and at C2 (14 505 rows, ~57 per tile) blocks of 8 would leave 11 % of the CUs without a tile.
        int fit = (int)(budget / row_bytes);
        if (direct_shape && fit > 160) fit = 160;   // (<= 8 entries per row: a bucket of at most 2 * 1 280 + 256 entries -- the row-direct pass's LDS list)
        if (fit < 1) return false;
        double best_eff = -1.0;
        for (int rb = (int)TILE_RB; rb >= 1; rb >>= 1) {
            if (rb > fit) continue;
            const int64_t blocks = (m->n_ents + rb - 1) / rb;
            int64_t per = 0;
            for (int64_t mm = 1;; ++mm) {   // smallest number of CU rounds whose tile fits the LDS
                per = (blocks + 256 * mm - 1) / (256 * mm);
                if (per * rb <= fit && per * rb <= 4096) break;
            }
            const int64_t nt = (blocks + per - 1) / per, rounds = (nt + 255) / 256;
            const double eff = (double)m->n_ents / (double)(per * rb) / (double)(rounds * 256);
            if (eff > best_eff + 0.03 || best_eff < 0) {   // a smaller block must buy at least 3 % of the chip
                best_eff = eff; p.rb = rb; p.tile_rows = (int)(per * rb); p.n_tiles = (int)nt;
            }
            if (eff >= 0.97) break;
        }
        if (best_eff < 0) return false;
        // bucket capacity: twice the mean + slack (Poisson tail; anything beyond goes to the overflow list)
        const int64_t mean = (entries + p.n_tiles - 1) / p.n_tiles;
        p.cap = (int)(2 * mean + (mean >= 224 ? 256 : 32 + mean));
        // development aid (scripts/nan_hunt.py): a tiny bucket capacity sends most entries through the shared overflow list
        static const int dbg_cap = [] { const char* e = getenv("AMDKGE_DEBUG_BUCKET_CAP"); return e ? atoi(e) : 0; }();
        if (dbg_cap > 0 && p.cap > dbg_cap) p.cap = dbg_cap;
        if (!det) break;
        // the sort buffer holds a whole bucket + slack for overflow entries.  (Rounds 2 - 4 sorted the bucket with a bitonic network
        // and needed a POWER OF TWO here: 4 096 entries = 64 KB for C2's buckets of 2 330, which did not fit beside 57-row tiles --
        // the deterministic mode ran 45-row tiles in two rounds on the 256 CUs.  The index sorts of round 5 need no such rounding:
        // the deterministic mode now has the default mode's tile geometry.)
        int sc = (p.cap + 64 + 63) & ~63;
        if (det_wide) {   // AMDKGE_TILED_DET_WIDE_SORT (skewed graphs): room for a hub's tile -- at least twice a bucket, a power of two as in rounds 2 - 4
            sc = 64;
            while (sc < 2 * (p.cap + 64)) sc <<= 1;
        }
        const size_t fixed = (size_t)p.tile_rows * K * 4 + 4096 + 16 + 1024;
        if (sc <= 8192 && fixed + (size_t)sc * 20 <= 158 * 1024) {   // per entry: its 16 bytes + two 16-bit index arrays (+ 1 KB of chunk counters)
            // take what the LDS still offers (up to 8192 entries): a hub's tile receives many times the mean
            while (sc * 2 <= 8192 && fixed + (size_t)sc * 2 * 20 <= 158 * 1024) sc *= 2;
            p.sort_cap = sc;
            break;
        }
        if (p.tile_rows == 1) return false;   // one row per tile and its bucket still does not fit: not a shape for this mode
    }
    p.ovf_cap = (int)(entries > 0 ? entries : 1);
    // row-direct pass: one wave group per tile, the bucket sorted in LDS, rows through registers -- long rows only, and only
    // while a bucket (+ slack for overflow entries) is a small LDS list
    p.direct = direct_shape && p.cap <= 3000 && p.tile_rows <= 2048;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    p.off_loss = o; o += up((size_t)LOSS_PARTS * LOSS_PART_STRIDE * 8);   // first: the same place in every plan (kept zero between steps)
    p.off_flag = o; o += 256;                                             // sticky status flag (amdkge_train_tiled_status): also plan-independent
    p.off_hot_map = o; o += up((size_t)m->n_ents);                        // hot-row map and replicas: also independent of B / eta / mode,
    p.off_hot_buf = o; o += up((size_t)HOT_MAX * HOT_REPL * K * 4);       // written by amdkge_train_tiled_set_hot_rows
    p.off_touch = o; o += up((size_t)m->n_ents);                          // byte per entity row (lazy optimizer + POS_ATOMIC), kept zero between steps
    // Everything from here on moves with the plan (n_tiles and cap depend on B through own_cache, the bucket capacity and the
    // deterministic mode's LDS split): see plan_guard() for what keeps the counters of one plan safe from the lists of another.
    p.off_cnt = o; o += up((size_t)(p.n_tiles + 2) * 32 * 4);   // bucket fills, overflow count, finished-tiles ticket
    p.off_lists = o; o += up((size_t)p.n_tiles * p.cap * sizeof(StageEntry));
    p.off_ovf = o; o += up((size_t)p.ovf_cap * sizeof(StageEntry));
    p.off_rows = o; o += up((size_t)B * p.ns * K * 4);
    p.off_cs = o; o += up(m->scoring_type == AMDKGE_ROTATE ? (size_t)m->n_rels * K * 4 : 0);
    const size_t code_bytes = (size_t)B * eta * (ks / 4) * 4;
    p.codes = !det && m->scoring_type == AMDKGE_TRANSE && ks <= 512 && eta <= 65535 && code_bytes <= ((size_t)8 << 30);   // (det: sorted path, three-row form)
    p.off_codes = o; o += up(p.codes ? code_bytes : 0);
    p.total = o + 256;
    return true;
}

// One workspace serves steps of different batch sizes (fit() ends every epoch with a short batch), and the position of the
// bucket lists depends on the plan: the lists and staged rows of one plan may lie where another plan keeps its counters.  A
// plan's counters are zero when its step ends, so they only have to be re-zeroed when a step with a DIFFERENT geometry has
// used the buffer in between.  The library remembers, per workspace address, the geometry of the last step enqueued on it and
// clears the counter region on the step's stream when it changes (first sight of an address counts as a change; the
// caller's zero-fill contract covers a buffer that is freed and re-allocated at the same address).
// The memory is per (device, address) -- the same address on another GPU is another workspace -- and bounded: past
// PLAN_GUARD_MAX remembered workspaces (or on amdkge_release_scratch()) everything is forgotten, which only costs the next step on
// each workspace one memset.  The guard is HOST state: a step captured into a hipGraph and replayed does not pass through it, so
// a graph must not be replayed on a workspace that a step of another geometry has used since the capture (include/amdkge.h).
namespace {
constexpr size_t PLAN_GUARD_MAX = 4096;
std::mutex g_plan_mu;
std::unordered_map<uint64_t, uint64_t> g_plan_last[16];   // per device ordinal (mod 16): workspace address -> geometry signature
}
void release_plan_guard() {   // (kge::, called by amdkge_release_scratch)
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (auto& mp : g_plan_last) mp.clear();
}
static int plan_guard(const void* d_work, const TiledPlan& p, char* w, hipStream_t st) {
    const uint64_t sig = ((uint64_t)(uint32_t)p.n_tiles << 32) ^ ((uint64_t)p.off_cnt * 0x9E3779B97F4A7C15ull) ^ (uint64_t)(uint32_t)p.cap;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return set_error_hip(e, "hipGetDevice");
    bool changed;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto& last = g_plan_last[dev & 15];
        if (last.size() >= PLAN_GUARD_MAX) last.clear();
        const uint64_t key = (uint64_t)(uintptr_t)d_work ^ ((uint64_t)(dev >> 4) << 56);
        auto it = last.find(key);
        changed = it == last.end() || it->second != sig;
        if (changed) {
            try {
                last[key] = sig;
            } catch (...) {   // (out of host memory for the registry node: forget everything -- every workspace then pays one memset)
                last.clear();
            }
        }
    }
    if (changed)
        if (hipError_t e = hipMemsetAsync(w + p.off_cnt, 0, (size_t)(p.n_tiles + 2) * 32 * 4, st)) return set_error_hip(e, "hipMemsetAsync(tile counters)");
    return AMDKGE_OK;
}

template <int MODEL, int CH, int UNROLL, bool DET = false>
static int launch_tile(const TileArgs& a, size_t shmem, hipStream_t st) {
    static PerDeviceOnce attr;
    if (attr.need()) {
        if (hipError_t e = hipFuncSetAttribute((const void*)tile_backward_kernel<MODEL, CH, UNROLL, DET>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256))   // (the kernel has a few bytes of static LDS)
            return set_error_hip(e, "hipFuncSetAttribute(tile_backward)");
        attr.done();
    }
    hipLaunchKernelGGL((tile_backward_kernel<MODEL, CH, UNROLL, DET>), dim3(a.n_tiles + a.rel_blocks), dim3(TILE_THREADS), shmem, st, a);
    return check_launch("tile_backward");
}

template <int MODEL, int W, int CHF, bool DET>
static int launch_forward_v(TrainArgs& f, hipStream_t st) {
    constexpr int slots = 4 / W;
    // LDS: per-slot score / id arrays, per-slot loss, and the transpose rows of emit_row (one per wave, or one per
    // workgroup when a positive spans the whole workgroup)
    size_t shmem = (size_t)slots * slot_lds_bytes(f.eta, W) + slots * sizeof(double) + (W == 1 ? 4 : 1) * (size_t)f.K * 4;
    f.sign_off = (int)shmem;
    if (W != 1 || CHF != 1) shmem += sign_stash_bytes(MODEL, f.eta, CHF);   // (one wave per positive, one quad per lane: TransE takes the single-pass form, no stash)
    if (shmem > 64 * 1024) {
        static PerDeviceOnce attr;
        if (attr.need()) {
            if (hipError_t e = hipFuncSetAttribute((const void*)train_fwdbwd_kernel<MODEL, 4, W, CHF, true, DET>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
                return set_error_hip(e, "hipFuncSetAttribute(train_forward_stage)");
            attr.done();
        }
    }
    const unsigned grid = KGE_DBG(f, 8192) ? 0u : (unsigned)((f.B + slots - 1) / slots);   // (ablation 8192: no forward launch)
    if (grid) hipLaunchKernelGGL((train_fwdbwd_kernel<MODEL, 4, W, CHF, true, DET>), dim3(grid), dim3(256), shmem, st, f);
    return check_launch("train_forward_stage");
}
template <int MODEL, int W, int CHF>
static int launch_forward(TrainArgs& f, hipStream_t st) {
    return f.det ? launch_forward_v<MODEL, W, CHF, true>(f, st) : launch_forward_v<MODEL, W, CHF, false>(f, st);
}

// C of the column-sharded step (kge_train_cols.h): the coefficients are given, the staging protocol is the forward kernel's
template <int MODEL, int G>
static int launch_cols_stage(const TrainArgs& f, float* given, hipStream_t st) {
    ColsArgs ca{f, given};
    const size_t sh = cols_stage_lds(G, f.eta, f.K);
    if (sh > 64 * 1024) return set_error(AMDKGE_EUNSUPPORTED, "train_step_tiled(GIVEN_COEFFS): eta too large for the column-sharded stage kernel");
    const unsigned grid = (unsigned)((f.B + 256 / G - 1) / (256 / G));
    if (grid) hipLaunchKernelGGL((cols_stage_kernel<MODEL, G>), dim3(grid), dim3(256), sh, st, ca);
    return check_launch("cols_stage");
}

template <int MODEL>
static int run_tiled(TrainArgs& f, TileArgs& te, hipStream_t st, float* given = nullptr) {
    constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
    // F: forward + staging.  Rows of up to 128 quads: one wave per positive (1 or 2 quads per lane); longer rows
    // (k <= 2048): the four waves of a workgroup share one positive.
    int rc;
    if (given) rc = f.nq <= 16 ? launch_cols_stage<MODEL, 16>(f, given, st) : (f.nq <= 32 ? launch_cols_stage<MODEL, 32>(f, given, st) : launch_cols_stage<MODEL, 64>(f, given, st));
    else if (f.nq <= 64) rc = launch_forward<MODEL, 1, 1>(f, st);
    else if (f.nq <= 128) rc = launch_forward<MODEL, 1, 2>(f, st);
    else if (f.nq <= 256) rc = launch_forward<MODEL, 4, 1>(f, st);
    else rc = launch_forward<MODEL, 4, 2>(f, st);
    if (rc) return rc;
    if (te.det && f.B > 0) {   // deterministic mode: relation-row gradient from the staged fifth rows, in batch order
        hipLaunchKernelGGL(rel_backward_det_kernel, dim3((unsigned)te.n_rels), dim3(256), 0, st, f.triples, f.B, f.stage_rows, f.ns, f.K, f.g_rel);
        if ((rc = check_launch("rel_backward_det"))) return rc;
    }
    // T: entity tiles (the owner applies the optimizer)
    te.gw = f.nq <= 128 ? 1 : (f.nq <= 256 ? 4 : 8);
    if (te.direct) {
        const size_t sh = direct_lds_bytes(te.cap, te.tile_rows);
        if (f.nq <= 128) return set_error(AMDKGE_EUNSUPPORTED, "tile_direct: rows of up to 128 quads per half take the LDS-accumulator tiles");
        if (te.gw == 4) hipLaunchKernelGGL((tile_direct_kernel<MODEL, 4>), dim3(te.n_tiles + te.rel_blocks), dim3(256), sh, st, te);
        else hipLaunchKernelGGL((tile_direct_kernel<MODEL, 8>), dim3(te.n_tiles + te.rel_blocks), dim3(512), sh, st, te);
        return check_launch("tile_direct");
    }
    const size_t shmem_t = te.det ? (((size_t)te.tile_rows * te.K * 4 + (size_t)te.tile_rows * te.gw + 15) & ~(size_t)15) + (size_t)te.sort_cap * 20 + 1024
                                  : (((size_t)te.tile_rows * te.K * 4 * (te.own_cache ? 2 : 1) + (te.lazy ? (size_t)te.tile_rows * te.gw : 0) + 15) & ~(size_t)15) +
                                        (tile_queued(MODEL, tile_ch_of(f.nq), te.K) ? TILE_QUEUE_BYTES : 0);
    // entries in flight per wave: bounded by the 128 VGPRs a 1024-thread workgroup leaves per lane (TransE holds three operand
    // rows per entry, RotatE two complex ones)
    constexpr int U1 = TRILINEAR ? 8 : 4;
    constexpr int U2 = TRILINEAR ? 4 : 2;
    if constexpr (MODEL == AMDKGE_ROTATE) {
        if (te.det) {
            if (f.nq <= 64 || f.nq > 128) return launch_tile<MODEL, 1, U1, true>(te, shmem_t, st);
            return launch_tile<MODEL, 2, U2, true>(te, shmem_t, st);
        }
    }
    if (f.nq <= 64 || f.nq > 128) return launch_tile<MODEL, 1, U1>(te, shmem_t, st);
    return launch_tile<MODEL, 2, U2>(te, shmem_t, st);
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_set_tile_direct(int on) {
    g_tile_direct = on ? 1 : 0;   // (2 selected round 5's short-row form, since removed: it now means 1)
    return AMDKGE_OK;
}

extern "C" int64_t amdkge_train_tiled_workspace_bytes(const amdkge_model* m, int64_t B, int32_t eta) {
    if (validate_model(m) != AMDKGE_OK || B < 0 || B >= (1ll << 30) || eta < 1) return -1;
    TiledPlan p, pd;
    if (!make_plan(m, B, eta, p)) return 0;   // 0 = shape not supported by the owner-computes path
    // one buffer serves both modes: the deterministic plan (five staged rows, smaller tiles) is the larger one where it exists
    size_t det_total = make_plan(m, B, eta, pd, true) ? pd.total : 0;
    if (make_plan(m, B, eta, pd, true, true) && pd.total > det_total) det_total = pd.total;   // (AMDKGE_TILED_DET_WIDE_SORT: smaller tiles, more of them)
    return (int64_t)(p.total > det_total ? p.total : det_total);
}

extern "C" int amdkge_train_step_tiled(const amdkge_model* m, const amdkge_loss* loss, const amdkge_opt* opt,
                                       float* d_ent, float* d_rel, float* d_ent_slot0, float* d_ent_slot1,
                                       float* d_rel_slot0, float* d_rel_slot1, float rel_reg_lambda,
                                       const int32_t* d_triples, int64_t B, int32_t eta, int64_t sample_base,
                                       int64_t sample_range, uint64_t seed, uint64_t step, int64_t row_offset,
                                       int64_t b_global, const int32_t* d_neg_override, float* d_grad_ent,
                                       float* d_grad_rel, int32_t apply_update, int32_t flags, double* d_loss_sum, double* d_reg_loss,
                                       float* d_pos_scores, float* d_neg_scores, void* d_work, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (int rc = validate_opt(opt)) return rc;
    if (!loss || loss->kind < 0 || loss->kind > AMDKGE_LOSS_MULTICLASS_NLL) return set_error(AMDKGE_EINVAL, "train_step_tiled: unknown loss kind");
    if (loss->focus_nonlinearity < AMDKGE_FOCUS_OFF || loss->focus_nonlinearity > AMDKGE_FOCUS_SOFTPLUS || (loss->focus_nonlinearity && !loss->d_focus_w))
        return set_error(AMDKGE_EINVAL, "train_step_tiled: bad FocusE settings (unknown non-linearity or NULL weights)");
    if (!d_ent || !d_rel || !d_grad_rel || !d_loss_sum || !d_work) return set_error(AMDKGE_EINVAL, "train_step_tiled: NULL pointer");
    if (!d_grad_ent && (!apply_update || (flags & AMDKGE_TILED_POS_ATOMIC))) return set_error(AMDKGE_EINVAL, "train_step_tiled: d_grad_ent is required unless the step updates in place with staged positives");
    if (B < 0 || B >= (1ll << 30) || eta < 1) return set_error(AMDKGE_EINVAL, "train_step_tiled: B must be in [0, 2^30) and eta >= 1");
    const bool det = (flags & AMDKGE_TILED_DETERMINISTIC) != 0;
    if (det && (flags & AMDKGE_TILED_POS_ATOMIC)) return set_error(AMDKGE_EINVAL, "train_step_tiled: DETERMINISTIC excludes POS_ATOMIC (atomics add in arrival order)");
    const bool given = (flags & AMDKGE_TILED_GIVEN_COEFFS) != 0;
    if (given) {   // column-sharded step, phase C: d_pos_scores / d_neg_scores carry dL/dscore IN (amdkge_cols_loss)
        if (flags & (AMDKGE_TILED_DETERMINISTIC | AMDKGE_TILED_POS_ATOMIC | AMDKGE_TILED_HOT_ROWS)) return set_error(AMDKGE_EUNSUPPORTED, "train_step_tiled: GIVEN_COEFFS excludes DETERMINISTIC / POS_ATOMIC / HOT_ROWS");
        if (loss->focus_nonlinearity) return set_error(AMDKGE_EUNSUPPORTED, "train_step_tiled: GIVEN_COEFFS with FocusE is not offered (fold the weights into the coefficients)");
        if (!d_pos_scores || d_neg_scores != d_pos_scores + B) return set_error(AMDKGE_EINVAL, "train_step_tiled: GIVEN_COEFFS needs one coefficient buffer: d_pos_scores [B], d_neg_scores = d_pos_scores + B [eta][B]");
        if (stored_k(m) > 256) return set_error(AMDKGE_EUNSUPPORTED, "train_step_tiled: GIVEN_COEFFS serves column slices of up to 256 stored units per half");
    }
    TiledPlan p;
    if (!make_plan(m, B, eta, p, det, det && (flags & AMDKGE_TILED_DET_WIDE_SORT) != 0))
        return set_error(AMDKGE_EUNSUPPORTED, "train_step_tiled: shape not supported (stored half width not a multiple of 4 -- set k_pad = amdkge_padded_k(k) --, > 2048, or eta too large); use amdkge_train_fwdbwd + amdkge_opt_step");
    if (apply_update) {
        if (opt_nslots(opt->kind) >= 1 && !d_ent_slot0) return set_error(AMDKGE_EINVAL, "train_step_tiled: optimizer slot 0 is NULL");
        if (opt_nslots(opt->kind) == 2 && !d_ent_slot1) return set_error(AMDKGE_EINVAL, "train_step_tiled: optimizer slot 1 is NULL");
    }
    const bool d_rel_slot_ok = (opt_nslots(opt->kind) < 1 || d_rel_slot0) && (opt_nslots(opt->kind) < 2 || d_rel_slot1);
    if (apply_update && !d_rel_slot_ok && (d_rel_slot0 || d_rel_slot1))
        return set_error(AMDKGE_EINVAL, "train_step_tiled: relation optimizer slots incomplete for this optimizer");
    if (B > 0 && !d_triples) return set_error(AMDKGE_EINVAL, "train_step_tiled: null triples");
    if (!d_neg_override && (sample_range <= 0 || sample_range > 0xFFFFFFFFll || sample_base < 0 || sample_base + sample_range > m->n_ents))
        return set_error(AMDKGE_EINVAL, "train_step_tiled: sampling range outside the entity table");
    const int ks = stored_k(m), K = row_floats(m);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    if (int rc = plan_guard(d_work, p, w, st)) return rc;
    int* counters = (int*)(w + p.off_cnt);
    StageEntry* lists = (StageEntry*)(w + p.off_lists);
    StageEntry* ovf = (StageEntry*)(w + p.off_ovf);
    float* stage_rows = (float*)(w + p.off_rows);
    float* rel_cs = (float*)(w + p.off_cs);
    const bool lazy = opt->lazy != 0;
    uint8_t* touched = (lazy && apply_update && (flags & AMDKGE_TILED_POS_ATOMIC)) ? (uint8_t*)(w + p.off_touch) : nullptr;

    TrainArgs f{};
    f.ent = d_ent; f.rel = d_rel; f.triples = d_triples; f.neg_override = d_neg_override;
    f.g_ent = d_grad_ent; f.g_rel = d_grad_rel; f.pos_atomic = (flags & AMDKGE_TILED_POS_ATOMIC) ? 1 : 0; f.loss_sum = d_loss_sum; f.pos_scores = d_pos_scores; f.neg_scores = d_neg_scores;
    f.B = B; f.eta = eta; f.k = ks; f.K = K; f.k_live = m->k; f.nq = ks / 4;
    f.sc = SampleCfg{sample_base, (uint32_t)sample_range, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                     (uint32_t)(step >> 32), row_offset, b_global > 0 ? b_global : B};
    f.mc = model_const(m); f.loss = *loss;
    const bool hot = (flags & AMDKGE_TILED_HOT_ROWS) && !det && !(flags & AMDKGE_TILED_POS_ATOMIC);
    f.hot_map = hot ? (const uint8_t*)(w + p.off_hot_map) : nullptr; f.hot_buf = (float*)(w + p.off_hot_buf);
    f.rel_cs = (m->scoring_type == AMDKGE_ROTATE && B > 0) ? rel_cs : nullptr;   // filled by rel_phase_kernel below, before F
    f.touched = touched; f.ns = p.ns; f.det = det ? 1 : 0;
    f.sign_codes = (p.codes && !given) ? (uint32_t*)(w + p.off_codes) : nullptr;   // (the column-sharded stage kernel hands TransE tiles the three-row form)
    if (given) { f.pos_scores = nullptr; f.neg_scores = nullptr; }
    f.loss_parts = (double*)(w + p.off_loss);
    f.stage_rows = stage_rows; f.st_lists = lists; f.st_ovf = ovf; f.st_counters = counters;
    f.st_tile_rows = p.tile_rows; f.st_n_tiles = p.n_tiles; f.st_cap = p.cap; f.st_ovf_cap = p.ovf_cap; f.st_rb = p.rb;
#ifdef KGE_ABLATE
    { const char* e = getenv("AMDKGE_DEBUG"); f.dbg = e ? atoi(e) : 0; }
#endif

    TileArgs te{};
    te.x = d_ent; te.s0 = d_ent_slot0; te.s1 = d_ent_slot1; te.g_ent = d_grad_ent; te.apply_update = apply_update ? 1 : 0; te.pos_atomic = (flags & AMDKGE_TILED_POS_ATOMIC) ? 1 : 0; te.rel = d_rel;
    te.rel_cs = rel_cs; te.lazy = lazy ? 1 : 0; te.touched = touched; te.ns = p.ns; te.det = det ? 1 : 0; te.sort_cap = p.sort_cap;
    te.pos_bits = 0;
    while (te.pos_bits < 31 && ((int64_t)1 << te.pos_bits) < B) ++te.pos_bits;
    te.sign_codes = f.sign_codes; te.eta = eta; te.own_cache = p.own_cache ? 1 : 0; te.direct = p.direct ? 1 : 0;
    te.n_rels = m->n_rels; te.loss_parts = f.loss_parts; te.loss_sum = d_loss_sum; te.hot_map = f.hot_map; te.hot_buf = f.hot_buf;
#ifdef KGE_ABLATE
    te.dbg = f.dbg;
#endif
    te.triples = d_triples; te.stage_rows = stage_rows; te.lists = lists; te.ovf = ovf; te.counters = counters; te.status_flag = (int*)(w + p.off_flag);
    te.reg_loss = d_reg_loss; te.n_rows = m->n_ents; te.k = ks; te.K = K; te.k_live = m->k; te.nq = ks / 4;
    te.tile_rows = p.tile_rows; te.n_tiles = p.n_tiles; te.cap = p.cap; te.ovf_cap = p.ovf_cap; te.rb = p.rb; te.mc = f.mc;
    fill_opt_args(te.opt, opt);
    // Whole step in two launches when nothing in the tile pass reads the live relation table and the tables are updated in
    // place: the relation sweep rides in extra workgroups of the tile kernel.  That holds for the trilinear models (their
    // entries need the staged side row only) and for RotatE (staged side row, already rotated, + the owner's row; the forward
    // kernel read the relation through the per-step phase table).  TransE tiles read live relation rows (the three-row form of
    // an entry: all of them in deterministic mode, the near-zero units otherwise), so its sweep stays a separate launch behind.
    const bool rel_here = apply_update && d_rel_slot_ok;
    // (the touched-rows relation sweep is row-wise: it runs as its own launch behind the tiles)
    const bool fuse_rel = rel_here && !lazy && m->scoring_type != AMDKGE_TRANSE;
    te.rel_blocks = 0;
    if (rel_here) {
        te.rel_opt = te.opt;
        te.rel_opt.x = d_rel; te.rel_opt.g = d_grad_rel; te.rel_opt.s0 = d_rel_slot0; te.rel_opt.s1 = d_rel_slot1;
        te.rel_opt.n = (int64_t)m->n_rels * K; te.rel_opt.reg_loss = d_reg_loss;
        set_reg_terms(te.rel_opt, opt->rel_reg_p > 0 ? opt->rel_reg_p : opt->reg_p, rel_reg_lambda, opt->rel_reg2_p, opt->rel_reg2_lambda);
        if (fuse_rel) {
            const int64_t n4 = (te.rel_opt.n + 3) / 4;
            te.rel_blocks = (int)((n4 + TILE_THREADS - 1) / TILE_THREADS < 64 ? (n4 + TILE_THREADS - 1) / TILE_THREADS : 64);
        }
    }
    // B == 0 still runs the tiles: with no gradient the optimizer sweep must decay the slots / apply the
    // regulariser exactly like the dense path does
    int rc;
    if (m->scoring_type == AMDKGE_ROTATE && B > 0) {
        const int64_t nel = (int64_t)m->n_rels * ks;
        hipLaunchKernelGGL(rel_phase_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, d_rel, (int64_t)m->n_rels, ks, K, f.mc, rel_cs);
        if ((rc = check_launch("rel_phase"))) return rc;
    }
    switch (m->scoring_type) {
        case AMDKGE_TRANSE: rc = run_tiled<AMDKGE_TRANSE>(f, te, st, given ? d_pos_scores : nullptr); break;
        case AMDKGE_DISTMULT: rc = run_tiled<AMDKGE_DISTMULT>(f, te, st, given ? d_pos_scores : nullptr); break;
        case AMDKGE_ROTATE: rc = run_tiled<AMDKGE_ROTATE>(f, te, st, given ? d_pos_scores : nullptr); break;
        default: rc = run_tiled<AMDKGE_COMPLEX>(f, te, st, given ? d_pos_scores : nullptr); break;   // ComplEx, HolE (scale folded into dL/dscore)
    }
    if (rc != AMDKGE_OK || !rel_here || fuse_rel) return rc;
    amdkge_opt ro = *opt;
    ro.reg_lambda = rel_reg_lambda;
    if (opt->rel_reg_p > 0) ro.reg_p = opt->rel_reg_p;
    ro.reg2_p = opt->rel_reg2_p; ro.reg2_lambda = opt->rel_reg2_lambda;
    ro.row_floats = K;
    return amdkge_opt_step(&ro, d_rel, d_grad_rel, d_rel_slot0, d_rel_slot1, (int64_t)m->n_rels * K, d_reg_loss, stream);
}

extern "C" int amdkge_train_tiled_status(const amdkge_model* m, int64_t B, int32_t eta, int32_t flags, void* d_work, int32_t* status, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (!d_work || !status) return set_error(AMDKGE_EINVAL, "train_tiled_status: NULL pointer");
    *status = 0;
    TiledPlan p;
    if (!make_plan(m, B, eta, p, (flags & AMDKGE_TILED_DETERMINISTIC) != 0)) return set_error(AMDKGE_EUNSUPPORTED, "train_tiled_status: shape not supported");
    char* w = (char*)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    int* flag = (int*)(w + p.off_flag);   // the same place whatever B / flags the flagged step ran with
    hipStream_t st = (hipStream_t)stream;
    if (hipError_t e = hipMemcpyAsync(status, flag, sizeof(int), hipMemcpyDeviceToHost, st)) return set_error_hip(e, "hipMemcpyAsync(status)");
    if (hipError_t e = hipStreamSynchronize(st)) return set_error_hip(e, "hipStreamSynchronize");
    if (*status) { if (hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st)) return set_error_hip(e, "hipMemsetAsync(status)"); }
    return AMDKGE_OK;
}

namespace kge {
__global__ void hot_map_kernel(const int32_t* ids, int n, int64_t n_ents, uint8_t* map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0 && ids[i] < n_ents) map[ids[i]] = (uint8_t)(i + 1);
}
}  // namespace kge

extern "C" int amdkge_train_tiled_set_hot_rows(const amdkge_model* m, void* d_work, const int32_t* d_hot_ids, int32_t n_hot, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (!d_work || n_hot < 0 || n_hot > HOT_MAX || (n_hot > 0 && !d_hot_ids)) return set_error(AMDKGE_EINVAL, "set_hot_rows: bad arguments (at most 64 hot rows)");
    TiledPlan p;
    if (!make_plan(m, 1, 1, p)) return set_error(AMDKGE_EUNSUPPORTED, "set_hot_rows: shape not supported by the owner-computes path");
    char* w = (char*)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    hipStream_t st = (hipStream_t)stream;
    const int K = row_floats(m);
    if (hipError_t e = hipMemsetAsync(w + p.off_hot_map, 0, (size_t)m->n_ents, st)) return set_error_hip(e, "hipMemsetAsync(hot map)");
    if (hipError_t e = hipMemsetAsync(w + p.off_hot_buf, 0, (size_t)HOT_MAX * HOT_REPL * K * 4, st)) return set_error_hip(e, "hipMemsetAsync(hot replicas)");
    if (n_hot > 0) {
        hipLaunchKernelGGL(hot_map_kernel, dim3(1), dim3(64), 0, st, d_hot_ids, (int)n_hot, (int64_t)m->n_ents, (uint8_t*)(w + p.off_hot_map));
        return check_launch("set_hot_rows");
    }
    return AMDKGE_OK;
}


// ---- column-sharded step, phases A and B (kge_train_cols.h) -------------------------------------------------------------------------
template <int MODEL>
static int launch_cols_scores(const ColsArgs& ca, hipStream_t st) {
    const TrainArgs& f = ca.t;
    const int G = f.nq <= 16 ? 16 : (f.nq <= 32 ? 32 : 64);
    const size_t sh = cols_scores_lds(G, f.eta);
    if (sh > 64 * 1024) return set_error(AMDKGE_EUNSUPPORTED, "cols_partial_scores: eta too large");
    const unsigned grid = (unsigned)((f.B + 256 / G - 1) / (256 / G));
    if (G == 16) hipLaunchKernelGGL((cols_scores_kernel<MODEL, 16>), dim3(grid), dim3(256), sh, st, ca);
    else if (G == 32) hipLaunchKernelGGL((cols_scores_kernel<MODEL, 32>), dim3(grid), dim3(256), sh, st, ca);
    else hipLaunchKernelGGL((cols_scores_kernel<MODEL, 64>), dim3(grid), dim3(256), sh, st, ca);
    return check_launch("cols_scores");
}

extern "C" int amdkge_cols_partial_scores(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples, int64_t B, int32_t eta,
                                          int64_t sample_base, int64_t sample_range, uint64_t seed, uint64_t step, int64_t row_offset, int64_t b_global,
                                          const int32_t* d_neg_override, float* d_scores, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (B < 0 || B >= (1ll << 30) || eta < 1) return set_error(AMDKGE_EINVAL, "cols_partial_scores: B must be in [0, 2^30) and eta >= 1");
    if (B == 0) return AMDKGE_OK;
    if (!d_ent || !d_rel || !d_triples || !d_scores) return set_error(AMDKGE_EINVAL, "cols_partial_scores: NULL pointer");
    const int ks = stored_k(m), K = row_floats(m);
    if (ks % 4 != 0 || ks > 256) return set_error(AMDKGE_EUNSUPPORTED, "cols_partial_scores: column slices are stored padded (k_pad = amdkge_padded_k(k)) and hold up to 256 units per half");
    if (!d_neg_override && (sample_range <= 0 || sample_range > 0xFFFFFFFFll || sample_base < 0 || sample_base + sample_range > m->n_ents))
        return set_error(AMDKGE_EINVAL, "cols_partial_scores: sampling range outside the entity table");
    ColsArgs ca{};
    TrainArgs& f = ca.t;
    f.ent = d_ent; f.rel = d_rel; f.triples = d_triples; f.neg_override = d_neg_override;
    f.B = B; f.eta = eta; f.k = ks; f.K = K; f.k_live = m->k; f.nq = ks / 4;
    f.sc = SampleCfg{sample_base, (uint32_t)sample_range, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32), row_offset,
                     b_global > 0 ? b_global : B};
    f.mc = model_const(m);
    ca.scores = d_scores;
    hipStream_t st = (hipStream_t)stream;
    switch (m->scoring_type) {
        case AMDKGE_TRANSE: return launch_cols_scores<AMDKGE_TRANSE>(ca, st);
        case AMDKGE_DISTMULT: return launch_cols_scores<AMDKGE_DISTMULT>(ca, st);
        case AMDKGE_ROTATE: return launch_cols_scores<AMDKGE_ROTATE>(ca, st);
        default: return launch_cols_scores<AMDKGE_COMPLEX>(ca, st);   // ComplEx, HolE (the scale is applied to the complete sum, in amdkge_cols_loss)
    }
}

extern "C" int amdkge_cols_loss(const amdkge_model* m, const amdkge_loss* loss, float* d_scores, int64_t B, int32_t eta, double* d_loss_sum, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (!loss || loss->kind < 0 || loss->kind > AMDKGE_LOSS_MULTICLASS_NLL) return set_error(AMDKGE_EINVAL, "cols_loss: unknown loss kind");
    if (loss->focus_nonlinearity) return set_error(AMDKGE_EUNSUPPORTED, "cols_loss: FocusE is not offered in the column-sharded step");
    if (B < 0 || eta < 1) return set_error(AMDKGE_EINVAL, "cols_loss: bad sizes");
    if (B == 0) return AMDKGE_OK;
    if (!d_scores) return set_error(AMDKGE_EINVAL, "cols_loss: NULL pointer");
    const ModelConst mc = model_const(m);
    const unsigned grid = (unsigned)((B + 255) / 256 < 1024 ? (B + 255) / 256 : 1024);
    hipLaunchKernelGGL(cols_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_scores, B, (int)eta, *loss, mc.score_sign * mc.score_scale, d_loss_sum);
    return check_launch("cols_loss");
}
