// evaluate(), distance models (TransE / RotatE): EXACT early exit of the 1-vs-all count pass.
// Included by kge_rank.hip (namespace kge) behind the tile kernels and kge_rank_screen.h, whose ScreenBufs it reuses.
//
// The declared score of a (query i, entity j) pair is  -acc_U,  acc_u = fl(acc_{u-1} + t_u)  in unit order, with  t_u = |q_u +- e_u|
// (TransE.py:77-83,107-113) or the correctly rounded modulus of RotatE's unit (RotatE.py:151-160,209-214) -- every t_u >= 0, so the
// fp32 partial sums are NON-DECREASING (round-to-nearest is monotone and acc + t >= acc exactly) and
//     q(u) = int(sgn_scale * acc_u * 1000),   sgn_scale < 0,
// is non-increasing in u.  A rank only asks whether  q(U) > qp  or  q(U) == qp  (qp: the positive's quantised score): once
// q(u) < qp at ANY u the pair is decided -- it counts for neither -- with no error bound involved.  Per query the test is one
// compare:  acc_u > T_i,  T_i = the largest fp32 acc with q(acc) >= qp (bisection over the ordered bit patterns with the kernels'
// own quantise()).  On a trained model the positive scores near the top and almost every candidate is decided after a
// fraction of its units.
//
// The tile kernels keep their geometry (64 queries x 64 entities per workgroup, 4 x 4 pairs per thread, units streamed through
// LDS): every `check` stages the workgroup counts its undecided pairs (one compare per pair, a wave reduction, published by the
// stage's own barrier); when few enough are left that recomputing them costs less than finishing the tile, they are appended to a
// list and the tile ENDS.  rank_recheck_dist_kernel then walks the listed pairs' full chains -- the same operations in the same
// order (rank_op / rot_exact_op: what the filter kernel uses) -- and counts them.  Counts are therefore bit-identical to
// rank_count_kernel / rank_rot_kernel whatever the tables hold:
//   * rows with a unit that is not finite or is huge (|x| >= 2^60: products could overflow to inf and inf - inf = NaN would
//     break monotonicity) are flagged by rank_rowflags_kernel and their pairs are never decided early;
//   * RotatE's fast modulus is exact only for x >= 2^-100: a workgroup that has met a smaller x (gmax > 2^50) does not exit
//     early; its tile is redone with libm's sqrtf as before;
//   * a full list raises the device flag and the plain kernel, guarded by that flag, redoes the call (no host round trip).
// Two parts (no include guard): part 1 -- what the tile kernels need, included BEFORE them; part 2 (KGE_RANK_EARLY_PART2) -- the
// workspace, the row-flag and recheck kernels, included behind kge_rank_screen.h (ScreenBufs) and rot_exact_op.
#ifndef KGE_RANK_EARLY_PART2

namespace kge {

constexpr int EARLY_STAGE_PAIRS = 512;   // undecided pairs a tile may hand over (LDS staging)

struct EarlyCfg {
    int on = 1;
    int check_l1 = 4, check_rot = 1;   // stages (of KT = 16 units) between two checks.  A check is ~15 % of a TransE stage (1.5 issue slots per
                                       // pair and unit) and ~2 % of a RotatE stage: measured at the C2 shape on planted tables
                                       // (profiles/r04d_distance_models_sweep.jsonl) TransE 1.25 ms with 4, 1.56 - 1.96 with 2 or 1;
                                       // RotatE 6.2 - 6.3 ms with 1, 6.6 - 6.8 with 2
    int cost = 16;                     // a re-checked pair's chain costs about this many tile-kernel pair chains (measured ~10: one lane
                                       // per pair against a 4 x 4 register tile; handing over 2 % of the pairs costs what it saves)
    int probe = 1;                     // 0: the early-exit kernel always does the work (tests)
};
static EarlyCfg g_early;

// The PROBE (rank_early_probe_kernel, part 2) samples 4 096 (query, candidate) pairs and counts those already decided at half
// their units; probe[0] = decided, probe[1] = sampled.  On tables whose positives do not stand out (an untrained model: nothing
// is decided before the last units) the early-exit kernel would only pay for its checks and its lower occupancy (measured: TransE
// k = 200, 12 % slower than the plain kernel), so the device decides which of the two kernels of the call does the work -- both
// are launched, one returns at once, no host round trip.
__device__ __forceinline__ bool early_probe_says_yes(const int* probe) { return probe[0] * 2 >= probe[1] && probe[1] > 0; }
enum { GUARD_NONE = 0, GUARD_FLAG = 1 /* run iff *guard != 0 */, GUARD_EARLY = 2 /* run iff the probe says yes */,
       GUARD_EARLY_FALLBACK = 3 /* run iff the probe says no, or *guard (the list overflowed) != 0 */ };
__device__ __forceinline__ bool guard_says_run(int mode, const int* guard, const int* probe) {
    if (mode == GUARD_FLAG) return *guard != 0;
    if (mode == GUARD_EARLY) return early_probe_says_yes(probe);
    if (mode == GUARD_EARLY_FALLBACK) return *guard != 0 || !early_probe_says_yes(probe);
    return true;
}

// T = the largest fp32 acc >= 0 with quantise(sgn_scale * acc) >= qp; -1 when not even acc = 0 reaches qp (every pair of the
// query is decided at once), +inf when every acc does (never decided).  quantise(sgn_scale * .) is non-increasing (sgn_scale < 0).
__device__ __forceinline__ float early_threshold(int qp, float sgn_scale) {
    if (!(quantise(sgn_scale * 0.f) >= qp)) return -1.f;
    if (quantise(sgn_scale * INFINITY) >= qp) return INFINITY;
    uint32_t lo = 0u, hi = 0x7f800000u;   // P(lo) holds, P(hi) does not
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (quantise(sgn_scale * __uint_as_float(mid)) >= qp) lo = mid; else hi = mid;
    }
    return __uint_as_float(lo);
}

// ---- what the tile kernels do at a check point (shared by rank_count_kernel<L1> and rank_rot_kernel) ---------------------------
struct EarlyShared {
    int red[4];      // per wave: undecided pairs | (1 << 30 if the wave has met an out-of-domain modulus)
    int n;           // staging fill
    int base;        // list position reserved for this tile
    int2 pairs[EARLY_STAGE_PAIRS];
};

// the workgroup's decision at a check point, after the barrier that published red[]: end this tile now?
__device__ __forceinline__ bool early_decide(const EarlyShared& s, int units_done, int U, int cost, int& total) {
    const int r = s.red[0] | s.red[1] | s.red[2] | s.red[3];
    total = (s.red[0] & 0xFFFFF) + (s.red[1] & 0xFFFFF) + (s.red[2] & 0xFFFFF) + (s.red[3] & 0xFFFFF);
    if (r & (1 << 30)) return false;
    if (total > EARLY_STAGE_PAIRS) return false;
    // re-checking `total` pairs costs ~cost x total full chains; finishing the tile costs 4096 x (U - done) / U
    return (int64_t)total * cost * U <= (int64_t)(U - units_done) * (QT * ET);
}

// hand the undecided pairs of this thread (bit 4 x + y of `und`) to the list; called by every thread of the workgroup
struct EarlyList { int* counter; int2* pairs; int64_t cap; };   // counter: [0] pairs appended, [1] overflow flag, [2] tiles ended early
__device__ __forceinline__ void early_spill(EarlyShared& s, const EarlyList& b, uint32_t und, int total, int64_t q_first, int64_t cand_first) {
    const int c = __popc(und);
    if (c) {
        int pos = atomicAdd(&s.n, c);
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (und & (1u << (4 * x + y))) s.pairs[pos++] = make_int2((int)(q_first + x), (int)(cand_first + y));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int base = total ? atomicAdd(&b.counter[0], total) : 0;
        s.base = base;
        if ((int64_t)base + total > b.cap || base < 0) atomicExch(&b.counter[1], 1);
        atomicAdd(&b.counter[2], 1);
        s.n = 0;
    }
    __syncthreads();
    const int64_t base = s.base;
    if (base >= 0 && base + total <= b.cap)
        for (int i = threadIdx.x; i < total; i += 256) b.pairs[base + i] = s.pairs[i];
    __syncthreads();
}


}  // namespace kge

#else   // ---------------------------------------------------------------- part 2

namespace kge {


// fixed part of the early-exit workspace: counters | this call's counts | row flags (queries, candidates)
static inline size_t early_fixed_bytes(int64_t n, int64_t m) {
    return 256 + scr_up((size_t)n * 8) + scr_up((size_t)n) + scr_up((size_t)m);
}

struct EarlyBufs {
    ScreenBufs b;        // counter ([0] pairs appended, [1] overflow flag, [2] tiles that ended early, [4] / [5] the probe's decided / sampled), counts, pairs, cap
    uint8_t* qbad;       // [n] query row holds a non-finite / huge value
    uint8_t* ebad;       // [m] candidate row (by position) does
};

static inline EarlyBufs carve_early(void* d_screen, size_t bytes, int64_t n, int64_t m) {
    EarlyBufs e{};
    char* p = (char*)(((uintptr_t)d_screen + 255) & ~(uintptr_t)255);
    const char* end = (char*)d_screen + bytes;
    e.b.counter = (int*)p; p += 256;
    e.b.counts = (int32_t*)p; p += scr_up((size_t)n * 8);
    e.qbad = (uint8_t*)p; p += scr_up((size_t)n);
    e.ebad = (uint8_t*)p; p += scr_up((size_t)m);
    e.b.pairs = (int2*)p;
    e.b.cap = end > p ? (int64_t)((end - p) / 8) : 0;
    return e;
}

// one wave per row: flag[r] = the row holds a value that is not finite or not below 2^60 in magnitude (width floats, whole float4s)
__global__ __launch_bounds__(256) void rank_rowflags_kernel(const float* __restrict__ table, int64_t stride, const int32_t* __restrict__ ids, int64_t lo,
                                                            int64_t nrows, int width, uint8_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t id = ids ? (int64_t)ids[lo + r] : lo + r;
    const float4* row = reinterpret_cast<const float4*>(table + id * stride);
    bool bad = false;
    for (int q = lane; q < (width >> 2); q += 64) {
        const float4 t = row[q];
        bad |= !(fabsf(t.x) < 0x1p60f) || !(fabsf(t.y) < 0x1p60f) || !(fabsf(t.z) < 0x1p60f) || !(fabsf(t.w) < 0x1p60f);
    }
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) flag[r] = any ? 1 : 0;
}

// ---- the probe: 64 queries x 64 candidates spread over the call, one pair per thread, the chain up to half the units --------------
struct ProbeArgs {
    const float* ent;
    const float* Q;
    const int* qpos;
    const int32_t* ent_ids;
    int64_t ent_lo, m, n;
    RankGeom g;
    float sgn_scale;
    int* probe;   // [0] += decided, [1] += sampled
};

template <int MODE>
__global__ __launch_bounds__(256) void rank_early_probe_kernel(ProbeArgs a) {
    constexpr int NQF = ModeTraits<MODE>::NQF, NEF = ModeTraits<MODE>::NEF;
    const int t = blockIdx.x * 256 + threadIdx.x;   // 4 096 threads
    const int64_t qi = ((int64_t)(t >> 6) * a.n) / 64, cj = ((int64_t)(t & 63) * a.m) / 64;
    const int64_t pos = a.ent_lo + cj;
    const float* qrow = a.Q + qi * (int64_t)a.g.QW;
    const float* erow = a.ent + (a.ent_ids ? (int64_t)a.ent_ids[pos] : pos) * a.g.K;
    const float thr = early_threshold(a.qpos[qi], a.sgn_scale);
    const int half = ((a.g.U / 2) + 3) & ~3;
    float acc = 0.f;
    // (four quads per trip, all their loads issued before the first is used: the probe sits on the call's critical path and a
    // thread's loads do not depend on its running sum -- one quad at a time cost ~25 dependent round trips to rows nobody has
    // touched yet, ~50 us)
    for (int u0 = 0; u0 < half; u0 += 16) {
        float4 q4[4][NQF], e4[4][NEF];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int uu = min(u0 + 4 * t, half - 4);   // (past the end: a re-read that is not used)
#pragma unroll
            for (int f = 0; f < NQF; ++f) q4[t][f] = *reinterpret_cast<const float4*>(qrow + (int64_t)f * a.g.qplane + uu);
#pragma unroll
            for (int f = 0; f < NEF; ++f) e4[t][f] = *reinterpret_cast<const float4*>(erow + (int64_t)f * a.g.eplane + uu);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u0 + 4 * t + u < half && u0 + 4 * t + u < a.g.U) {
                    float qq[NQF], ee[NEF];
#pragma unroll
                    for (int f = 0; f < NQF; ++f) qq[f] = (&q4[t][f].x)[u];
#pragma unroll
                    for (int f = 0; f < NEF; ++f) ee[f] = (&e4[t][f].x)[u];
                    if constexpr (MODE == MODE_ROT_O || MODE == MODE_ROT_S) acc = rot_exact_op<MODE>(acc, qq, ee);
                    else acc = rank_op<MODE>(acc, qq, ee, a.g.sgn);
                }
            }
        }
    }
    const unsigned long long dec = __ballot(acc > thr);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&a.probe[0], __popcll(dec)); atomicAdd(&a.probe[1], 64); }
}

// the early path's counts join the caller's (+=) when it did the work: the probe said yes and the list did not overflow
__global__ void rank_early_merge_kernel(ScreenBufs b, int64_t n, int32_t* __restrict__ counts) {
    if (b.counter[1] != 0 || !early_probe_says_yes(b.counter + 4)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n && b.counts[i]) counts[i] += b.counts[i];
}

// ---- exact recheck of the listed pairs: one lane per pair, the declared chain of the mode ---------------------------------------
struct RecheckDistArgs {
    const float* ent;
    const float* Q;
    const int* qpos;
    const int32_t* ent_ids;
    int64_t ent_lo;
    RankGeom g;
    float sgn_scale;
    ScreenBufs b;
};

constexpr int RD_CH = 16;              // units per chunk
constexpr int RD_LD = RD_CH + 4;       // LDS row stride in floats (80 bytes: 16 lanes walking their own rows hit 16 distinct bank quads)
template <int MODE> constexpr size_t rd_lds_bytes() { return (size_t)4 * (ModeTraits<MODE>::NQF + ModeTraits<MODE>::NEF) * 64 * RD_LD * sizeof(float); }

// As rank_recheck_kernel (kge_rank_screen.h): a wave fetches RD_CH-unit chunks of its 64 query rows and 64 entity rows COALESCED
// (16 rows x one 64-byte piece per instruction and plane), parks them in its private LDS region, and every lane then walks its
// own pair's chunk in unit order; the next chunk is in flight meanwhile.  Live units only (U), pieces loaded whole (the stored
// planes are whole float4s).
template <int MODE>
__global__ __launch_bounds__(256) void rank_recheck_dist_kernel(RecheckDistArgs a) {
    constexpr int NQF = ModeTraits<MODE>::NQF, NEF = ModeTraits<MODE>::NEF;
    extern __shared__ __attribute__((aligned(16))) char smem_rd[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Qb = reinterpret_cast<float*>(smem_rd) + (size_t)wv * (NQF + NEF) * 64 * RD_LD;
    float* Eb = Qb + (size_t)NQF * 64 * RD_LD;
    if (a.b.counter[1] || !early_probe_says_yes(a.b.counter + 4)) return;   // the list overflowed / the probe chose the plain kernel: it does the whole call
    const int64_t npairs = min((int64_t)a.b.counter[0], a.b.cap);
    const int64_t ngroups = (npairs + 63) / 64;
    const int lrow = lane >> 2, lpc = lane & 3;   // loader: 16 rows per instruction, 4 16-byte pieces per row chunk
    const int U = a.g.U;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        const int64_t p = grp * 64 + lane;
        const bool have = p < npairs;
        const int2 pr = a.b.pairs[have ? p : npairs - 1];
        const int64_t pos = a.ent_lo + pr.y;
        const int64_t qoff = (int64_t)pr.x * a.g.QW, eoff = (a.ent_ids ? (int64_t)a.ent_ids[pos] : pos) * a.g.K;
        int64_t qo[4], eo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { qo[i] = __shfl(qoff, 16 * i + lrow, 64) + 4 * lpc; eo[i] = __shfl(eoff, 16 * i + lrow, 64) + 4 * lpc; }
        float4 rq[NQF][4], re[NEF][4];
        auto fetch = [&](int u0) __attribute__((always_inline)) {
            const bool in = u0 + 4 * lpc < U;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int f = 0; f < NQF; ++f)
                    rq[f][i] = in ? *reinterpret_cast<const float4*>(a.Q + qo[i] + (int64_t)f * a.g.qplane + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int f = 0; f < NEF; ++f)
                    re[f][i] = in ? *reinterpret_cast<const float4*>(a.ent + eo[i] + (int64_t)f * a.g.eplane + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float acc = 0.f;
        fetch(0);
        for (int u0 = 0; u0 < U; u0 += RD_CH) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int f = 0; f < NQF; ++f) *reinterpret_cast<float4*>(Qb + ((size_t)f * 64 + 16 * i + lrow) * RD_LD + 4 * lpc) = rq[f][i];
#pragma unroll
                for (int f = 0; f < NEF; ++f) *reinterpret_cast<float4*>(Eb + ((size_t)f * 64 + 16 * i + lrow) * RD_LD + 4 * lpc) = re[f][i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (u0 + RD_CH < U) fetch(u0 + RD_CH);
#pragma unroll
            for (int c = 0; c < RD_CH / 4; ++c) {
                float qv[NQF][4], ev[NEF][4];
#pragma unroll
                for (int f = 0; f < NQF; ++f) {
                    const float4 t = *reinterpret_cast<const float4*>(Qb + ((size_t)f * 64 + lane) * RD_LD + 4 * c);
                    qv[f][0] = t.x; qv[f][1] = t.y; qv[f][2] = t.z; qv[f][3] = t.w;
                }
#pragma unroll
                for (int f = 0; f < NEF; ++f) {
                    const float4 t = *reinterpret_cast<const float4*>(Eb + ((size_t)f * 64 + lane) * RD_LD + 4 * c);
                    ev[f][0] = t.x; ev[f][1] = t.y; ev[f][2] = t.z; ev[f][3] = t.w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u0 + 4 * c + u < U) {
                        float qq[NQF], ee[NEF];
#pragma unroll
                        for (int f = 0; f < NQF; ++f) qq[f] = qv[f][u];
#pragma unroll
                        for (int f = 0; f < NEF; ++f) ee[f] = ev[f][u];
                        if constexpr (MODE == MODE_ROT_O || MODE == MODE_ROT_S) acc = rot_exact_op<MODE>(acc, qq, ee);
                        else acc = rank_op<MODE>(acc, qq, ee, a.g.sgn);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (have) {
            const int qs = quantise(a.sgn_scale * acc), qp = a.qpos[pr.x];
            if (qp < qs) atomicAdd(&a.b.counts[2 * (int64_t)pr.x + 0], 1);
            else if (qp == qs) atomicAdd(&a.b.counts[2 * (int64_t)pr.x + 1], 1);
        }
    }
}


}  // namespace kge

#endif
