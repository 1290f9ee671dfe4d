// evaluate(), contraction models (DistMult / ComplEx / HolE): EXACT ranks from an int8 matrix-core screening pass.
// Included by kge_rank.hip (namespace kge) behind the fp32 kernels, whose CountArgs / quantise / prep it shares.
//
// The declared score of a (query i, entity j) pair is the fp32 chain  S_ij = fmaf(q_U e_U, ... fmaf(q_1 e_1, 0))  in table
// order (rank_op<MODE_DOT>; v_mfma_f32_32x32x2_f32 reproduces it bit for bit at the fp32 VECTOR rate, 157 TFLOP/s).  A rank
// does not need S_ij, only on which side of two thresholds it lies: with qp = q(positive) and q(.) = int(sgn_scale * S * 1000)
// monotone in S (sgn_scale > 0 for these models),
//     "greater"  <=>  S_ij >= T_gt(i)          "equal"  <=>  T_ge(i) <= S_ij < T_gt(i)
// where T_ge / T_gt are the smallest fp32 values whose quantised score reaches qp / qp + 1 (found per query by bisection over
// the ordered fp32 bit patterns, with the kernels' own quantise()).  So:
//   1. rows are converted to 24-bit fixed point with a power-of-two scale PER ROW, as three signed 8-bit limbs
//      (rank_limbs_kernel; also the row's 1- and 2-norms);
//   2. the screening kernel forms  D~_ij = sum_k vq_ik ve_jk  with v_mfma_i32_32x32x32_i8 -- INTEGER arithmetic, exact, at ~28x
//      the fp32 matrix rate per instruction -- keeping the limb products down to weight 2^16 (6 of the 9: three int32
//      accumulators per output), and bounds |D~_ij A_i B_j - S_ij| RIGOROUSLY by
//          E_ij = u (1 + 2 U u) |W q_i|_2 |W e_j|_2            (the fp32 chain's own rounding: the partial sum behind unit j is
//                                                               rounded U - j more times, |acc_k| <= sum_{j<=k} |q_j e_j| (1+u)^k,
//                                                               so |S - D| <= u sum_j (U - j) |q_j e_j| (1 + u)^U, and Cauchy-Schwarz
//                                                               with W = diag(sqrt(U - j)): half of the classic gamma_U |q| |e|)
//               + (A_i |e_j|_1 + B_j |q_i|_1) / 2 + U A_i B_j / 4   (rounding of the rows to their fixed-point grids)
//               + U (2^23 + 2^14) A_i B_j                      (the three dropped limb products)
//               + 2^27 A_i B_j + 2^-22 |S~|                   (the epilogue's own fp32 roundings, U <= 2048; the relative part sits in the
//                                                               thresholds), the rest inflated by 2^-10;
//      a pair whose interval [S~ - E, S~ + E] lies on one side of both thresholds (or between them) is DECIDED and counted;
//   3. the others (a fraction of a per cent: those within ~1e-4 of the positive's quantisation cell) go to a list and are
//      recomputed by rank_recheck_kernel with the exact fp32 chain -- so the counts are bit-identical to the fp32 kernels
//      whatever the tables hold (rows with inf / NaN get an infinite bound: everything of theirs is rechecked).
// If the list overflows its capacity the whole call falls back to the exact fp32 MFMA kernel (device-side flag, no host
// round trip).  Nothing here approximates a result: the int8 pass only decides which comparisons need the exact chain.
#pragma once

namespace kge {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int SCR_Q = 128, SCR_K = 32;   // queries per workgroup (4 waves x 32), units per stage
constexpr int SCR_ROW_SLAB = 96;                       // bytes of one row per K slab: 3 limbs x 32 units
// Limb storage is FRAGMENT-MAJOR: [block of 32 rows][slab][limb][half][row % 32][16 bytes] -- the 64 lanes of a matrix operand
// fragment (lane = half * 32 + row % 32, 16 units each) read 1 KB of CONSECUTIVE memory, from global memory as from LDS.
constexpr int SCR_BLK_SLAB = 32 * SCR_ROW_SLAB;        // bytes of one 32-row block per K slab (3 072)

struct ScreenBufs {
    int* counter;        // [0] undecided pairs appended, [1] overflow flag, [2] candidate rows > 4 bits below their tile's scale (rank_limbs_tile_kernel)
    int32_t* counts;     // [n][2] this call's (greater, equal) counts (merged into the caller's unless the call fell back)
    int8_t* qlimbs;      // [ceil(n / 32)][S][3][2][32][16]
    float4* qm;          // [n] {A = 2^-a, u (1 + 2 U u) |W q|_2, |q|_1 / 2, 0}, all rounded up
    float2* qt;          // [n] {T_ge, T_gt}
    int8_t* elimbs;      // [ceil(m / 32)][S][3][2][32][16]
    float4* em;          // [m] {B, |W e|_2, |e|_1 / 2, 0}
    float4* tm;          // [ceil(m / 64)] per tile of 64 candidates {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t}: rank_limbs_tile_kernel (kernel r)
    int2* pairs;         // [cap] (query, candidate position)
    int64_t cap;
    int S;               // K slabs per row
};

static inline size_t scr_up(size_t x) { return (x + 255) & ~(size_t)255; }

// fixed part of the workspace (everything but the pair list), for n queries against m candidates of U units
static inline size_t screen_fixed_bytes(int64_t n, int64_t m, int U) {
    const size_t S = (size_t)(U + SCR_K - 1) / SCR_K;
    const size_t nb = (size_t)(n + 31) / 32 + 4, mb = (size_t)(m + 31) / 32 + 4;   // (+ a tile of slack: loaders read whole 128-row tiles)
    return 256 + scr_up((size_t)n * 8) + scr_up(nb * S * SCR_BLK_SLAB) + scr_up((size_t)n * 16) + scr_up((size_t)n * 8) +
           scr_up(mb * S * SCR_BLK_SLAB) + scr_up((size_t)m * 16) + scr_up(((size_t)(m + 63) / 64 + 4) * 16);
}

static inline ScreenBufs carve_screen(void* d_screen, size_t bytes, int64_t n, int64_t m, int U) {
    ScreenBufs b;
    b.S = (U + SCR_K - 1) / SCR_K;
    char* p = (char*)(((uintptr_t)d_screen + 255) & ~(uintptr_t)255);
    const char* end = (char*)d_screen + bytes;
    b.counter = (int*)p; p += 256;
    b.counts = (int32_t*)p; p += scr_up((size_t)n * 8);
    b.qlimbs = (int8_t*)p; p += scr_up(((size_t)(n + 31) / 32 + 4) * b.S * SCR_BLK_SLAB);
    b.qm = (float4*)p; p += scr_up((size_t)n * 16);
    b.qt = (float2*)p; p += scr_up((size_t)n * 8);
    b.elimbs = (int8_t*)p; p += scr_up(((size_t)(m + 31) / 32 + 4) * b.S * SCR_BLK_SLAB);
    b.em = (float4*)p; p += scr_up((size_t)m * 16);
    b.tm = (float4*)p; p += scr_up(((size_t)(m + 63) / 64 + 4) * 16);
    b.pairs = (int2*)p;
    b.cap = end > p ? (int64_t)((end - p) / 8) : 0;
    return b;
}

// ---- 1. rows -> fixed point limbs + norms: one wave per row --------------------------------------------------------------------
// src row r: table[ids ? ids[lo + r] : lo + r] (stride K floats), U units walked (whole float4s: U % 4 == 0).
// out: limbs in the fragment-major layout (units beyond U are zero), meta[r] = {scale, n2 * gamma (gamma = 1 for the entity side), n1 / 2, 0}.
// "Wild" candidate table: more than 1 / 64 of its rows lie > 4 bits below the scale of their tile of 64 (rank_limbs_tile_kernel counts
// them).  rank_screen_kernel_r's tile-wide scale would leave such rows without significant bits and send their outputs to the recheck
// list; the per-row scales of rank_limbs_kernel + rank_screen_kernel_v1 take the call instead.  Decided on the device: the kernels of
// both paths are launched, the ones of the path not taken return at once.
__device__ __forceinline__ bool screen_wild(const int* counter, int64_t m) { return (int64_t)counter[2] * 64 > m; }

__global__ __launch_bounds__(256) void rank_limbs_kernel(const float* __restrict__ table, int64_t stride, const int32_t* __restrict__ ids, int64_t lo,
                                                         int64_t nrows, int U, int S, float gamma, int8_t* __restrict__ limbs, float4* __restrict__ meta,
                                                         const int* __restrict__ only_if_wild = nullptr) {
    // (only_if_wild: the fall-back behind rank_limbs_tile_kernel -- a small persistent grid that walks the rows only for a wild table)
    if (only_if_wild && !screen_wild(only_if_wild, nrows)) return;
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (int64_t)gridDim.x * 4) {
    const int64_t id = ids ? (int64_t)ids[lo + r] : lo + r;
    const float4* row = reinterpret_cast<const float4*>(table + id * stride);
    const int nq = U >> 2;
    float mx = 0.f, n1 = 0.f, n2 = 0.f;
    bool bad = false;
    for (int q = lane; q < nq; q += 64) {
        const float4 t = row[q];
        const float a0 = fabsf(t.x), a1 = fabsf(t.y), a2 = fabsf(t.z), a3 = fabsf(t.w);
        bad |= !(a0 < INFINITY) || !(a1 < INFINITY) || !(a2 < INFINITY) || !(a3 < INFINITY);
        mx = fmaxf(fmaxf(mx, fmaxf(a0, a1)), fmaxf(a2, a3));
        n1 += (a0 + a1) + (a2 + a3);
        // position-weighted square norm: unit j of the chain passes through U - j roundings (see the bound in the header comment)
        const float w = (float)(U - 4 * q);
        n2 = fmaf(w * a0, a0, fmaf((w - 1.f) * a1, a1, fmaf((w - 2.f) * a2, a2, fmaf((w - 3.f) * a3, a3, n2))));
    }
    mx = wave_max(mx);
    n1 = wave_sum_shfl(n1);
    n2 = wave_sum_shfl(n2);
    bad = __ballot(bad) != 0ull;
    // scale: |x| * 2^a <= 2^22 for every unit of the row (limb l0 within +-65); a zero row takes a = 0
    int ex = 0;
    if (mx > 0.f) (void)frexpf(mx, &ex);          // mx = f * 2^ex, f in [0.5, 1)
    int a = 22 - ex;
    a = a > 120 ? 120 : (a < -100 ? -100 : a);    // (denormal-sized or huge rows: the scale stays a normal fp32; see the norms below)
    const float up = ldexpf(1.f, a), A = ldexpf(1.f, -a);
    // norms rounded up: fp32 accumulation of U non-negative terms is within (U + 8) u relative -- 2^-10 covers every U <= 8192
    const float infl = 1.f + 0x1p-10f;
    float m_n2 = sqrtf(n2) * infl * infl * gamma, m_n1 = 0.5f * n1 * infl;
    // rows the fixed-point grid cannot represent to within A / 2 (inf / NaN units, or units that would overflow 2^22 after the
    // clamp of `a`): an infinite bound sends every pair of the row to the exact recheck
    if (bad || mx * up > 4194304.f) { m_n2 = INFINITY; m_n1 = INFINITY; }
    if (lane == 0) meta[r] = make_float4(A, m_n2, m_n1, 0.f);
    int8_t* out = limbs + (r >> 5) * (int64_t)S * SCR_BLK_SLAB + (r & 31) * 16;
    for (int q = lane; q < S * 8; q += 64) {   // 4 units per step, 8 steps per slab
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq && !bad) t = row[q];
        int v[4] = {(int)rintf(t.x * up), (int)rintf(t.y * up), (int)rintf(t.z * up), (int)rintf(t.w * up)};
        uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // balanced base-256 digits: v = l0 2^16 + l1 2^8 + l2, l1, l2 in [-128, 127], |l0| <= 65
            const int l2 = ((v[c] + 128) & 255) - 128;
            const int v1 = (v[c] - l2) >> 8;
            const int l1 = ((v1 + 128) & 255) - 128;
            const int l0 = (v1 - l1) >> 8;
            p0 |= (uint32_t)(l0 & 255) << (8 * c); p1 |= (uint32_t)(l1 & 255) << (8 * c); p2 |= (uint32_t)(l2 & 255) << (8 * c);
        }
        // units 4 w .. 4 w + 3 of slab `slab`: half = w >> 2, dword w & 3 of this row's 16-byte piece; pieces of (limb, half) are
        // 512 bytes apart, limbs 1 024
        const int slab = q >> 3, w = q & 7;
        uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)slab * SCR_BLK_SLAB + (w >> 2) * 512) + (w & 3);
        o[0] = p0; o[256] = p1; o[512] = p2;
    }
    }
}

// ---- 1b. the candidate rows with ONE scale per tile of 64 candidate positions (rank_screen_kernel_r: a tile's outputs then share
// their scale with their query row only, and the decision thresholds of a (query row, tile) can be put into the accumulators' own
// integer units once instead of scaling every output).  Two passes, one wave per row each: rank_rowstats_kernel (largest magnitude,
// norms, "holds inf / NaN"), then rank_limbs_tile_kernel: every wave reads the 64 row records of its tile (lane = row), derives the
// tile's scale from its largest finite row and puts its own row on that grid.  Row metas as rank_limbs_kernel's (meta.x = the TILE's
// scale), limbs POSITION-major: [tile][slab][block of the tile][limb][half][row % 32][16 bytes] (6 144 bytes per (tile, slab): what
// rank_screen_kernel_r's loader moves per stage); tmeta[tile] = {B_t, max |W e|_2, max |e|_1 / 2, 1 / B_t} over the tile's rows (an
// inf / NaN row: infinite maxima -- the tile's outputs all go to the exact recheck); counter[2] += rows > 4 bits below their tile's
// scale (screen_wild).
__global__ __launch_bounds__(256) void rank_rowstats_kernel(const float* __restrict__ table, int64_t stride, const int32_t* __restrict__ ids, int64_t lo,
                                                            int64_t nrows, int U, float4* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t id = ids ? (int64_t)ids[lo + r] : lo + r;
    const float4* row = reinterpret_cast<const float4*>(table + id * stride);
    const int nq = U >> 2;
    float mx = 0.f, n1 = 0.f, n2 = 0.f;
    bool bad = false;
    for (int q = lane; q < nq; q += 64) {
        const float4 t = row[q];
        const float a0 = fabsf(t.x), a1 = fabsf(t.y), a2 = fabsf(t.z), a3 = fabsf(t.w);
        bad |= !(a0 < INFINITY) || !(a1 < INFINITY) || !(a2 < INFINITY) || !(a3 < INFINITY);
        mx = fmaxf(fmaxf(mx, fmaxf(a0, a1)), fmaxf(a2, a3));
        n1 += (a0 + a1) + (a2 + a3);
        const float w = (float)(U - 4 * q);   // (position-weighted square norm: see rank_limbs_kernel)
        n2 = fmaf(w * a0, a0, fmaf((w - 1.f) * a1, a1, fmaf((w - 2.f) * a2, a2, fmaf((w - 3.f) * a3, a3, n2))));
    }
    mx = wave_max(mx);
    n1 = wave_sum_shfl(n1);
    n2 = wave_sum_shfl(n2);
    bad = __ballot(bad) != 0ull;
    if (lane == 0) stats[r] = make_float4(bad ? 0.f : mx, n1, n2, bad ? 1.f : 0.f);
}

__global__ __launch_bounds__(256) void rank_limbs_tile_kernel(const float* __restrict__ table, int64_t stride, const int32_t* __restrict__ ids, int64_t lo,
                                                              int64_t nrows, int U, int S, const float4* __restrict__ stats, int8_t* __restrict__ limbs,
                                                              float4* __restrict__ meta, float4* __restrict__ tmeta, int* __restrict__ counter) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (rows up to the end of the last tile: zero limbs beyond the range)
    const int64_t tile = r >> 6;
    if (tile * 64 >= nrows) return;
    // the tile's 64 row records, one per lane
    const int64_t lr_row = tile * 64 + lane;
    const bool have = lr_row < nrows;
    const float4 st = have ? stats[lr_row] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool lbad = st.w != 0.f;
    // the tile's scale: |x| * 2^a <= 2^22 for every unit of every finite row (limb l0 within +-65); all-zero tile: a = 0
    const float tmx = wave_max(st.x);
    int ex = 0;
    if (tmx > 0.f) (void)frexpf(tmx, &ex);
    int a = 22 - ex;
    a = a > 120 ? 120 : (a < -100 ? -100 : a);
    const float up = ldexpf(1.f, a), A = ldexpf(1.f, -a);
    const bool tile_bad = tmx * up > 4194304.f;   // (the clamp of `a` bit: nothing of this tile is representable)
    const float infl = 1.f + 0x1p-10f;
    float m_n2 = sqrtf(st.z) * infl * infl, m_n1 = 0.5f * st.y * infl;
    if (lbad || tile_bad) { m_n2 = INFINITY; m_n1 = INFINITY; }
    const int mine = (int)(r & 63);   // this wave's row inside the tile
    if (mine == 0) {   // the tile's first wave writes the row metas and the tile record
        if (have) meta[lr_row] = make_float4(A, m_n2, m_n1, 0.f);
        const float t2 = wave_max(have ? m_n2 : 0.f), t1 = wave_max(have ? m_n1 : 0.f);
        int exr = 0;
        if (st.x > 0.f) (void)frexpf(st.x, &exr);
        const bool far = have && !lbad && !tile_bad && st.x > 0.f && (ex - exr) > 4;   // own scale > 4 bits finer than the tile's (see screen_wild)
        const int nfar = __popcll(__ballot(far));
        if (lane == 0) {
            tmeta[tile] = make_float4(A, t2, t1, up);
            if (nfar) atomicAdd(counter + 2, nfar);
        }
    }
    const bool rbad = __shfl(lbad ? 1 : 0, mine, 64) != 0;
    const bool live = r < nrows && !rbad && !tile_bad;
    const int64_t id = r < nrows ? (ids ? (int64_t)ids[lo + r] : lo + r) : 0;
    const float4* row = reinterpret_cast<const float4*>(table + id * stride);
    const int nq = U >> 2;
    int8_t* out = limbs + tile * (int64_t)S * (2 * SCR_BLK_SLAB) + (mine >> 5) * SCR_BLK_SLAB + (mine & 31) * 16;
    for (int q = lane; q < S * 8; q += 64) {   // 4 units per step, 8 steps per slab (rows beyond the range / non-finite rows: zeros)
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq && live) t = row[q];
        int v[4] = {(int)rintf(t.x * up), (int)rintf(t.y * up), (int)rintf(t.z * up), (int)rintf(t.w * up)};
        uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // balanced base-256 digits: v = l0 2^16 + l1 2^8 + l2, l1, l2 in [-128, 127], |l0| <= 65
            const int l2 = ((v[c] + 128) & 255) - 128;
            const int v1 = (v[c] - l2) >> 8;
            const int l1 = ((v1 + 128) & 255) - 128;
            const int l0 = (v1 - l1) >> 8;
            p0 |= (uint32_t)(l0 & 255) << (8 * c); p1 |= (uint32_t)(l1 & 255) << (8 * c); p2 |= (uint32_t)(l2 & 255) << (8 * c);
        }
        const int slab = q >> 3, w = q & 7;
        uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)slab * (2 * SCR_BLK_SLAB) + (w >> 2) * 512) + (w & 3);
        o[0] = p0; o[256] = p1; o[512] = p2;
    }
}

// ---- thresholds: smallest fp32 S with quantise(sgn_scale * S) >= target, by bisection over the ordered bit patterns ----------
__device__ __forceinline__ float f_of_ord(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o); }
__device__ __forceinline__ float least_with_quantised_at_least(float sgn_scale, long long target) {
    if (target > 2147483647ll) return INFINITY;
    // ordered patterns of the finite floats: ord(-FLT_MAX) .. ord(+FLT_MAX)
    uint32_t lo = ~0xFF7FFFFFu, hi = 0x7F7FFFFFu ^ 0x80000000u;   // ord(x) = x < 0 ? ~bits : bits ^ 0x80000000
    if ((long long)quantise(sgn_scale * f_of_ord(hi)) < target) return INFINITY;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((long long)quantise(sgn_scale * f_of_ord(mid)) >= target) hi = mid; else lo = mid + 1;
    }
    return f_of_ord(lo);
}
__global__ void rank_thresholds_kernel(const int* __restrict__ qpos, int64_t n, float sgn_scale, float2* __restrict__ qt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long qp = qpos[i];
    qt[i] = make_float2(least_with_quantised_at_least(sgn_scale, qp), least_with_quantised_at_least(sgn_scale, qp + 1));
}

// ---- 2. the screening kernel ------------------------------------------------------------------------------------------------------
struct ScreenArgs {
    ScreenBufs b;
    int64_t n, m;          // queries, candidates (positions [0, m) of the call's candidate range)
    int ent_per_block;     // candidates per workgroup (a multiple of SCR_ET)
    int qtiles, splits;
    int U;
    float drop;            // U * (2^23 + 2^14 + 1/4)
    int wild_mode;         // 0: run; 2: run only if the tile-scale pass found the candidate table NOT wild (screen_wild; rank_screen_kernel_r)
    int nblk;              // rank_screen_kernel_v1_wild: virtual blocks of the ordinary launch
};

constexpr int SCR_THREADS = 256;   // 4 waves, each a 32-query block against the workgroup's 64-entity tile; TWO workgroups per CU
constexpr int SCR_ET = 64;         // entities per tile
constexpr int SCR_PEND = 512;      // undecided pairs a wave parks in LDS before they go to the list
constexpr size_t SCR_LDS_BYTES = (size_t)2 * 3 * 2 * SCR_ET * 16 + 128 * 16 + 128 * 16 + SCR_ET * 16 + 4 * SCR_PEND * 8;   // E double-buffered + row metas + parked pairs

// Operand feed.  The i8 matrix instruction retires 65 536 multiply-adds in ~33 cycles, so the kernel is bound by how fast the
// operands arrive and by its epilogue, not by the matrix pipe.  A wave's QUERY fragments come straight from global memory
// (L2 / L1) into registers, one stage ahead: in the fragment-major limb layout that is one coalesced 1 KB read per limb.  The
// ENTITY slab, shared by the workgroup's four query blocks, is staged through LDS in the same order (a fragment read is 512
// contiguous bytes per half-wave).  Two 256-thread workgroups share a CU (2 waves per SIMD, 256 registers each) and run out of
// phase: one's epilogue (VALU) under the other's matrix work.
// (the kernel's body, for one workgroup's share `vblock` of the launch: rank_screen_kernel_v1 runs it once per workgroup,
// rank_screen_kernel_v1_wild -- the fall-back behind rank_screen_kernel_r -- in a loop over a small persistent grid)
__device__ __forceinline__ void rank_screen_v1_body(const ScreenArgs& a, const int vblock) {
    extern __shared__ __attribute__((aligned(16))) char smem_scr[];
    typedef uint4 (*slab_t)[2][3][2][32];   // [buffer][entity block][limb][half][row]: a buffer is the two blocks' slabs back to back
    slab_t Es = reinterpret_cast<slab_t>(smem_scr);
    float4* qm_s = reinterpret_cast<float4*>(smem_scr + (size_t)2 * 3 * 2 * SCR_ET * 16);
    float4* qt_s = qm_s + 128;
    float4* em_s = qt_s + 128;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wq = wv * 32;
    int bx, by;   // XCD-aware work order, as rank_count_mfma_kernel
    {
        const int xcd = vblock & 7;
        const int64_t i = vblock >> 3;
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
    const char* const qbase = reinterpret_cast<const char*>(a.b.qlimbs) + ((q0 + 32 * wv_s) >> 5) * (int64_t)S * SCR_BLK_SLAB;
    const uint32_t qoff = (uint32_t)lane * 16u;
    auto load_q = [&](int s, v4i32 (&f)[3]) {
        const char* const qs = qbase + (size_t)s * SCR_BLK_SLAB;
#pragma unroll
        for (int lb = 0; lb < 3; ++lb) { const uint4 u = *reinterpret_cast<const uint4*>(qs + (qoff + 1024u * lb)); f[lb] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
    };
    // entity slab loader: a stage's LDS image is the two 32-row blocks' 3 072-byte slabs back to back (6 144 bytes), exactly as
    // they lie in memory.  Every thread copies 16 bytes at 16 tid (the first 4 096) and 8 bytes at 4 096 + 8 tid: both loads of
    // ALL threads are unconditional (a predicated load leaves the compiler without a vmcnt it can count on -- it then waited for
    // loads it had just issued) and every instruction reads and writes one contiguous run.
    const uint32_t blk_stride = (uint32_t)S * SCR_BLK_SLAB;   // bytes between consecutive 32-row blocks (S <= 64 slabs: < 2^18)
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

    v4i32 qf[2][3];
    set_src(e_begin);
    load_e(eA0, eB0);        // position 0 -> LDS buffer 0
    load_e(eA1, eB1);        // position 1 -> register set 1
    load_q(0, qf[0]);
    store_e(0, eA0, eB0);
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
        // candidate metas (used at stage 0 only), the entity pieces of position g + 2, the query fragments of g + 1.
        // (the metas of up to 63 candidates beyond the range are read: rows of later candidates or the head of the recheck list)
        float4 m4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.b.em + et) + emoff);
        if constexpr (P == 0) load_e(eA0, eB0); else load_e(eA1, eB1);
        load_q(st + 1 == S ? 0 : st + 1, qf[P ^ 1]);   // (the query rows do not change with the entity tile)
        __builtin_amdgcn_sched_barrier(0);
        // both entity blocks' fragments first, then the 12 matrix instructions ordered so that two of them on the SAME accumulator
        // are never adjacent (a dependent pair would wait out the first one's latency: twice its issue time)
        v4i32 eb[2][3];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int lb = 0; lb < 3; ++lb) { const uint4 u = Es[P][ni][lb][lh][l31]; eb[ni][lb] = v4i32{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[0][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[1][0], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[0][1], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[1][1], acc[1][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[0][2], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][0], eb[1][2], acc[2][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][1], eb[0][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][1], eb[1][0], acc[1][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][1], eb[0][1], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][1], eb[1][1], acc[2][1], 0, 0, 0);
        acc[2][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][2], eb[0][0], acc[2][0], 0, 0, 0);
        acc[2][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[P][2], eb[1][0], acc[2][1], 0, 0, 0);
        if constexpr (P == 0) store_e(1, eA1, eB1); else store_e(0, eA0, eB0);
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

__global__ __launch_bounds__(SCR_THREADS, 2) void rank_screen_kernel_v1(ScreenArgs a) {
    rank_screen_v1_body(a, (int)blockIdx.x);
}

// The per-row-scale path behind rank_screen_kernel_r (a.wild_mode == 1): it has work only when the tile-scale pass found the candidate
// table wild (screen_wild) -- decided on the device, so the launch always happens.  A launch of thousands of workgroups that each read
// a flag and leave still costs ~100 us on a busy device (every workgroup waits for a slot): this one is a few hundred persistent
// workgroups that walk the virtual blocks a.nblk of the ordinary launch.
__global__ __launch_bounds__(SCR_THREADS, 2) void rank_screen_kernel_v1_wild(ScreenArgs a) {
    if (!screen_wild(a.b.counter, a.m)) return;
    for (int vb = (int)blockIdx.x; vb < a.nblk; vb += (int)gridDim.x) {
        rank_screen_v1_body(a, vb);
        __syncthreads();   // (the next share rewrites the workgroup's LDS tables)
    }
}

// ---- 3. exact recheck of the undecided pairs: one lane per pair, the fp32 chain of rank_op<MODE_DOT> --------------------------
struct RecheckArgs {
    const float* ent;
    const float* Q;
    const int* qpos;
    const int32_t* ent_ids;
    int64_t ent_lo;
    int U, K, QW;
    float sgn_scale;
    ScreenBufs b;
};

// One lane per pair, 64 pairs per wave -- but the rows are NOT read lane by lane (64 lanes x 16 bytes of 64 different rows per
// load instruction: the address path, not the bytes, bound the first version at ~2 TB/s for 650 000 pairs).  A wave fetches
// 32-unit chunks of its 64 query rows and 64 entity rows COALESCED (8 rows x one 128-byte line per instruction), parks them in
// its private LDS region (row stride 144 bytes: a lane walking its own row is conflict-free) and every lane then runs its
// 32 fused multiply-adds in unit order from LDS.  Same chain, same bits as rank_op<MODE_DOT>.
constexpr int RCK_CH = 32;                 // units per chunk
constexpr int RCK_LD = RCK_CH + 4;         // LDS row stride in floats
constexpr size_t RCK_LDS_BYTES = (size_t)4 * 2 * 64 * RCK_LD * sizeof(float);   // 4 waves x (Q, E) x 64 rows

// FILTER = true: the same chain for the FILTER pass of get_ranks (amdkge_rank_filter, contraction models): the pairs are (query,
// known positive's table row) from filter_pairs_kernel ((q, -1): an id outside the candidate set), the outcome "the filtered
// entity scores at least the positive" is subtracted -- b.counts is the caller's `sub` array there, one int per query.
template <bool FILTER>
__global__ __launch_bounds__(256) void rank_recheck_kernel(RecheckArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_rck[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Qb = reinterpret_cast<float*>(smem_rck) + (size_t)wv * 2 * 64 * RCK_LD;
    float* Eb = Qb + 64 * RCK_LD;
    if (FILTER && a.b.counter[1]) return;   // the pair list overflowed: rank_filter_kernel does the whole pass
    const int64_t npairs = min((int64_t)a.b.counter[0], a.b.cap);
    const int64_t ngroups = (npairs + 63) / 64;
    const int lrow = lane >> 3, lpc = lane & 7;   // loader: 8 rows per instruction, 8 16-byte pieces per row chunk
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        const int64_t p = grp * 64 + lane;
        bool have = p < npairs;
        int2 pr = a.b.pairs[have ? p : npairs - 1];
        if (FILTER && pr.y < 0) { have = false; pr.y = 0; }
        const int64_t pos = FILTER ? (int64_t)pr.y : a.ent_lo + pr.y;
        const int64_t qoff = (int64_t)pr.x * a.QW, eoff = (!FILTER && a.ent_ids ? (int64_t)a.ent_ids[pos] : pos) * a.K;
        float acc = 0.f;
        // rows 8 i + lrow of this wave's 64 pairs: their offsets come from the lanes that own them.  The chunk after the one being
        // multiplied is already in flight (registers): a wave owns one or two groups, so its 13 load round trips in a row, not
        // the bytes, were the time of this kernel (98 us for 150 000 pairs).
        int64_t qo[8], eo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { qo[i] = __shfl(qoff, 8 * i + lrow, 64) + 4 * lpc; eo[i] = __shfl(eoff, 8 * i + lrow, 64) + 4 * lpc; }
        float4 rq[8], re[8];
        auto fetch = [&](int u0) {
            const bool in = u0 + 4 * lpc < a.U;   // (U % 4 == 0: a piece is inside or outside as a whole)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rq[i] = in ? *reinterpret_cast<const float4*>(a.Q + qo[i] + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
                re[i] = in ? *reinterpret_cast<const float4*>(a.ent + eo[i] + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        fetch(0);
        for (int u0 = 0; u0 < a.U; u0 += RCK_CH) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                *reinterpret_cast<float4*>(Qb + (8 * i + lrow) * RCK_LD + 4 * lpc) = rq[i];
                *reinterpret_cast<float4*>(Eb + (8 * i + lrow) * RCK_LD + 4 * lpc) = re[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (u0 + RCK_CH < a.U) fetch(u0 + RCK_CH);
            const int lim = min(RCK_CH, a.U - u0) >> 2;
#pragma unroll
            for (int c = 0; c < RCK_CH / 4; ++c) {
                if (c < lim) {
                    const float4 qv = *reinterpret_cast<const float4*>(Qb + lane * RCK_LD + 4 * c);
                    const float4 ev = *reinterpret_cast<const float4*>(Eb + lane * RCK_LD + 4 * c);
                    acc = fmaf(qv.x, ev.x, acc); acc = fmaf(qv.y, ev.y, acc);
                    acc = fmaf(qv.z, ev.z, acc); acc = fmaf(qv.w, ev.w, acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (have) {
            const int qs = quantise(a.sgn_scale * acc), qp = a.qpos[pr.x];
            if constexpr (FILTER) {
                if (qp <= qs) atomicAdd(&a.b.counts[pr.x], 1);   // always <=, whatever the tie strategy (AbstractScoringLayer.py:292-307)
            } else {
                if (qp < qs) atomicAdd(&a.b.counts[2 * (int64_t)pr.x + 0], 1);
                else if (qp == qs) atomicAdd(&a.b.counts[2 * (int64_t)pr.x + 1], 1);
            }
        }
    }
}

// the call's counts join the caller's (+=) unless the pair list overflowed (then the exact kernel, guarded by the same flag,
// produces them)
__global__ void rank_screen_merge_kernel(ScreenBufs b, int64_t n, int32_t* __restrict__ counts) {
    if (b.counter[1] != 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n && b.counts[i]) counts[i] += b.counts[i];
}

}  // namespace kge
