// Device-side building blocks shared by the gfx950 kernels of libamdkge.
// wave = 64 lanes everywhere; compiled with -ffp-contract=off so that every rounding point is the
// one written here (explicit fmaf where a fused multiply-add is intended).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amdkge.h"

#define KGE_WAVE 64

namespace kge {

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Same contract as oracle/philox.py.
// ----------------------------------------------------------------------------------------------
struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

struct SampleCfg {
    int64_t base;      // first row id of the sampling range
    uint32_t range;    // number of rows negatives are drawn from
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    int64_t row_offset;  // global index of positive 0 of this launch
    int64_t b_global;    // positives in the whole (all-rank) batch
};

// Draw for corruption j of (local) positive i.  keep_subj = bit 0 of x0, replacement = mulhi(x1, range).
__device__ __forceinline__ void draw_corruption(const SampleCfg& sc, int64_t i, int j, int& keep_subj, int& repl) {
    const uint64_t row = (uint64_t)j * (uint64_t)sc.b_global + (uint64_t)(sc.row_offset + i);
    const u32x4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), sc.step_lo, sc.step_hi, sc.seed_lo, sc.seed_hi);
    keep_subj = (int)(r.x & 1u);
    repl = (int)(sc.base + (int64_t)__umulhi(r.y, sc.range));
}

// ----------------------------------------------------------------------------------------------
// wave64 reductions.  DPP row ops + row_bcast (GCN3/CDNA wave64 idiom): 6 VALU-rate steps, total in
// lane 63, broadcast through an SGPR.
// ----------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov0<0x111>(v);        // row_shr:1
    v += dpp_mov0<0x112>(v);        // row_shr:2
    v += dpp_mov0<0x114>(v);        // row_shr:4
    v += dpp_mov0<0x118>(v);        // row_shr:8   -> lane 15 of each row holds the row total
    v += dpp_mov0<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v += dpp_mov0<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The wave totals of N <= 8 per-lane values AT ONCE (a transposing reduction): at each of the first three levels a lane keeps
// half of its values and hands the other half to its partner, so N values cost (N + N/2 + N/4) exchange steps instead of 6 N,
// and the steps of one level are independent of each other (no DPP wait states to pad).  The additions are those of wave_sum's
// tree -- lanes paired at distance 1, 2, 4, 8, then the rows, then the halves (fp32 addition commutes) -- so every total is
// bit-identical to wave_sum of that value.  Afterwards EVERY lane holds one total: lane l that of value wave_multi_slot(l);
// value f is found in lane wave_multi_lane(f) of every group of eight lanes.
// Partners: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (l ^ 7: the half-row's quads mirror each other, hence the
// selectors of the first two levels are the lane's bits 0 / 1 XOR bit 2), row_ror:8, then gfx950's v_permlane16_swap /
// v_permlane32_swap for the rows and the halves.
struct WaveMultiSel {
    bool x0, x1, x2;
};
__device__ __forceinline__ WaveMultiSel wave_multi_sel(int lane) {
    return {(bool)((lane ^ (lane >> 2)) & 1), (bool)(((lane >> 1) ^ (lane >> 2)) & 1), (bool)((lane >> 2) & 1)};
}
__device__ __forceinline__ int wave_multi_slot(int lane) {
    const WaveMultiSel s = wave_multi_sel(lane);
    return (int)s.x0 | ((int)s.x1 << 1) | ((int)s.x2 << 2);
}
__host__ __device__ constexpr int wave_multi_lane(int f) { return f < 4 ? f : 11 - f; }

template <int N, int CTRL>
__device__ __forceinline__ void wave_multi_level(const float (&in)[N], float (&out)[(N + 1) / 2], bool sel) {
#pragma unroll
    for (int j = 0; j < (N + 1) / 2; ++j) {
        if (2 * j + 1 < N) {
            const float lo = in[2 * j], hi = in[2 * j + 1];   // (values first: a conditional on the array elements selects ADDRESSES)
            const float keep = sel ? hi : lo;
            const float send = sel ? lo : hi;
            out[j] = keep + dpp_mov0<CTRL>(send);
        } else {
            out[j] = in[2 * j] + dpp_mov0<CTRL>(in[2 * j]);   // no second value: a plain exchange
        }
    }
}

template <int N>
__device__ __forceinline__ float wave_sum_multi(const float (&a)[N], const WaveMultiSel& s) {
    static_assert(N >= 1 && N <= 8, "wave_sum_multi: up to eight values");
    constexpr int M = (N + 1) / 2, K = (M + 1) / 2;
    float b[M], c[K], d[1];
    wave_multi_level<N, 0xB1>(a, b, s.x0);    // quad_perm:[1,0,3,2]
    wave_multi_level<M, 0x4E>(b, c, s.x1);    // quad_perm:[2,3,0,1]
    wave_multi_level<K, 0x141>(c, d, s.x2);   // row_half_mirror
    float x = d[0];
    x += dpp_mov0<0x128>(x);                  // row_ror:8
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ----------------------------------------------------------------------------------------------
// small fixed vectors with 16/8/4-byte global accesses
// ----------------------------------------------------------------------------------------------
template <int VEC>
struct fvec {
    float v[VEC];
};

template <int VEC>
__device__ __forceinline__ fvec<VEC> ldg(const float* p) {
    fvec<VEC> r;
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
        r.v[0] = *p;
    }
    return r;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    // gfx950 global_atomic_add_f32 (no CAS loop, no return); HBM allocated by hipMalloc is coarse grained
    unsafeAtomicAdd(p, v);
}

__device__ __forceinline__ void atomic_add_f32_wg(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ----------------------------------------------------------------------------------------------
// Scoring-model arithmetic on one "unit": one float of each row (TransE, DistMult) or one complex
// component pair (ComplEx, HolE, RotatE: re at column c, im at column k+c).
// Rounding points follow the reference forward code (file:line in each block).
// ----------------------------------------------------------------------------------------------
template <int MODEL>
struct ModelTraits {
    static constexpr bool kComplex = (MODEL == AMDKGE_COMPLEX || MODEL == AMDKGE_HOLE || MODEL == AMDKGE_ROTATE);
    static constexpr int NC = kComplex ? 2 : 1;
};

struct ModelConst {
    float score_scale;   // HolE: fp32(2/k) (HolE.py:45); others 1
    float score_sign;    // TransE/RotatE: -1 (tf.negative), others +1
    float phase_div;     // RotatE: fp32(embedding_range/pi) (RotatE.py:95-98)
};

// relation unit as loaded from the table -> what the scoring needs (RotatE: cos/sin of the phase)
template <int MODEL>
__device__ __forceinline__ void prep_rel(const ModelConst& mc, float (&p)[ModelTraits<MODEL>::NC]) {
    if constexpr (MODEL == AMDKGE_ROTATE) {
        const float phi = p[0] / mc.phase_div;  // RotatE.py:96: theta / (embedding_range / pi)
        p[0] = cosf(phi);
        p[1] = sinf(phi);
    }
}

// RotatE's DECLARED relation form (evaluate / predict / the owner-computes train step's per-step phase table): the phase
// theta / (embedding_range / pi) rounded to fp32 (RotatE.py:96), then cos and sin CORRECTLY ROUNDED to fp32 -- evaluated in
// fp64 (ocml, <= 2 ulp of fp64) and rounded once, which differs from the correctly rounded fp32 value only when the fp64
// result lies within ~2^-52 relative of an fp32 rounding boundary (probability ~2^-27 per value).  A CPU restatement does the
// same with libm's fp64 cos / sin (oracle/rank_ordered.py), which is what makes RotatE's ranks bit-comparable; cosf / sinf
// (prep_rel above, 1-2 ulp, implementation-defined bits) stay in the atomic-path train kernel that evaluates them per positive.
template <int MODEL>
__device__ __forceinline__ void prep_rel_exact(const ModelConst& mc, float (&p)[ModelTraits<MODEL>::NC]) {
    if constexpr (MODEL == AMDKGE_ROTATE) {
        const double phi = (double)(p[0] / mc.phase_div);
        double sn, cs;
        sincos(phi, &sn, &cs);
        p[0] = (float)cs;
        p[1] = (float)sn;
    }
}

// Correctly rounded fp32 square root in 6 issue slots + one quarter-rate instruction.  g = v_rsq_f32(x) (1 ulp), y = x g,
// r = x - y^2 EXACTLY (one fma: y is within a few ulp of sqrt(x), so the residual is representable), y' = y + r (g/2).
// Verified EXHAUSTIVELY on gfx950 against sqrtf for every fp32 x >= 2^-102 (scripts/experiments/sqrt_exact.hip,
// profiles/r03_sqrt_exact.log: 0 mismatches in [2^-100, FLT_MAX]); below that the residual underflows, and v_rsq_f32 flushes
// denormal inputs, so x < 2^-100 (g > 2^50), 0 (g = inf), inf (g = 0) and NaN take the libm path.
__device__ __forceinline__ float sqrt_rn(float x) {
    const float g = __builtin_amdgcn_rsqf(x);
    if (!(g > 0.f && g <= 0x1p50f)) return sqrtf(x);
    const float y = x * g, h = 0.5f * g;
    const float r = fmaf(-y, y, x);
    return fmaf(r, h, y);
}

#ifdef KGE_FAST_ROTATE
#define KGE_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define KGE_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#else
#define KGE_SQRT(x) sqrtf(x)
#define KGE_DIV(a, b) ((a) / (b))
#endif

// EXACT (the deterministic train mode): IEEE square root and division -- correctly rounded, the same bits as numpy's -- instead
// of the hardware approximations (v_sqrt_f32 / v_rcp_f32, 1 ulp, implementation-defined bits) the train kernels otherwise use
template <bool EXACT>
__device__ __forceinline__ float kge_sqrt_t(float x) { if constexpr (EXACT) return sqrtf(x); else return KGE_SQRT(x); }
template <bool EXACT>
__device__ __forceinline__ float kge_div_t(float a, float b) { if constexpr (EXACT) return a / b; else return KGE_DIV(a, b); }

// un-negated, un-scaled contribution of one unit to the score
template <int MODEL, bool EXACT = false>
__device__ __forceinline__ float score_unit(const float (&s)[ModelTraits<MODEL>::NC], const float (&p)[ModelTraits<MODEL>::NC],
                                            const float (&o)[ModelTraits<MODEL>::NC]) {
    if constexpr (MODEL == AMDKGE_TRANSE) {
        return fabsf(s[0] + p[0] - o[0]);  // TransE.py:51-53 (L1 norm of s+p-o)
    } else if constexpr (MODEL == AMDKGE_DISTMULT) {
        return s[0] * p[0] * o[0];  // DistMult.py:48
    } else if constexpr (MODEL == AMDKGE_COMPLEX || MODEL == AMDKGE_HOLE) {
        // ComplEx.py:58-62
        return s[0] * (p[0] * o[0] + p[1] * o[1]) + s[1] * (p[0] * o[1] - p[1] * o[0]);
    } else {
        // RotatE.py:100-104 ; p = (cos, sin)
        const float re = s[0] * p[0] - s[1] * p[1] - o[0];
        const float im = s[0] * p[1] + s[1] * p[0] - o[1];
        return kge_sqrt_t<EXACT>(re * re + im * im);
    }
}

// g * d(score)/d(s,p,o) for one unit.  `g` already contains score_sign*score_scale.
// RotatE: dp[0] is d/d(phi) (the caller divides by phase_div once at the end), dp[1] = 0.
// pad1: 0 for a live unit, 1 for a PADDING unit of the stored row layout (include/amdkge.h).  Only RotatE uses it: a
// padding unit has s = o = 0, hence modulus 0 and g / 0 * 0 = NaN; adding pad1 to the modulus makes its gradient the
// exact zero it must be, while a live unit's modulus is unchanged bit for bit (m + 0.0f == m, NaN for a true m == 0 as
// in the reference).  Every other model's padding units give exact zeros on their own.
template <int MODEL, bool EXACT = false>
__device__ __forceinline__ void grad_unit(const float (&s)[ModelTraits<MODEL>::NC], const float (&p)[ModelTraits<MODEL>::NC],
                                          const float (&o)[ModelTraits<MODEL>::NC], float g,
                                          float (&ds)[ModelTraits<MODEL>::NC], float (&dp)[ModelTraits<MODEL>::NC],
                                          float (&dd)[ModelTraits<MODEL>::NC], float pad1 = 0.f) {
    if constexpr (MODEL == AMDKGE_TRANSE) {
        const float d = s[0] + p[0] - o[0];
        const float sg = (d > 0.f) ? g : ((d < 0.f) ? -g : ((d == d) ? 0.f : d));  // g*sign(d), sign(0)=0, sign(NaN)=NaN (tf.sign) ; g carries the minus
        ds[0] = sg; dp[0] = sg; dd[0] = -sg;
    } else if constexpr (MODEL == AMDKGE_DISTMULT) {
        ds[0] = g * (p[0] * o[0]); dp[0] = g * (s[0] * o[0]); dd[0] = g * (s[0] * p[0]);
    } else if constexpr (MODEL == AMDKGE_COMPLEX || MODEL == AMDKGE_HOLE) {
        ds[0] = g * (p[0] * o[0] + p[1] * o[1]); ds[1] = g * (p[0] * o[1] - p[1] * o[0]);
        dp[0] = g * (s[0] * o[0] + s[1] * o[1]); dp[1] = g * (s[0] * o[1] - s[1] * o[0]);
        dd[0] = g * (s[0] * p[0] - s[1] * p[1]); dd[1] = g * (s[0] * p[1] + s[1] * p[0]);
    } else {
        const float c = p[0], sn = p[1];
        const float re = s[0] * c - s[1] * sn - o[0];
        const float im = s[0] * sn + s[1] * c - o[1];
        const float m = kge_sqrt_t<EXACT>(re * re + im * im) + pad1;
        const float gm = kge_div_t<EXACT>(g, m);  // no epsilon: m == 0 -> NaN exactly like the reference (RotatE.py:102-104)
        ds[0] = gm * (re * c + im * sn);
        ds[1] = gm * (-re * sn + im * c);
        dp[0] = gm * (re * (-s[0] * sn - s[1] * c) + im * (s[0] * c - s[1] * sn));
        dp[1] = 0.f;
        dd[0] = -gm * re;
        dd[1] = -gm * im;
    }
}

__host__ __device__ inline int internal_k_of(int model, int k) {
    return (model == AMDKGE_COMPLEX || model == AMDKGE_HOLE || model == AMDKGE_ROTATE) ? 2 * k : k;
}

}  // namespace kge
