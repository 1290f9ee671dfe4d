// evaluate(): 1-vs-all ranking for gfx950.  Replaces AbstractScoringLayer.get_ranks
// (/root/reference/ampligraph/latent_features/layers/scoring/AbstractScoringLayer.py:156-422) and the
// five `_get_{subject,object}_corruption_scores` (TransE.py:56-114, DistMult.py:51-99,
// ComplEx.py:65-151, HolE.py:47-89, RotatE.py:107-217) without ever materialising the reference's
// (n, m, K) broadcast temporaries or the (n, m) score matrix:
//
//   rank_prep   : per test triple, the quantised positive score q(pos) and the side's "query vector"
//                 (everything of the corruption score that does not depend on the corrupting entity,
//                 rounded exactly where the reference rounds it, e.g. DistMult fl(p*o)).
//   rank_counts : LDS-tiled (64 queries x 64 entities x 16 units) score tile with a
//                 quantise -> compare -> count epilogue; only two int32 counters per triple leave the CU.
//   rank_filter : per triple, recomputes the few true-positive corruptions with the SAME k-ordered
//                 accumulation chain (bitwise the scores the tile kernel produced) and counts those that
//                 outrank the positive (always "<=", AbstractScoringLayer.py:292-303).
//   rank_compose: tie strategy + filter subtraction + 1 (ScoringBasedEmbeddingModel.py:1684).
#include <stdlib.h>
#include <type_traits>

#include "kge_host.h"

namespace kge {

enum { MODE_DOT = 0, MODE_L1 = 1, MODE_ROT_O = 2, MODE_ROT_S = 3, MODE_L1_SUB = 4 };   // L1: |q + e| (subject side), L1_SUB: |q - e| (object side)

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<MODE_DOT>   { static constexpr int NQF = 1, NEF = 1; };
template <> struct ModeTraits<MODE_L1>    { static constexpr int NQF = 1, NEF = 1; };
template <> struct ModeTraits<MODE_L1_SUB> { static constexpr int NQF = 1, NEF = 1; };
template <> struct ModeTraits<MODE_ROT_O> { static constexpr int NQF = 2, NEF = 2; };
template <> struct ModeTraits<MODE_ROT_S> { static constexpr int NQF = 4, NEF = 2; };

// RotatE's per-unit modulus: the hardware v_sqrt_f32 (1 ulp).  One sqrt per (query, entity, unit) is what bounds RotatE's
// evaluation; its 1-ulp error is below the fp32 summation-order noise the ranks already tolerate (oracle.fragile_rank_mask),
// and the tile and the filter kernel share this function, so they still agree bit for bit.
__device__ __forceinline__ float rank_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// acc + |d| as ONE v_add_f32 with the abs source modifier.  Written as plain C the SLP vectoriser pairs the accumulations into
// v_pk_add_f32, which has no abs modifier, and pays a v_and_b32 per unit for it: 4 issue slots per 2 units (pk sub, 2 and, pk add)
// instead of 3 (pk sub, 2 add-abs).  Same IEEE operations, same order: bitwise identical scores.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float add_abs(float acc, float d) {
    float r;
    asm("v_add_f32 %0, %1, |%2|" : "=v"(r) : "v"(acc), "v"(d));
    return r;
}

// One unit of the corruption score, accumulated in unit order.  Shared by the tile kernel and the
// filter kernel so that both produce bitwise identical scores (compiled with -ffp-contract=off).
template <int MODE>
__device__ __forceinline__ float rank_op(float acc, const float (&q)[ModeTraits<MODE>::NQF],
                                         const float (&e)[ModeTraits<MODE>::NEF], float sgn) {
    if constexpr (MODE == MODE_DOT) {
        return fmaf(q[0], e[0], acc);
    } else if constexpr (MODE == MODE_L1) {
        return add_abs(acc, q[0] + e[0]);  // subject side: e + (p - o)        (TransE.py:77-83)
    } else if constexpr (MODE == MODE_L1_SUB) {
        return add_abs(acc, q[0] - e[0]);  // object side: (s + p) - e         (TransE.py:107-113); the sign is a template
                                           // parameter, not a multiply: 2 VALU instructions per unit instead of 3
    } else if constexpr (MODE == MODE_ROT_O) {
        const float re = q[0] - e[0], im = q[1] - e[1];   // RotatE.py:209-214
        return acc + rank_sqrt(re * re + im * im);
    } else {
        // q = (cos, sin, o_re, o_im) ; RotatE.py:151-160
        const float re = e[0] * q[0] - e[1] * q[1] - q[2];
        const float im = e[0] * q[1] + e[1] * q[0] - q[3];
        return acc + rank_sqrt(re * re + im * im);
    }
}

struct RankGeom {
    int U;        // units accumulated per (query, entity)
    int eplane;   // float offset between entity planes (re/im halves), 0 if NEF == 1
    int qplane;   // float offset between query planes
    int QW;       // floats per query row in the workspace
    int K;        // floats per table row
    float sgn;    // MODE_L1: +1 subject side, -1 object side
};

__host__ __device__ inline int mode_of(int model, int side) {
    if (model == AMDKGE_TRANSE) return side == AMDKGE_SIDE_S ? MODE_L1 : MODE_L1_SUB;
    if (model == AMDKGE_ROTATE) return side == AMDKGE_SIDE_S ? MODE_ROT_S : MODE_ROT_O;
    return MODE_DOT;
}

// RotatE: 0 (default) = exact mode (correctly rounded modulus, rank_rot_kernel / sqrt_rn), 1 = the 1-ulp hardware v_sqrt_f32
// in the generic tile kernel (amdkge_set_rank_rotate_fast).  The tile and the filter kernel of one mode share their chain.
static int g_rotate_fast = 0;

inline RankGeom geom_of(const amdkge_model* m, int side) {
    RankGeom g{};
    // stored layout (include/amdkge.h): the zero padding units add exact zeros to every accumulation chain (fmaf(0, 0, acc),
    // acc + |0|, acc + sqrt(0)), so the chains of a padded and of a dense table produce the same bits
    const int ks = stored_k(m);
    g.K = row_floats(m);
    const int mode = mode_of(m->scoring_type, side);
    if (mode == MODE_DOT || mode == MODE_L1 || mode == MODE_L1_SUB) { g.U = g.K; g.eplane = 0; g.qplane = 0; g.QW = g.K; }
    else {
        // exact mode walks the LIVE units only: a padding unit's modulus is sqrt(0), outside the fast sequence's domain
        g.U = g_rotate_fast ? ks : m->k; g.eplane = ks; g.qplane = ks; g.QW = (mode == MODE_ROT_S ? 4 : 2) * ks;
    }
    g.sgn = (side == AMDKGE_SIDE_S) ? 1.f : -1.f;
    return g;
}

__device__ __forceinline__ int quantise(float score) {
    return (int)(score * 1000.0f);   // AbstractScoringLayer.py:201 tf.cast(score * 1e3, int32): truncation
}

// ------------------------------------------------------------------------------------------------
// prep: one wave per test triple
// ------------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(256) void rank_prep_kernel(const float* __restrict__ ent, const float* __restrict__ rel,
                                                        const int32_t* __restrict__ triples, int64_t n, int k, int K,
                                                        int side, int QW, ModelConst mc, float* __restrict__ Q,
                                                        int* __restrict__ qpos) {
    constexpr int NC = ModelTraits<MODEL>::NC;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* rs = ent + (int64_t)triples[3 * i + 0] * K;
    const float* rp = rel + (int64_t)triples[3 * i + 1] * K;
    const float* ro = ent + (int64_t)triples[3 * i + 2] * K;
    float* q = Q + i * (int64_t)QW;
    float part = 0.f;
    for (int c = lane; c < k; c += KGE_WAVE) {
        float s[NC], p[NC], o[NC];
#pragma unroll
        for (int h = 0; h < NC; ++h) { s[h] = rs[c + h * k]; p[h] = rp[c + h * k]; o[h] = ro[c + h * k]; }
        prep_rel_exact<MODEL>(mc, p);   // RotatE: correctly rounded cos / sin (kge_device.h)
        part += score_unit<MODEL>(s, p, o);
        if constexpr (MODEL == AMDKGE_TRANSE) {
            q[c] = (side == AMDKGE_SIDE_S) ? (p[0] - o[0]) : (s[0] + p[0]);           // TransE.py:77-83,107-113
        } else if constexpr (MODEL == AMDKGE_DISTMULT) {
            q[c] = (side == AMDKGE_SIDE_S) ? (p[0] * o[0]) : (s[0] * p[0]);           // DistMult.py:71-73,96-98
        } else if constexpr (MODEL == AMDKGE_COMPLEX) {
            if (side == AMDKGE_SIDE_S) {   // ComplEx.py:93-107
                q[c] = p[0] * o[0] + p[1] * o[1];
                q[c + k] = p[0] * o[1] - p[1] * o[0];
            } else {                        // ComplEx.py:138-150
                q[c] = s[0] * p[0] - s[1] * p[1];
                q[c + k] = s[1] * p[0] + s[0] * p[1];
            }
        } else {
            if (side == AMDKGE_SIDE_S) {   // RotatE.py:151-160: needs cos, sin, o_re, o_im per unit
                q[c] = p[0]; q[c + k] = p[1]; q[c + 2 * k] = o[0]; q[c + 3 * k] = o[1];
            } else {                        // RotatE.py:209-212
                q[c] = s[0] * p[0] - s[1] * p[1];
                q[c + k] = s[0] * p[1] + s[1] * p[0];
            }
        }
    }
    const float tot = wave_sum(part);
    if (lane == 0) qpos[i] = quantise(mc.score_sign * mc.score_scale * tot);
}

// ------------------------------------------------------------------------------------------------
// tile kernel
// ------------------------------------------------------------------------------------------------
constexpr int QT = 64, ET = 64, KT = 16, LDP = 68;   // LDP: padded LDS row (floats), keeps float4 reads aligned

}  // namespace kge
#include "kge_rank_early.h"   // part 1: the distance models' exact early exit (thresholds, the check-point protocol)
namespace kge {

struct CountArgs {
    const float* ent;
    const float* Q;
    const int* qpos;
    const int32_t* ent_ids;
    int32_t* counts;
    int64_t n;
    int64_t ent_lo, ent_hi;
    int ent_per_block;
    RankGeom g;
    float sgn_scale;
    int qtiles, splits;   // MFMA kernel: logical grid, decoded from a 1-D XCD-aware launch
    float* scores;        // STORE variant of the VALU tile kernel: [n][ld] un-quantised scores instead of counts
    int64_t ld;
    const int* guard;     // a kernel launched as the fall-back of the screening / early-exit pass: runs only if *guard != 0 (guard_mode
                          // refines this for the early-exit path: see guard_says_run in kge_rank_early.h)
    int guard_mode;       // GUARD_* ; 0 with guard != NULL means GUARD_FLAG
    const int* e_probe;   // the early-exit probe's {decided, sampled}
    // EARLY variants of the distance models' tile kernels (kge_rank_early.h)
    EarlyList e_list;         // the list undecided pairs are handed to
    const uint8_t* e_qbad;    // [n] / [candidates]: rows whose pairs must not be decided early (non-finite or huge values)
    const uint8_t* e_ebad;
    int e_check, e_cost;      // stages between two checks; relative cost of a re-checked pair
};

template <int MODE, bool V4, bool STORE = false, bool EARLY = false>
__global__ __launch_bounds__(256) void rank_count_kernel(CountArgs a) {
    constexpr int NQF = ModeTraits<MODE>::NQF, NEF = ModeTraits<MODE>::NEF;
    static_assert(!EARLY || ((MODE == MODE_L1 || MODE == MODE_L1_SUB) && V4 && !STORE), "early exit: the TransE count kernels");
    __shared__ __attribute__((aligned(16))) float Qs[NQF][KT][LDP];
    __shared__ __attribute__((aligned(16))) float Es[NEF][KT][LDP];
    __shared__ EarlyShared es_;   // (referenced by the EARLY variants only: elsewhere it is never allocated)
    if (a.guard_mode ? !guard_says_run(a.guard_mode, a.guard, a.e_probe) : (a.guard && *a.guard == 0)) return;   // a launch that turned out not to be needed

    const int tid = threadIdx.x;
    const int tq = tid >> 4, te = tid & 15;
    const int64_t q0 = (int64_t)blockIdx.x * QT;
    const int64_t e_begin = a.ent_lo + (int64_t)blockIdx.y * a.ent_per_block;
    const int64_t e_end = min(a.ent_hi, e_begin + a.ent_per_block);

    int qp[4] = {0, 0, 0, 0};
    if constexpr (!STORE) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int64_t qi = q0 + tq * 4 + x;
            qp[x] = a.qpos[qi < a.n ? qi : a.n - 1];
        }
    }
    int cgt[4] = {0, 0, 0, 0}, ceq[4] = {0, 0, 0, 0};
    // EARLY: per query the partial sum beyond which the pair is decided (it can no longer reach the positive's quantised score);
    // +inf for a query row that must not be decided early; rows beyond n take no part
    float thr[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t qvalid = 0u;
    if constexpr (EARLY) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int64_t qi = q0 + tq * 4 + x;
            thr[x] = early_threshold(qp[x], a.sgn_scale);
            if (qi < a.n) { qvalid |= 0xFu << (4 * x); if (a.e_qbad[qi]) thr[x] = INFINITY; }
        }
        if (tid == 0) es_.n = 0;
    }

    // loader mapping: row = tid / 4 (0..63), 4-unit group = tid % 4
    const int lrow = tid >> 2, lgrp = tid & 3;
    const int64_t lq = q0 + lrow;
    const float* qrow = a.Q + (lq < a.n ? lq : a.n - 1) * (int64_t)a.g.QW;

    for (int64_t et = e_begin; et < e_end; et += ET) {
        const int64_t le = et + lrow;
        const int64_t le_c = le < e_end ? le : e_end - 1;
        const int64_t erow_id = a.ent_ids ? (int64_t)a.ent_ids[le_c] : le_c;
        const float* erow = a.ent + erow_id * a.g.K;
        float acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = 0.f;
        // EARLY: which of the thread's 16 pairs exist at all (query < n, candidate inside the range) and which must stay undecided
        uint32_t pvalid = 0u, pkeep = 0u;
        bool ended = false;
        if constexpr (EARLY) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int64_t ej = et + te * 4 + y;
                if (ej < e_end) { pvalid |= 0x1111u << y; if (a.e_ebad[ej - a.ent_lo]) pkeep |= 0x1111u << y; }
            }
            pvalid &= qvalid;
            pkeep &= pvalid;
        }

        for (int k0 = 0; k0 < a.g.U; k0 += KT) {
            const int ku = k0 + lgrp * 4;
            // ---- global -> LDS (transposed: [plane][unit][row]) ----
#pragma unroll
            for (int f = 0; f < NQF; ++f) {
                float v[4];
                if (V4 && ku + 3 < a.g.U) {
                    const float4 t = *reinterpret_cast<const float4*>(qrow + f * a.g.qplane + ku);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (ku + u < a.g.U) ? qrow[f * a.g.qplane + ku + u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) Qs[f][lgrp * 4 + u][lrow] = v[u];
            }
#pragma unroll
            for (int f = 0; f < NEF; ++f) {
                float v[4];
                if (V4 && ku + 3 < a.g.U) {
                    const float4 t = *reinterpret_cast<const float4*>(erow + f * a.g.eplane + ku);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (ku + u < a.g.U) ? erow[f * a.g.eplane + ku + u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) Es[f][lgrp * 4 + u][lrow] = v[u];
            }
            __syncthreads();
            // ---- 4x4 micro tile over the KT units, strictly in unit order ----
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                float qv[NQF][4], ev[NEF][4];
#pragma unroll
                for (int f = 0; f < NQF; ++f) {
                    const float4 t = *reinterpret_cast<const float4*>(&Qs[f][kk][tq * 4]);
                    qv[f][0] = t.x; qv[f][1] = t.y; qv[f][2] = t.z; qv[f][3] = t.w;
                }
#pragma unroll
                for (int f = 0; f < NEF; ++f) {
                    const float4 t = *reinterpret_cast<const float4*>(&Es[f][kk][te * 4]);
                    ev[f][0] = t.x; ev[f][1] = t.y; ev[f][2] = t.z; ev[f][3] = t.w;
                }
                if constexpr (MODE == MODE_L1 || MODE == MODE_L1_SUB) {
                    // rank_op's two operations with the first one packed: v_pk_add_f32 forms q +- e for two entities at
                    // once (q broadcast through op_sel), add_abs accumulates each -- 3 issue slots per 2 units
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; y += 2) {
                            const f32x2 qq = {qv[0][x], qv[0][x]}, ee = {ev[0][y], ev[0][y + 1]};
                            const f32x2 d = (MODE == MODE_L1) ? qq + ee : qq - ee;
                            acc[x][y] = add_abs(acc[x][y], d.x);
                            acc[x][y + 1] = add_abs(acc[x][y + 1], d.y);
                        }
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            float qq[NQF], ee[NEF];
#pragma unroll
                            for (int f = 0; f < NQF; ++f) qq[f] = qv[f][x];
#pragma unroll
                            for (int f = 0; f < NEF; ++f) ee[f] = ev[f][y];
                            acc[x][y] = rank_op<MODE>(acc[x][y], qq, ee, a.g.sgn);
                        }
                }
            }
            // EARLY, every e_check stages (not behind the last one): count the pairs that are still undecided; the stage's own
            // barrier publishes the four wave sums
            bool chk = false;
            uint32_t und = 0u;
            if constexpr (EARLY) {
                chk = ((k0 / KT + 1) % a.e_check == 0) && (k0 + KT < a.g.U);
                if (chk) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) und |= (acc[x][y] > thr[x]) ? 0u : (1u << (4 * x + y));   // (NaN: undecided)
                    und = (und | pkeep) & pvalid;
                    const int c = wave_sum_i(__popc(und));
                    if ((tid & 63) == 0) es_.red[tid >> 6] = c;
                }
            }
            __syncthreads();
            if constexpr (EARLY) {
                int total;
                if (chk && early_decide(es_, k0 + KT, a.g.U, a.e_cost, total)) {
                    early_spill(es_, a.e_list, und, total, q0 + tq * 4, et + te * 4 - a.ent_lo);
                    ended = true;
                    break;
                }
            }
        }
        if constexpr (EARLY) { if (ended) continue; }   // decided or handed over: nothing of this tile is counted here
        if constexpr (STORE) {
            // ---- epilogue of the STORE variant: the scores themselves (discovery: top-k / nearest neighbours) ----
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int64_t qi = q0 + tq * 4 + x;
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int64_t ej = et + te * 4 + y;
                    if (qi < a.n && ej < e_end) a.scores[qi * a.ld + (ej - a.ent_lo)] = a.sgn_scale * acc[x][y];
                }
            }
            continue;
        }
        // ---- epilogue: quantise, compare, count ----
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const bool valid = (et + te * 4 + y) < e_end;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int q = quantise(a.sgn_scale * acc[x][y]);
                cgt[x] += (valid && qp[x] < q) ? 1 : 0;
                ceq[x] += (valid && qp[x] == q) ? 1 : 0;
            }
        }
    }
    if constexpr (STORE) return;
    // reduce over the 16 lanes (te) that share the same queries, one atomic pair per query per block
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        int g = cgt[x], e = ceq[x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
        const int64_t qi = q0 + tq * 4 + x;
        if (te == 0 && qi < a.n) {
            if (g) atomicAdd(&a.counts[2 * qi + 0], g);
            if (e) atomicAdd(&a.counts[2 * qi + 1], e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RotatE, exact mode (the default): the tile kernel with the per-unit modulus CORRECTLY ROUNDED, so that the whole chain
//     acc = fl(acc + sqrt_rn(fl(fl(re * re) + fl(im * im))))          (RotatE.py:151-160,209-214, unit order)
// is a function of the inputs alone and a CPU restatement (oracle/csrc/rank_ordered.c) reproduces the ranks bit for bit.
// Same tiling as rank_count_kernel (64 queries x 64 entities, 16 units per LDS stage, 4 x 4 micro tile per thread); the
// arithmetic is written on PAIRS of entities so that it issues as packed fp32 (v_pk_add / v_pk_mul / v_pk_fma_f32: two lanes
// of work per slot) and the modulus is sqrt_rn's fast sequence without its branch: v_rsq_f32 + 4 packed operations per pair.
// Its domain (x >= 2^-100: exhaustively verified, see sqrt_rn in kge_device.h) is checked per entity tile and costs half a
// slot per unit: every thread keeps the maximum of its v_rsq results (x < 2^-100, zero or denormal <=> g > 2^50) and looks at
// its 16 accumulators (x = inf or NaN poisons them); if anything in the WORKGROUP is outside, the tile is redone with libm's
// sqrtf.  Padding units of the stored layout are not walked at all (U = the model's k: their x is an exact 0, which is
// outside the fast domain); live units with re = im = 0 exactly (a corruption that coincides with the rotated subject in
// both components) take the slow path and are the only realistic trigger.
// ------------------------------------------------------------------------------------------------
template <bool SLOW, bool SUBJ>
__device__ __forceinline__ void rot_micro(const float (&qv)[SUBJ ? 4 : 2][4], const float (&ev)[2][4], f32x2 (&acc)[4][2], float& gmax) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const f32x2 e0 = {ev[0][2 * y], ev[0][2 * y + 1]}, e1 = {ev[1][2 * y], ev[1][2 * y + 1]};
            f32x2 re, im;
            if constexpr (!SUBJ) {   // q = s o r                                    RotatE.py:209-214
                const f32x2 q0 = {qv[0][x], qv[0][x]}, q1 = {qv[1][x], qv[1][x]};
                re = q0 - e0;
                im = q1 - e1;
            } else {                 // q = (cos, sin, o_re, o_im)                   RotatE.py:151-160
                const f32x2 c = {qv[0][x], qv[0][x]}, sn = {qv[1][x], qv[1][x]}, orr = {qv[2][x], qv[2][x]}, oi = {qv[3][x], qv[3][x]};
                re = e0 * c - e1 * sn - orr;
                im = e0 * sn + e1 * c - oi;
            }
            const f32x2 xx = re * re + im * im;
            f32x2 m;
            if constexpr (SLOW) {
                m.x = sqrtf(xx.x);
                m.y = sqrtf(xx.y);
            } else {
                f32x2 g;
                g.x = __builtin_amdgcn_rsqf(xx.x);
                g.y = __builtin_amdgcn_rsqf(xx.y);
                gmax = fmaxf(fmaxf(gmax, g.x), g.y);
                const f32x2 yv = xx * g, h = g * 0.5f;
                const f32x2 r = __builtin_elementwise_fma(-yv, yv, xx);
                m = __builtin_elementwise_fma(r, h, yv);
            }
            acc[x][y] = acc[x][y] + m;
        }
}

template <bool SUBJ, bool STORE, bool EARLY = false>
__global__ __launch_bounds__(256) void rank_rot_kernel(CountArgs a) {
    constexpr int NQF = SUBJ ? 4 : 2, NEF = 2;
    static_assert(!EARLY || !STORE, "early exit: the count form only");
    __shared__ __attribute__((aligned(16))) float Qs[NQF][KT][LDP];
    __shared__ __attribute__((aligned(16))) float Es[NEF][KT][LDP];
    __shared__ EarlyShared es_;   // (referenced by the EARLY variants only: elsewhere it is never allocated)
    if (a.guard_mode ? !guard_says_run(a.guard_mode, a.guard, a.e_probe) : (a.guard && *a.guard == 0)) return;   // a launch that turned out not to be needed

    const int tid = threadIdx.x;
    const int tq = tid >> 4, te = tid & 15;
    const int64_t q0 = (int64_t)blockIdx.x * QT;
    const int64_t e_begin = a.ent_lo + (int64_t)blockIdx.y * a.ent_per_block;
    const int64_t e_end = min(a.ent_hi, e_begin + a.ent_per_block);
    const int U = a.g.U;   // live units; the planes of Q and of a table row are a.g.qplane / a.g.eplane (stored width) apart

    int qp[4] = {0, 0, 0, 0};
    if constexpr (!STORE) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int64_t qi = q0 + tq * 4 + x;
            qp[x] = a.qpos[qi < a.n ? qi : a.n - 1];
        }
    }
    int cgt[4] = {0, 0, 0, 0}, ceq[4] = {0, 0, 0, 0};
    float thr[4] = {0.f, 0.f, 0.f, 0.f};   // EARLY: see rank_count_kernel
    uint32_t qvalid = 0u;
    if constexpr (EARLY) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int64_t qi = q0 + tq * 4 + x;
            thr[x] = early_threshold(qp[x], a.sgn_scale);
            if (qi < a.n) { qvalid |= 0xFu << (4 * x); if (a.e_qbad[qi]) thr[x] = INFINITY; }
        }
        if (tid == 0) es_.n = 0;
    }

    const int lrow = tid >> 2, lgrp = tid & 3;   // loader: row 0..63, 4-unit group 0..3
    const int64_t lq = q0 + lrow;
    const float* qrow = a.Q + (lq < a.n ? lq : a.n - 1) * (int64_t)a.g.QW;

    for (int64_t et = e_begin; et < e_end; et += ET) {
        const int64_t le = et + lrow;
        const int64_t le_c = le < e_end ? le : e_end - 1;
        const int64_t erow_id = a.ent_ids ? (int64_t)a.ent_ids[le_c] : le_c;
        const float* erow = a.ent + erow_id * a.g.K;
        f32x2 acc[4][2];
        float gmax = 0.f;
        uint32_t pvalid = 0u, pkeep = 0u;   // EARLY: the thread's existing pairs / those that must stay undecided (bit 4 x + y)
        if constexpr (EARLY) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int64_t ej = et + te * 4 + y;
                if (ej < e_end) { pvalid |= 0x1111u << y; if (a.e_ebad[ej - a.ent_lo]) pkeep |= 0x1111u << y; }
            }
            pvalid &= qvalid;
            pkeep &= pvalid;
        }

        // returns true when the tile ended early (EARLY, fast form only): its undecided pairs are on the list
        auto run_tile = [&](auto slow_c) __attribute__((always_inline)) -> bool {
            constexpr bool SLOW = decltype(slow_c)::value;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = f32x2{0.f, 0.f};
            for (int k0 = 0; k0 < U; k0 += KT) {
                // ---- global -> LDS, transposed to [plane][unit][row]; a 4-unit group that starts inside the row is loaded
                //      whole (the stored row is a whole number of float4s; what lies beyond U is never multiplied) ----
                const int ku = k0 + lgrp * 4;
                const bool in = ku < U;
#pragma unroll
                for (int f = 0; f < NQF; ++f) {
                    const float4 t = in ? *reinterpret_cast<const float4*>(qrow + f * a.g.qplane + ku) : make_float4(0.f, 0.f, 0.f, 0.f);
                    Qs[f][lgrp * 4 + 0][lrow] = t.x; Qs[f][lgrp * 4 + 1][lrow] = t.y; Qs[f][lgrp * 4 + 2][lrow] = t.z; Qs[f][lgrp * 4 + 3][lrow] = t.w;
                }
#pragma unroll
                for (int f = 0; f < NEF; ++f) {
                    const float4 t = in ? *reinterpret_cast<const float4*>(erow + f * a.g.eplane + ku) : make_float4(0.f, 0.f, 0.f, 0.f);
                    Es[f][lgrp * 4 + 0][lrow] = t.x; Es[f][lgrp * 4 + 1][lrow] = t.y; Es[f][lgrp * 4 + 2][lrow] = t.z; Es[f][lgrp * 4 + 3][lrow] = t.w;
                }
                __syncthreads();
                auto unit = [&](int kk) __attribute__((always_inline)) {
                    float qv[NQF][4], ev[NEF][4];
#pragma unroll
                    for (int f = 0; f < NQF; ++f) {
                        const float4 t = *reinterpret_cast<const float4*>(&Qs[f][kk][tq * 4]);
                        qv[f][0] = t.x; qv[f][1] = t.y; qv[f][2] = t.z; qv[f][3] = t.w;
                    }
#pragma unroll
                    for (int f = 0; f < NEF; ++f) {
                        const float4 t = *reinterpret_cast<const float4*>(&Es[f][kk][te * 4]);
                        ev[f][0] = t.x; ev[f][1] = t.y; ev[f][2] = t.z; ev[f][3] = t.w;
                    }
                    rot_micro<SLOW, SUBJ>(qv, ev, acc, gmax);
                };
                if (U - k0 >= KT) {
#pragma unroll
                    for (int kk = 0; kk < KT; ++kk) unit(kk);
                } else {
                    for (int kk = 0; kk < U - k0; ++kk) unit(kk);   // the row's last, partial stage: live units only
                }
                bool chk = false;
                uint32_t und = 0u;
                if constexpr (EARLY && !SLOW) {
                    chk = ((k0 / KT + 1) % a.e_check == 0) && (k0 + KT < U);
                    if (chk) {
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int y = 0; y < 4; ++y) {
                                const float sc = (y & 1) ? acc[x][y >> 1].y : acc[x][y >> 1].x;
                                und |= (sc > thr[x]) ? 0u : (1u << (4 * x + y));   // (NaN: undecided)
                            }
                        und = (und | pkeep) & pvalid;
                        const int c = wave_sum_i(__popc(und));
                        // a modulus outside the fast form's domain so far (the partial sums cannot be trusted): no exit for this tile
                        const bool dom = __ballot(!(gmax <= 0x1p50f)) != 0ull;
                        if ((tid & 63) == 0) es_.red[tid >> 6] = c | (dom ? (1 << 30) : 0);
                    }
                }
                __syncthreads();
                if constexpr (EARLY && !SLOW) {
                    int total;
                    if (chk && early_decide(es_, k0 + KT, U, a.e_cost, total)) {
                        early_spill(es_, a.e_list, und, total, q0 + tq * 4, et + te * 4 - a.ent_lo);
                        return true;
                    }
                }
            }
            return false;
        };
        if (run_tile(std::false_type{})) continue;
        bool bad = !(gmax <= 0x1p50f);
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) bad |= !(fabsf(acc[x][y].x) < INFINITY) || !(fabsf(acc[x][y].y) < INFINITY);
        if (__syncthreads_or(bad ? 1 : 0)) run_tile(std::true_type{});

        if constexpr (STORE) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int64_t qi = q0 + tq * 4 + x;
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int64_t ej = et + te * 4 + y;
                    const float sc = (y & 1) ? acc[x][y >> 1].y : acc[x][y >> 1].x;
                    if (qi < a.n && ej < e_end) a.scores[qi * a.ld + (ej - a.ent_lo)] = a.sgn_scale * sc;
                }
            }
            continue;
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const bool valid = (et + te * 4 + y) < e_end;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float sc = (y & 1) ? acc[x][y >> 1].y : acc[x][y >> 1].x;
                const int q = quantise(a.sgn_scale * sc);
                cgt[x] += (valid && qp[x] < q) ? 1 : 0;
                ceq[x] += (valid && qp[x] == q) ? 1 : 0;
            }
        }
    }
    if constexpr (STORE) return;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        int g = cgt[x], e = ceq[x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
        const int64_t qi = q0 + tq * 4 + x;
        if (te == 0 && qi < a.n) {
            if (g) atomicAdd(&a.counts[2 * qi + 0], g);
            if (e) atomicAdd(&a.counts[2 * qi + 1], e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA tile kernel for the contraction models (DistMult / ComplEx / HolE: score = query . entity row).
// v_mfma_f32_32x32x2_f32 is exact fp32 and bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md,
// "FP32-input MFMA"), i.e. it produces the very bits of rank_op<MODE_DOT> accumulated in unit order: the
// VALU tile kernel above, this kernel and the filter kernel stay bitwise interchangeable.
//   workgroup = 4 waves = 128 queries x 128 entities; each wave owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator
//   registers); K is streamed through LDS 32 units at a time in [unit][row] layout (the lane->operand map of
//   the instruction, A[i = l & 31][k = l >> 5], then reads consecutive LDS words), next stage prefetched into
//   registers while the current one is multiplied; epilogue = quantise -> compare with q(pos) -> packed count.
// ------------------------------------------------------------------------------------------------
#ifndef KGE_MLD
#define KGE_MLD 132
#endif
constexpr int MQ = 128, ME = 128, MK = 32, MLD = KGE_MLD;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr size_t MFMA_LDS_BYTES = (size_t)2 * 2 * MK * MLD * sizeof(float) + MQ * sizeof(int);

template <bool V4>
__global__ __launch_bounds__(256, 2) void rank_count_mfma_kernel(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_rank[];
    typedef float (*tile_t)[MK][MLD];
    tile_t Qs = reinterpret_cast<tile_t>(smem_rank);                                     // [2][MK][MLD]
    tile_t Es = reinterpret_cast<tile_t>(smem_rank + (size_t)2 * MK * MLD * sizeof(float));
    int* qps = reinterpret_cast<int*>(smem_rank + (size_t)4 * MK * MLD * sizeof(float));

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = (wv >> 1) * 64, we = (wv & 1) * 64;   // this wave's 64 x 64 corner of the 128 x 128 tile
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware work order (speed only, no correctness dependence): workgroup b lands on XCD b % 8 (round-robin
    // dispatch), every XCD has its own 4 MB L2.  XCD c takes a contiguous eighth of the query tiles and walks it
    // in groups of 8 query tiles x all entity splits, so the ~64 workgroups resident on one XCD at a time are
    // 8 query tiles x 8 entity ranges: their Q and E slabs (~3 MB) are shared through that L2 instead of each
    // workgroup streaming its own from the Infinity Cache.
    int bx, by;
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
    const int64_t q0 = (int64_t)bx * MQ;
    const int64_t e_begin = a.ent_lo + (int64_t)by * a.ent_per_block;
    const int64_t e_end = min(a.ent_hi, e_begin + a.ent_per_block);
    const int U = a.g.U;
    const int S = (U + MK - 1) / MK;                       // LDS stages per tile
    const int64_t ntile = (e_end - e_begin + ME - 1) / ME;

    if (tid < MQ) { const int64_t qi = q0 + tid; qps[tid] = a.qpos[qi < a.n ? qi : a.n - 1]; }

    // loader: float4 f = tid + 256 * i, i < 4 : row = f >> 3 (128 rows), 4-unit group = f & 7 (8 groups = 32 units)
    const float* qrow[4];
    const float* erow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t lq = q0 + ((tid + 256 * i) >> 3);
        qrow[i] = a.Q + (lq < a.n ? lq : a.n - 1) * (int64_t)a.g.QW + ((tid + 256 * i) & 7) * 4;
    }
    auto set_erow = [&](int64_t et) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t le = et + ((tid + 256 * i) >> 3);
            const int64_t le_c = le < e_end ? le : e_end - 1;
            const int64_t id = a.ent_ids ? (int64_t)a.ent_ids[le_c] : le_c;
            erow[i] = a.ent + id * a.g.K + ((tid + 256 * i) & 7) * 4;
        }
    };
    auto fetch = [&](const float* src, int ku) -> float4 {   // 4 consecutive units starting at ku, zero beyond U
        if (V4) {
            if (ku < U) return *reinterpret_cast<const float4*>(src);
            return make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 t;
        t.x = (ku + 0 < U) ? src[0] : 0.f; t.y = (ku + 1 < U) ? src[1] : 0.f;
        t.z = (ku + 2 < U) ? src[2] : 0.f; t.w = (ku + 3 < U) ? src[3] : 0.f;
        return t;
    };
    float4 pq[4], pe[4];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ku = k0 + ((tid + 256 * i) & 7) * 4;
            pq[i] = fetch(qrow[i] + k0, ku);
            pe[i] = fetch(erow[i] + k0, ku);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kg = (f & 7) * 4;
            Qs[buf][kg + 0][row] = pq[i].x; Qs[buf][kg + 1][row] = pq[i].y; Qs[buf][kg + 2][row] = pq[i].z; Qs[buf][kg + 3][row] = pq[i].w;
            Es[buf][kg + 0][row] = pe[i].x; Es[buf][kg + 1][row] = pe[i].y; Es[buf][kg + 2][row] = pe[i].z; Es[buf][kg + 3][row] = pe[i].w;
        }
    };

    int cnt[2][16];   // per (query tile mi, accumulator register): gt | eq << 16 over this lane's entity columns
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[mi][r] = 0;

    // The (tile, stage) sequence is one software pipeline: while stage s is multiplied out of LDS buffer `buf`,
    // the global loads of the next stage (possibly the first stage of the NEXT entity tile) are in flight and are
    // written to the other buffer afterwards: one workgroup barrier per stage.
    set_erow(e_begin);
    load_stage(0);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (int64_t t = 0; t < ntile; ++t) {
        const int64_t et = e_begin + t * ME;
        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int st = 0; st < S; ++st) {
            const bool last_stage = (st == S - 1);
            const bool has_next = !(last_stage && t == ntile - 1);
            if (has_next) {
                if (last_stage) set_erow(et + ME);
                load_stage(last_stage ? 0 : (st + 1) * MK);
            }
            // operands of unit pair kk + 2 are read from LDS before the four MFMAs of pair kk are issued, so the
            // LDS latency hides behind 256 cycles of matrix-pipe work even for a lone wave on the SIMD
            float opa[2][2], opb[2][2];
            opa[0][0] = Qs[buf][lh][wq + l31]; opa[0][1] = Qs[buf][lh][wq + 32 + l31];
            opb[0][0] = Es[buf][lh][we + l31]; opb[0][1] = Es[buf][lh][we + 32 + l31];
#pragma unroll
            for (int kk = 0; kk < MK; kk += 2) {
                const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
                if (kk + 2 < MK) {
                    opa[nxt][0] = Qs[buf][kk + 2 + lh][wq + l31]; opa[nxt][1] = Qs[buf][kk + 2 + lh][wq + 32 + l31];
                    opb[nxt][0] = Es[buf][kk + 2 + lh][we + l31]; opb[nxt][1] = Es[buf][kk + 2 + lh][we + 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][1], acc[1][1], 0, 0, 0);
            }
            if (has_next) store_stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        // ---- epilogue: C/D map col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const bool valid = (et + we + ni * 32 + l31) < e_end;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qp = qps[wq + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                    const int q = quantise(a.sgn_scale * acc[mi][ni][r]);
                    cnt[mi][r] += (valid && qp < q) ? 1 : 0;
                    cnt[mi][r] += (valid && qp == q) ? 0x10000 : 0;
                }
        }
    }
    // ---- per query row: sum over the 32 lanes that share it, one atomic pair per row per wave ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int g = cnt[mi][r] & 0xFFFF, e = cnt[mi][r] >> 16;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
            const int64_t qi = q0 + wq + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (l31 == 0 && qi < a.n) {
                if (g) atomicAdd(&a.counts[2 * qi + 0], g);
                if (e) atomicAdd(&a.counts[2 * qi + 1], e);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// The same tile computation as rank_count_mfma_kernel<true>, organised as ONE instruction stream per stage in which every
// non-matrix instruction sits between two MFMAs.  A wave issues in order, so whatever is placed after the four MFMAs of a
// unit pair only starts when the last of them has been issued; in the kernel above the global prefetch (with its bounds
// branches), the 32 transposing ds_write_b32 and the barrier therefore run with the matrix pipe idle (MfmaUtil 0.70).  Here:
//   * loads are two stages ahead (two register sets): stage g issues the global loads of stage g + 2 during its first four
//     unit pairs and writes the set loaded during stage g - 1 to the other LDS buffer during its last eight pairs, one or
//     two instructions behind each MFMA; a stage of matrix work (>= 4 096 cycles) covers the load latency;
//   * no branches inside a stage: out-of-range units read the row start and are zeroed by a select, the load cursor runs
//     past the last stage onto clamped addresses instead of being guarded;
//   * operands of pair kk + 2 are read behind the first two MFMAs of pair kk.
// Same MFMA order per accumulator => the same bits as the kernel above and as rank_op<MODE_DOT>.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void rank_count_mfma_pipe_kernel(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_rank[];
    typedef float (*tile_t)[MK][MLD];
    tile_t Qs = reinterpret_cast<tile_t>(smem_rank);                                     // [2][MK][MLD]
    tile_t Es = reinterpret_cast<tile_t>(smem_rank + (size_t)2 * MK * MLD * sizeof(float));
    int* qps = reinterpret_cast<int*>(smem_rank + (size_t)4 * MK * MLD * sizeof(float));

    if (a.guard && *a.guard == 0) return;   // (screened call that did not overflow its recheck list: nothing to do)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wq = (wv >> 1) * 64, we = (wv & 1) * 64;
    const int l31 = lane & 31, lh = lane >> 5;
    int bx, by;   // XCD-aware work order, see rank_count_mfma_kernel
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
    const int64_t q0 = (int64_t)bx * MQ;
    const int64_t e_begin = a.ent_lo + (int64_t)by * a.ent_per_block;
    const int64_t e_end = min(a.ent_hi, e_begin + a.ent_per_block);
    const int U = a.g.U;
    const int S = (U + MK - 1) / MK;
    const int64_t ntile = (e_end - e_begin + ME - 1) / ME;
    const int64_t G = ntile * S;

    if (tid < MQ) { const int64_t qi = q0 + tid; qps[tid] = a.qpos[qi < a.n ? qi : a.n - 1]; }

    // loader: float4 f = tid + 256 * i, i < 4 : row = f >> 3 (128 rows), 4-unit group kg = (f & 7) * 4 (32 units)
    const int kg = (tid & 7) * 4, lrow = tid >> 3;   // (tid + 256 i) & 7 == tid & 7 ; row = lrow + 32 i
    const float* qbase[4];
    const float* ebase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t lq = q0 + lrow + 32 * i;
        qbase[i] = a.Q + (lq < a.n ? lq : a.n - 1) * (int64_t)a.g.QW;
    }
    auto set_erow = [&](int64_t et) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t le = et + lrow + 32 * i;
            const int64_t le_c = le < e_end ? le : e_end - 1;
            const int64_t id = a.ent_ids ? (int64_t)a.ent_ids[le_c] : le_c;
            ebase[i] = a.ent + id * a.g.K;
        }
    };
    int ld_k0 = 0;          // load cursor: unit offset of the stage the next loads belong to ...
    int64_t ld_tile = 0;    // ... and its entity tile (clamped to the last one once the cursor runs past the end)
    // raw load of 4 units; units beyond U read the row start instead and are zeroed when the registers go to LDS (a select
    // right here would make the wave wait for the load it has just issued)
    auto fetch = [&](const float* base, int k0) -> float4 {
        const int ku = k0 + kg;
        return *reinterpret_cast<const float4*>(base + (ku < U ? ku : 0));
    };
    auto advance = [&]() {
        ld_k0 += MK;
        if (ld_k0 >= S * MK) {
            ld_k0 = 0;
            ld_tile = (ld_tile + 1 < ntile) ? ld_tile + 1 : ntile - 1;
            set_erow(e_begin + ld_tile * ME);
        }
    };
    float4 pq[2][4], pe[2][4];

    int cnt[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[mi][r] = 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // prologue: stage 0 -> LDS buffer 0, stage 1 -> register set 1
    set_erow(e_begin);
#pragma unroll
    for (int i = 0; i < 4; ++i) { pq[0][i] = fetch(qbase[i], 0); pe[0][i] = fetch(ebase[i], 0); }
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) { pq[1][i] = fetch(qbase[i], ld_k0); pe[1][i] = fetch(ebase[i], ld_k0); }
    advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lrow + 32 * i;
        const float z = (kg < U) ? 1.f : 0.f;   // U % 4 == 0: a 4-unit group is inside or outside as a whole
        Qs[0][kg + 0][row] = z * pq[0][i].x; Qs[0][kg + 1][row] = z * pq[0][i].y; Qs[0][kg + 2][row] = z * pq[0][i].z; Qs[0][kg + 3][row] = z * pq[0][i].w;
        Es[0][kg + 0][row] = z * pe[0][i].x; Es[0][kg + 1][row] = z * pe[0][i].y; Es[0][kg + 2][row] = z * pe[0][i].z; Es[0][kg + 3][row] = z * pe[0][i].w;
    }
    __syncthreads();

    int st = 0;
    int64_t t = 0;
    auto stage = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;   // LDS buffer of this stage == register set that is free for new loads
        constexpr int OTH = SET ^ 1;                  // register set holding the next stage's data == LDS buffer it goes to
        const bool okn = ((st + 1 == S) ? 0 : (st + 1) * MK) + kg < U;   // is this lane's unit group of the NEXT stage inside the row?
        float opa[2][2], opb[2][2];
        opa[0][0] = Qs[SET][lh][wq + l31]; opa[0][1] = Qs[SET][lh][wq + 32 + l31];
        opb[0][0] = Es[SET][lh][we + l31]; opb[0][1] = Es[SET][lh][we + 32 + l31];
        __builtin_amdgcn_sched_barrier(0);
        // unit pairs 0..7: the four MFMAs of a pair, each followed by its share of the stage's other work
#pragma unroll
        for (int it = 0; it < MK / 4; ++it) {
            const int kk = 2 * it, cur = it & 1, nxt = cur ^ 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][0], acc[0][0], 0, 0, 0);
            opa[nxt][0] = Qs[SET][kk + 2 + lh][wq + l31]; opb[nxt][0] = Es[SET][kk + 2 + lh][we + l31];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][1], acc[0][1], 0, 0, 0);
            opa[nxt][1] = Qs[SET][kk + 2 + lh][wq + 32 + l31]; opb[nxt][1] = Es[SET][kk + 2 + lh][we + 32 + l31];
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][0], acc[1][0], 0, 0, 0);
            if (it < 4) {
                pq[SET][it] = fetch(qbase[it], ld_k0);
            } else {
                const int i = it - 4, row = lrow + 32 * i;
                Qs[OTH][kg + 0][row] = okn ? pq[OTH][i].x : 0.f; Qs[OTH][kg + 1][row] = okn ? pq[OTH][i].y : 0.f;
                Qs[OTH][kg + 2][row] = okn ? pq[OTH][i].z : 0.f; Qs[OTH][kg + 3][row] = okn ? pq[OTH][i].w : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][1], acc[1][1], 0, 0, 0);
            if (it < 4) {
                pe[SET][it] = fetch(ebase[it], ld_k0);
            } else {
                const int i = it - 4, row = lrow + 32 * i;
                Es[OTH][kg + 0][row] = okn ? pe[OTH][i].x : 0.f; Es[OTH][kg + 1][row] = okn ? pe[OTH][i].y : 0.f;
                Es[OTH][kg + 2][row] = okn ? pe[OTH][i].z : 0.f; Es[OTH][kg + 3][row] = okn ? pe[OTH][i].w : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // unit pairs 8..15: nothing but matrix work and operand reads -- skipped as a whole when the stage holds <= 16 real
        // units (the zero-padded half of a row's last stage: U = 400 -> 16 of 32 units; acc + 0 * 0 == acc)
        if (U - st * MK > MK / 2) {
#pragma unroll
            for (int it = MK / 4; it < MK / 2; ++it) {
                const int kk = 2 * it, cur = it & 1, nxt = cur ^ 1;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][0], acc[0][0], 0, 0, 0);
                if (kk + 2 < MK) { opa[nxt][0] = Qs[SET][kk + 2 + lh][wq + l31]; opb[nxt][0] = Es[SET][kk + 2 + lh][we + l31]; }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][0], opb[cur][1], acc[0][1], 0, 0, 0);
                if (kk + 2 < MK) { opa[nxt][1] = Qs[SET][kk + 2 + lh][wq + 32 + l31]; opb[nxt][1] = Es[SET][kk + 2 + lh][we + 32 + l31]; }
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[cur][1], opb[cur][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        advance();
        __syncthreads();
        if (++st == S) {   // ---- tile epilogue: C/D map col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
            const int64_t et = e_begin + t * ME;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bool valid = (et + we + ni * 32 + l31) < e_end;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qp = qps[wq + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                        const int q = quantise(a.sgn_scale * acc[mi][ni][r]);
                        cnt[mi][r] += (valid && qp < q) ? 1 : 0;
                        cnt[mi][r] += (valid && qp == q) ? 0x10000 : 0;
                        acc[mi][ni][r] = 0.f;
                    }
            }
            st = 0;
            ++t;
        }
    };
    for (int64_t g = 0; g < G; g += 2) {
        stage(std::integral_constant<int, 0>{});
        if (g + 1 < G) stage(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int g = cnt[mi][r] & 0xFFFF, e = cnt[mi][r] >> 16;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
            const int64_t qi = q0 + wq + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (l31 == 0 && qi < a.n) {
                if (g) atomicAdd(&a.counts[2 * qi + 0], g);
                if (e) atomicAdd(&a.counts[2 * qi + 1], e);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// filter kernel: one wave per test triple, one lane per true-positive id
// ------------------------------------------------------------------------------------------------
struct FilterArgs {
    const float* ent;
    const float* Q;
    const int* qpos;
    const int64_t* flt_lo;
    const int64_t* flt_hi;
    const int32_t* flt_ids;
    const int32_t* subset_pos;
    int32_t* sub;
    int64_t n;
    int64_t ent_lo, ent_hi;
    RankGeom g;
    float sgn_scale;
    const int* guard;    // non-NULL: run only if *guard != 0 (the pair list of the contraction models' filter pass overflowed)
};

// Filter pass, contraction models: the (query, known positive) pairs as a flat list for rank_recheck_kernel<true> -- 64 pairs
// per wave with coalesced row fetches, instead of one wave per query whose lanes each walk a whole row 16 bytes at a time
// (at C2 a query has 1.1 known positives on average: 63 idle lanes, 100 dependent load steps: 150 us).  A block takes 256
// queries, scans their list lengths, reserves its run of the list with ONE atomic and writes it cooperatively (pair j of the
// block: its query by binary search in the scanned offsets), so a query with thousands of known positives is no slower than
// thousands of queries with one.  An id outside the candidate set is listed as (query, -1).
__global__ __launch_bounds__(256) void filter_pairs_kernel(FilterArgs a, int2* __restrict__ pairs, int* __restrict__ counter, int64_t cap) {
    __shared__ long long off_s[257];
    __shared__ long long lo_s[256];
    __shared__ long long base_s;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    long long lo = 0, c = 0;
    if (i < a.n) { lo = a.flt_lo[i]; c = a.flt_hi[i] - lo; if (c < 0) c = 0; }
    lo_s[tid] = lo;
    off_s[tid + 1] = c;
    if (tid == 0) off_s[0] = 0;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {   // inclusive scan of the 256 lengths (off_s[1..256])
        const long long v = (tid >= o) ? off_s[tid + 1 - o] : 0;
        __syncthreads();
        off_s[tid + 1] += v;
        __syncthreads();
    }
    const long long total = off_s[256];
    if (tid == 0) {
        long long b = -1;
        if (total > 0 && total <= cap) b = (long long)atomicAdd(counter, (int)total);
        if (total > cap || (b >= 0 && b + total > cap)) { counter[1] = 1; b = -1; }
        base_s = b;
    }
    __syncthreads();
    const long long base = base_s;
    if (base < 0) return;
    for (long long j = tid; j < total; j += 256) {
        int x = 0, y = 256;   // the query q with off_s[q] <= j < off_s[q + 1]
        while (y - x > 1) { const int mid = (x + y) >> 1; if (off_s[mid] <= j) x = mid; else y = mid; }
        const int64_t q = (int64_t)blockIdx.x * 256 + x;
        int64_t id = (int64_t)a.flt_ids[lo_s[x] + (j - off_s[x])];
        bool ok;
        if (a.subset_pos) {   // mapping_dict.lookup + drop -1 (AbstractScoringLayer.py:266-275)
            const int pos = a.subset_pos[id];
            ok = pos >= 0 && pos >= a.ent_lo && pos < a.ent_hi;
        } else {
            ok = id >= a.ent_lo && id < a.ent_hi;   // partition rule :280-288
        }
        pairs[base + j] = make_int2((int)q, ok ? (int)id : -1);
    }
}

// One unit of RotatE's exact-mode chain for ONE (query, entity) pair: the operations of rot_micro, scalar (sqrt_rn == the
// packed sequence inside its domain, libm's sqrtf outside: bitwise the tile kernel's value either way).
template <int MODE>
__device__ __forceinline__ float rot_exact_op(float acc, const float (&q)[ModeTraits<MODE>::NQF], const float (&e)[2]) {
    float re, im;
    if constexpr (MODE == MODE_ROT_O) { re = q[0] - e[0]; im = q[1] - e[1]; }
    else { re = e[0] * q[0] - e[1] * q[1] - q[2]; im = e[0] * q[1] + e[1] * q[0] - q[3]; }
    return acc + sqrt_rn(re * re + im * im);
}

template <int MODE, bool V4, bool EXACT_ROT = false>
__global__ __launch_bounds__(256) void rank_filter_kernel(FilterArgs a) {
    constexpr int NQF = ModeTraits<MODE>::NQF, NEF = ModeTraits<MODE>::NEF;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;
    if (a.guard && *a.guard == 0) return;
    const int64_t lo = a.flt_lo[i], hi = a.flt_hi[i];
    const float* qrow = a.Q + i * (int64_t)a.g.QW;
    const int qp = a.qpos[i];
    int cnt = 0;
    for (int64_t f0 = lo; f0 < hi; f0 += KGE_WAVE) {
        const int64_t f = f0 + lane;
        bool ok = f < hi;
        int64_t id = ok ? (int64_t)a.flt_ids[f] : 0;
        if (ok && a.subset_pos) {   // mapping_dict.lookup + drop -1 (AbstractScoringLayer.py:266-275)
            const int pos = a.subset_pos[id];
            ok = pos >= 0;
            // the corruption row of position `pos` is the table row `id` itself
            if (ok) ok = (pos >= a.ent_lo) && (pos < a.ent_hi);
        } else if (ok) {
            ok = (id >= a.ent_lo) && (id < a.ent_hi);   // partition rule :280-288
        }
        const float* erow = a.ent + (ok ? id : 0) * a.g.K;
        float acc = 0.f;
        if constexpr (EXACT_ROT) {   // live units only (a.g.U = k), float4 loads inside the stored (padded) row
            for (int u0 = 0; u0 < a.g.U; u0 += 4) {
                float qv[NQF][4], ev[NEF][4];
#pragma unroll
                for (int p = 0; p < NQF; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(qrow + p * a.g.qplane + u0);
                    qv[p][0] = t.x; qv[p][1] = t.y; qv[p][2] = t.z; qv[p][3] = t.w;
                }
#pragma unroll
                for (int p = 0; p < NEF; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(erow + p * a.g.eplane + u0);
                    ev[p][0] = t.x; ev[p][1] = t.y; ev[p][2] = t.z; ev[p][3] = t.w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u0 + u >= a.g.U) break;
                    float qq[NQF], ee[NEF];
#pragma unroll
                    for (int p = 0; p < NQF; ++p) qq[p] = qv[p][u];
#pragma unroll
                    for (int p = 0; p < NEF; ++p) ee[p] = ev[p][u];
                    acc = rot_exact_op<MODE>(acc, qq, ee);
                }
            }
        } else if (V4) {
            for (int u0 = 0; u0 < a.g.U; u0 += 4) {
                float qv[NQF][4], ev[NEF][4];
#pragma unroll
                for (int p = 0; p < NQF; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(qrow + p * a.g.qplane + u0);
                    qv[p][0] = t.x; qv[p][1] = t.y; qv[p][2] = t.z; qv[p][3] = t.w;
                }
#pragma unroll
                for (int p = 0; p < NEF; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(erow + p * a.g.eplane + u0);
                    ev[p][0] = t.x; ev[p][1] = t.y; ev[p][2] = t.z; ev[p][3] = t.w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float qq[NQF], ee[NEF];
#pragma unroll
                    for (int p = 0; p < NQF; ++p) qq[p] = qv[p][u];
#pragma unroll
                    for (int p = 0; p < NEF; ++p) ee[p] = ev[p][u];
                    acc = rank_op<MODE>(acc, qq, ee, a.g.sgn);
                }
            }
        } else {
            for (int u = 0; u < a.g.U; ++u) {
                float qq[NQF], ee[NEF];
#pragma unroll
                for (int p = 0; p < NQF; ++p) qq[p] = qrow[p * a.g.qplane + u];
#pragma unroll
                for (int p = 0; p < NEF; ++p) ee[p] = erow[p * a.g.eplane + u];
                acc = rank_op<MODE>(acc, qq, ee, a.g.sgn);
            }
        }
        const int q = quantise(a.sgn_scale * acc);
        cnt += (ok && qp <= q) ? 1 : 0;
    }
    cnt = wave_sum_i(cnt);
    if (lane == 0 && cnt) atomicAdd(&a.sub[i], cnt);
}

__global__ void rank_compose_kernel(const int32_t* counts, const int32_t* sub, int64_t n, int strategy,
                                    int32_t* ranks, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int gt = counts[2 * i], eq = counts[2 * i + 1];
    int r;
    if (strategy == AMDKGE_RANK_BEST) r = gt;                       // AbstractScoringLayer.py:221-227
    else if (strategy == AMDKGE_RANK_MIDDLE) r = gt + (eq + 1) / 2; // :232-244 ceil(#equal / 2)
    else r = gt + eq;                                               // :252-258
    if (sub) r -= sub[i];
    ranks[i * stride] = r + 1;                                      // ScoringBasedEmbeddingModel.py:1684
}

}  // namespace kge
#include "kge_rank_screen.h"
#include "kge_rank_screen_r.h"
constexpr int SCREEN_KERNEL_DEFAULT = 4;   // (see run_screen: rank_screen_kernel_r where it applies -- rows of 4 .. 13 slabs --, rank_screen_kernel_v1 elsewhere)
#define KGE_RANK_EARLY_PART2
#include "kge_rank_early.h"   // part 2: workspace, row flags, the exact recheck of the distance models
namespace kge {

static inline char* align_up(char* p, size_t a) { return (char*)(((uintptr_t)p + a - 1) & ~(uintptr_t)(a - 1)); }

struct Workspace {
    float* Q;
    int* qpos;
    int* flt_counter;    // filter pass, contraction models: [0] pairs listed, [1] overflow flag
    int2* flt_pairs;     // (query, table row of a known positive)
    int64_t flt_cap;
};

static int64_t query_row_floats(const amdkge_model* m) { return (m->scoring_type == AMDKGE_ROTATE) ? 4ll * stored_k(m) : row_floats(m); }
static int64_t filter_pair_cap(int64_t n) {   // 64 known positives per query on average; the list counter is an int32
    const int64_t c = n * 64 > 65536 ? n * 64 : 65536;
    return c < (1ll << 30) ? c : (1ll << 30);
}

static Workspace carve(void* d_work, const amdkge_model* m, int64_t n) {
    Workspace w;
    char* p = align_up((char*)d_work, 256);
    w.qpos = (int*)p;
    p = align_up(p + n * sizeof(int), 256);
    w.Q = (float*)p;
    p = align_up(p + n * query_row_floats(m) * sizeof(float), 256);
    w.flt_counter = (int*)p;
    w.flt_pairs = (int2*)(p + 256);
    w.flt_cap = filter_pair_cap(n);
    return w;
}

static int run_prep(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples, int64_t n,
                    int side, const RankGeom& g, const Workspace& w, hipStream_t st) {
    const ModelConst mc = model_const(m);
    const unsigned grid = (unsigned)((n + 3) / 4);
#define KGE_PREP(M) hipLaunchKernelGGL((rank_prep_kernel<M>), dim3(grid), dim3(256), 0, st, d_ent, d_rel, d_triples, n, stored_k(m), g.K, side, g.QW, mc, w.Q, w.qpos)
    switch (m->scoring_type) {
        case AMDKGE_TRANSE: KGE_PREP(AMDKGE_TRANSE); break;
        case AMDKGE_DISTMULT: KGE_PREP(AMDKGE_DISTMULT); break;
        case AMDKGE_COMPLEX:
        case AMDKGE_HOLE: KGE_PREP(AMDKGE_COMPLEX); break;
        default: KGE_PREP(AMDKGE_ROTATE); break;
    }
#undef KGE_PREP
    return check_launch("rank_prep");
}

static inline bool sgn_scale_positive(const ModelConst& mc) { return mc.score_sign * mc.score_scale > 0.f; }

// the screening sequence of one rank_counts call (see kge_rank_screen.h); counts of decided + rechecked pairs are merged into
// d_counts unless the recheck list overflowed (flag at counter[1]: the guarded exact kernel then produces them)
static int run_screen(const amdkge_model* m, const float* d_ent, const int32_t* d_ent_ids, int64_t ent_lo, int64_t mcand, int64_t n,
                      const RankGeom& g, const Workspace& w, const ModelConst& mc, int32_t* d_counts, void* d_screen, size_t screen_bytes,
                      hipStream_t st) {
    ScreenBufs b = carve_screen(d_screen, screen_bytes, n, mcand, g.U);
    const float sgn_scale = mc.score_sign * mc.score_scale;
    if (hipError_t e = hipMemsetAsync(b.counter, 0, 256 + scr_up((size_t)n * 8), st)) return set_error_hip(e, "hipMemsetAsync(screen counters)");
    const double u = ldexp(1.0, -24), gam = u * (1.0 + 2.0 * (double)g.U * u);   // x |W q|_2 |W e|_2: the chain's rounding bound
    hipLaunchKernelGGL(rank_limbs_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, w.Q, (int64_t)g.QW, (const int32_t*)nullptr, (int64_t)0, n, g.U, b.S,
                       (float)(gam * (1.0 + 1e-6)), b.qlimbs, b.qm, (const int*)nullptr);
    if (int rc = check_launch("rank_limbs(Q)")) return rc;
    // Which screening kernel: rank_screen_kernel_r (round 6, kge_rank_screen_r.h: one wave per SIMD, the query limbs resident in registers,
    // candidates on one scale per tile of 64) for rows of 4 .. 13 slabs -- 97 .. 416 int8 units: ComplEx k = 50 .. 208, DistMult k = 97 .. 416 (BASELINE's ComplEx k = 200 and
    // DistMult k = 400 are 13-slab rows; the reference's published DistMult k = 350 is 11) --,
    // rank_screen_kernel_v1 (rounds 3 - 5: query fragments L2 -> registers, entity slab register-staged through LDS) for every other width
    // and behind kernel r for wild tables.  The same counts either way; AMDKGE_SCREEN_KERNEL=1 pins v1 for A/B runs (read once).  The
    // variants that measured slower or no faster live in scripts/experiments/: round 5's register-staged LDS form, round 6's LDS-DMA
    // ring for both operands (g) and the paired-wave split of the limb products (p).
    static const int screen_kernel_env = [] { const char* ev = getenv("AMDKGE_SCREEN_KERNEL"); const int v = ev ? atoi(ev) : 0; return (v == 1 || v == 4) ? v : SCREEN_KERNEL_DEFAULT; }();
    int screen_kernel = screen_kernel_env;
    if (screen_kernel == 4 && (b.S < 4 || b.S > 13 || b.cap * 8 < mcand * 16)) screen_kernel = 1;   // (the instantiated widths; room for the row records)
    if (screen_kernel == 4) {
        // (the row records of the first pass live in the head of the pair list, unused until the screening kernel)
        float4* const stats = reinterpret_cast<float4*>(b.pairs);
        hipLaunchKernelGGL(rank_rowstats_kernel, dim3((unsigned)((mcand + 3) / 4)), dim3(256), 0, st, d_ent, (int64_t)g.K, d_ent_ids, ent_lo, mcand, g.U, stats);
        if (int rc = check_launch("rank_rowstats(E)")) return rc;
        hipLaunchKernelGGL(rank_limbs_tile_kernel, dim3((unsigned)(16 * ((mcand + 63) / 64))), dim3(256), 0, st, d_ent, (int64_t)g.K, d_ent_ids, ent_lo, mcand, g.U, b.S,
                           (const float4*)stats, b.elimbs, b.em, b.tm, b.counter);
        if (int rc = check_launch("rank_limbs_tile(E)")) return rc;
        // (a wild table -- see screen_wild -- is redone on per-row scales for rank_screen_kernel_v1; otherwise this launch returns at once)
        hipLaunchKernelGGL(rank_limbs_kernel, dim3((unsigned)std::min<int64_t>((mcand + 3) / 4, 512)), dim3(256), 0, st, d_ent, (int64_t)g.K, d_ent_ids, ent_lo, mcand, g.U, b.S, 1.f,
                           b.elimbs, b.em, (const int*)b.counter);
    } else
        hipLaunchKernelGGL(rank_limbs_kernel, dim3((unsigned)((mcand + 3) / 4)), dim3(256), 0, st, d_ent, (int64_t)g.K, d_ent_ids, ent_lo, mcand, g.U, b.S, 1.f,
                           b.elimbs, b.em, (const int*)nullptr);
    if (int rc = check_launch("rank_limbs(E)")) return rc;
    hipLaunchKernelGGL(rank_thresholds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.qpos, n, sgn_scale, b.qt);
    if (int rc = check_launch("rank_thresholds")) return rc;
    ScreenArgs sa{};
    sa.b = b; sa.n = n; sa.m = mcand; sa.U = g.U;
    // per unit: the three dropped limb products (2^23 + 2^14) and the cross term of the two fixed-point roundings (1/4), in units of
    // A B; + 2^25: the fp32 reconstruction of the 40-bit integer sum in the epilogue (inner sum rounds by <= 2^8 in units of 2^16 A B)
    // ... + the fp32 rebuild of f = L0 2^16 + L1 2^8 + L2 in the epilogue, in units of f: |L1| <= U 2^15 and |L2| <= 3 U 2^14 exceed
    // 2^24 for U > 512, so their conversions round too (half an ulp: <= 2 resp. 4 for U <= 2048, the first times 2^8), and the
    // inner fma rounds at up to 2^34 (half an ulp: 2^10): 2^9 + 4 + 2^10 < 2^11, i.e. 2^27 A B (the screening condition holds
    // U <= 2048; the outer fma's rounding is relative to |f| and sits in the thresholds)
    sa.drop = (float)(((double)g.U * (8388608.0 + 16384.0 + 0.25) + 134217728.0) * (1.0 + 1e-6));
    const int64_t qtiles = (n + SCR_Q - 1) / SCR_Q, etiles = (mcand + SCR_ET - 1) / SCR_ET;
    // Each block takes a run of entity tiles of one 128-query block.  The run length is the one with the shortest schedule: rounds of
    // `slots` co-resident blocks x (tiles + a block's start-up in tile-times) -- v1 / g: two workgroups per CU, ~0.35 (at C2, 160 x 227
    // tiles, runs of 4 give 18 rounds of 4: 78 tile-times instead of 84 with runs of 9); r: one per CU, and its 39 KB of query limbs come
    // first (~1 tile-time).
    static const int64_t run_cap = [] { const char* ev = getenv("AMDKGE_SCREEN_RUN"); const int v = ev ? atoi(ev) : 0; return (int64_t)(v > 0 ? v : 64); }();   // (A/B runs: longest run of tiles per block)
    auto schedule = [&](ScreenArgs& x, int64_t slots, double startup, int64_t max_run, int64_t& nblk) -> bool {
        int64_t tiles_per = 1;
        const int64_t qt8 = 8 * ((qtiles + 7) / 8), lim = std::min(etiles < run_cap ? etiles : run_cap, max_run);
        double best = 1e300;
        for (int64_t tp = 1; tp <= lim; ++tp) {
            const int64_t blocks = qt8 * ((etiles + tp - 1) / tp);
            const double cost = (double)((blocks + slots - 1) / slots) * ((double)tp + startup);
            if (cost <= best) { best = cost; tiles_per = tp; }   // (ties: the longer run)
        }
        // very large problems: keep the launch below 2^31 blocks and a lane's 16-bit counters (2 candidates per tile) in range
        while (tiles_per < etiles && tiles_per < max_run && qt8 * ((etiles + tiles_per - 1) / tiles_per) > (1ll << 24)) tiles_per *= 2;
        if (tiles_per > max_run) tiles_per = max_run;
        if (tiles_per > etiles) tiles_per = etiles;
        if (tiles_per < 1) tiles_per = 1;
        const int64_t splits = (etiles + tiles_per - 1) / tiles_per;
        x.ent_per_block = (int)(tiles_per * SCR_ET); x.qtiles = (int)qtiles; x.splits = (int)splits;
        nblk = qt8 * splits;
        return nblk <= 0x7FFFFFFFll;
    };
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        if (hipError_t e = hipFuncSetAttribute((const void*)rank_screen_kernel_v1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCR_LDS_BYTES))
            return set_error_hip(e, "hipFuncSetAttribute(rank_screen_v1)");
        if (hipError_t e = hipFuncSetAttribute((const void*)rank_screen_kernel_v1_wild, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCR_LDS_BYTES))
            return set_error_hip(e, "hipFuncSetAttribute(rank_screen_v1_wild)");
        for (const void* f : {(const void*)rank_screen_kernel_r<13>, (const void*)rank_screen_kernel_r<12>, (const void*)rank_screen_kernel_r<11>, (const void*)rank_screen_kernel_r<10>,
                              (const void*)rank_screen_kernel_r<9>, (const void*)rank_screen_kernel_r<8>, (const void*)rank_screen_kernel_r<7>, (const void*)rank_screen_kernel_r<6>,
                              (const void*)rank_screen_kernel_r<5>, (const void*)rank_screen_kernel_r<4>})
            if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCRR_LDS_BYTES))
                return set_error_hip(e, "hipFuncSetAttribute(rank_screen_r)");
        attr_done.done();
    }
    int64_t nblk = 0;
    if (screen_kernel == 4) {
        ScreenArgs sr = sa;
        sr.wild_mode = 2;
        if (!schedule(sr, 256, 1.0, SCRR_TMCAP, nblk)) return set_error(AMDKGE_EUNSUPPORTED, "rank_counts: too many tiles for one launch");
        switch (b.S) {
#define KGE_SCR_R(N) case N: hipLaunchKernelGGL(rank_screen_kernel_r<N>, dim3((unsigned)nblk), dim3(SCR_THREADS), SCRR_LDS_BYTES, st, sr); break
            KGE_SCR_R(13); KGE_SCR_R(12); KGE_SCR_R(11); KGE_SCR_R(10); KGE_SCR_R(9); KGE_SCR_R(8); KGE_SCR_R(7); KGE_SCR_R(6); KGE_SCR_R(5);
            default: hipLaunchKernelGGL(rank_screen_kernel_r<4>, dim3((unsigned)nblk), dim3(SCR_THREADS), SCRR_LDS_BYTES, st, sr); break;
#undef KGE_SCR_R
        }
        if (int rc = check_launch("rank_screen_r")) return rc;
    }
    if (!schedule(sa, 512, 0.35, 16384, nblk)) return set_error(AMDKGE_EUNSUPPORTED, "rank_counts: too many tiles for one launch");
    sa.nblk = (int)nblk;
    if (screen_kernel == 4)   // (the per-row-scale kernel behind rank_screen_kernel_r: a wild table only)
        hipLaunchKernelGGL(rank_screen_kernel_v1_wild, dim3((unsigned)std::min<int64_t>(nblk, 512)), dim3(SCR_THREADS), SCR_LDS_BYTES, st, sa);
    else hipLaunchKernelGGL(rank_screen_kernel_v1, dim3((unsigned)nblk), dim3(SCR_THREADS), SCR_LDS_BYTES, st, sa);
    if (int rc = check_launch("rank_screen")) return rc;
    RecheckArgs ra{};
    ra.ent = d_ent; ra.Q = w.Q; ra.qpos = w.qpos; ra.ent_ids = d_ent_ids; ra.ent_lo = ent_lo; ra.U = g.U; ra.K = g.K; ra.QW = g.QW;
    ra.sgn_scale = sgn_scale; ra.b = b;
    static PerDeviceOnce rck_attr;
    if (rck_attr.need()) {
        if (hipError_t e = hipFuncSetAttribute((const void*)rank_recheck_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCK_LDS_BYTES))
            return set_error_hip(e, "hipFuncSetAttribute(rank_recheck)");
        rck_attr.done();
    }
    hipLaunchKernelGGL(rank_recheck_kernel<false>, dim3(1024), dim3(256), RCK_LDS_BYTES, st, ra);
    if (int rc = check_launch("rank_recheck")) return rc;
    hipLaunchKernelGGL(rank_screen_merge_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, st, b, n, d_counts);
    return check_launch("rank_screen_merge");
}

// The early-exit PROBE of one rank_counts call of a distance model (kge_rank_early.h): 4 096 sampled pairs, how many are decided at
// half their units.  Round 5: the answer is read back on the HOST (8 bytes, one stream synchronisation of ~15 us against a
// count pass of a millisecond and more) and only the kernel it picks is launched, with the geometry that suits IT.  (Remembering the
// answer per table address was tried and dropped: a model that trains between two evaluations keeps its address, and a stale "no"
// cost the planted TransE tables 10.7 -> 6.3 M ranks/s, profiles/r05f_*; the round trip itself is not what an untrained table's
// evaluation loses against the plain kernel alone -- see bench.py eval_bench on the order of the two measurements.)  Round 4 let
// the device decide: both tile kernels were launched with the early kernel's geometry (runs of >= 4 tiles) and one returned at
// once -- on tables where the exit does not fire the plain kernel then ran in a geometry that costs it 8 - 10 % (C2 shape, TransE
// k = 200: 7.37 vs 8.00 M ranks/s, profiles/r04u_models.jsonl) behind ~70 000 empty workgroups.
static int early_probe(int mode, const float* d_ent, const int32_t* d_ent_ids, int64_t ent_lo, int64_t mcand, int64_t n, const RankGeom& g,
                       const Workspace& w, float sgn_scale, void* d_screen, size_t screen_bytes, bool* yes, bool* measured, hipStream_t st) {
    *yes = true;
    *measured = false;   // (true: this call ran the probe kernel -- the workspace's probe words hold its counts)
    if (!g_early.probe) return AMDKGE_OK;   // (tests: the early-exit kernel always)
    EarlyBufs eb = carve_early(d_screen, screen_bytes, n, mcand);
    if (hipError_t e = hipMemsetAsync(eb.b.counter, 0, 256, st)) return set_error_hip(e, "hipMemsetAsync(early probe)");
    ProbeArgs pa{};
    pa.ent = d_ent; pa.Q = w.Q; pa.qpos = w.qpos; pa.ent_ids = d_ent_ids; pa.ent_lo = ent_lo; pa.m = mcand; pa.n = n; pa.g = g; pa.sgn_scale = sgn_scale;
    pa.probe = eb.b.counter + 4;
    switch (mode) {
        case MODE_L1: hipLaunchKernelGGL(rank_early_probe_kernel<MODE_L1>, dim3(16), dim3(256), 0, st, pa); break;
        case MODE_L1_SUB: hipLaunchKernelGGL(rank_early_probe_kernel<MODE_L1_SUB>, dim3(16), dim3(256), 0, st, pa); break;
        case MODE_ROT_S: hipLaunchKernelGGL(rank_early_probe_kernel<MODE_ROT_S>, dim3(16), dim3(256), 0, st, pa); break;
        default: hipLaunchKernelGGL(rank_early_probe_kernel<MODE_ROT_O>, dim3(16), dim3(256), 0, st, pa); break;
    }
    if (int rc = check_launch("rank_early_probe")) return rc;
    int h[2] = {0, 0};
    if (hipError_t e = hipMemcpyAsync(h, eb.b.counter + 4, sizeof(h), hipMemcpyDeviceToHost, st)) return set_error_hip(e, "hipMemcpyAsync(early probe)");
    if (hipError_t e = hipStreamSynchronize(st)) return set_error_hip(e, "hipStreamSynchronize(early probe)");
    *yes = h[0] * 2 >= h[1] && h[1] > 0;   // (early_probe_says_yes)
    *measured = true;
    return AMDKGE_OK;
}

// the early-exit sequence of one rank_counts call of a distance model (kge_rank_early.h): row flags, the EARLY tile kernel (counts
// of the tiles it finishes + the list of the pairs it hands over), the exact recheck of the list, the merge into the caller's
// counts (skipped when the list overflowed: the caller then runs the plain kernel behind the same flag).  `a`: the plain
// kernel's arguments (grid geometry included).  Called when the probe said yes.
static int run_early(int mode, const amdkge_model* m, const float* d_ent, const int32_t* d_ent_ids, int64_t ent_lo, int64_t mcand, int64_t n,
                     const RankGeom& g, const Workspace& w, CountArgs a, dim3 grid, void* d_screen, size_t screen_bytes, const int** guard_out,
                     bool probe_measured, hipStream_t st) {
    int32_t* const caller_counts = a.counts;
    EarlyBufs eb = carve_early(d_screen, screen_bytes, n, mcand);
    // counters [0 .. 3] and everything behind the probe words; words [4], [5] keep what the probe counted in THIS call (decided,
    // sampled: "yes" -- the host read them before this sequence was enqueued; the recheck and merge kernels still look at them), or
    // are set to 1, 1 when no probe ran (a remembered answer, or the probe switched off)
    if (hipError_t e = hipMemsetAsync(eb.b.counter, 0, 16, st)) return set_error_hip(e, "hipMemsetAsync(early counters)");
    if (hipError_t e = hipMemsetAsync(eb.b.counter + 8, 0, 256 - 32 + scr_up((size_t)n * 8), st)) return set_error_hip(e, "hipMemsetAsync(early counts)");
    if (!probe_measured)
        if (hipError_t e = hipMemsetD32Async((hipDeviceptr_t)(eb.b.counter + 4), 1, 2, st)) return set_error_hip(e, "hipMemsetD32Async(probe)");
    // rows that must not be decided early: the query vectors (every plane) and the candidate rows (stored width)
    hipLaunchKernelGGL(rank_rowflags_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, w.Q, (int64_t)g.QW, (const int32_t*)nullptr, (int64_t)0, n, g.QW, eb.qbad);
    if (int rc = check_launch("rank_rowflags(Q)")) return rc;
    hipLaunchKernelGGL(rank_rowflags_kernel, dim3((unsigned)((mcand + 3) / 4)), dim3(256), 0, st, d_ent, (int64_t)g.K, d_ent_ids, ent_lo, mcand, g.K, eb.ebad);
    if (int rc = check_launch("rank_rowflags(E)")) return rc;
    a.counts = eb.b.counts;
    a.guard = nullptr; a.guard_mode = GUARD_NONE; a.e_probe = eb.b.counter + 4;
    a.e_list = EarlyList{eb.b.counter, eb.b.pairs, eb.b.cap};
    a.e_qbad = eb.qbad; a.e_ebad = eb.ebad;
    a.e_cost = g_early.cost < 1 ? 1 : g_early.cost;
    const bool rot = mode == MODE_ROT_O || mode == MODE_ROT_S;
    a.e_check = rot ? g_early.check_rot : g_early.check_l1;
    const int nstages = (g.U + KT - 1) / KT;   // short rows: at least three checks per row
    if (a.e_check > nstages / 4) a.e_check = nstages / 4;
    if (a.e_check < 1) a.e_check = 1;
    RecheckDistArgs ra{};
    ra.ent = d_ent; ra.Q = w.Q; ra.qpos = w.qpos; ra.ent_ids = d_ent_ids; ra.ent_lo = ent_lo; ra.g = g; ra.sgn_scale = a.sgn_scale; ra.b = eb.b;
#define KGE_RD(MODE) do { \
        static PerDeviceOnce attr; \
        if (attr.need()) { \
            if (hipError_t e = hipFuncSetAttribute((const void*)rank_recheck_dist_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rd_lds_bytes<MODE>())) \
                return set_error_hip(e, "hipFuncSetAttribute(rank_recheck_dist)"); \
            attr.done(); \
        } \
        hipLaunchKernelGGL(rank_recheck_dist_kernel<MODE>, dim3(1024), dim3(256), rd_lds_bytes<MODE>(), st, ra); } while (0)
    switch (mode) {
        case MODE_L1:
            hipLaunchKernelGGL((rank_count_kernel<MODE_L1, true, false, true>), grid, dim3(256), 0, st, a);
            if (int rc = check_launch("rank_counts_early")) return rc;
            KGE_RD(MODE_L1); break;
        case MODE_L1_SUB:
            hipLaunchKernelGGL((rank_count_kernel<MODE_L1_SUB, true, false, true>), grid, dim3(256), 0, st, a);
            if (int rc = check_launch("rank_counts_early")) return rc;
            KGE_RD(MODE_L1_SUB); break;
        case MODE_ROT_S:
            hipLaunchKernelGGL((rank_rot_kernel<true, false, true>), grid, dim3(256), 0, st, a);
            if (int rc = check_launch("rank_counts_early")) return rc;
            KGE_RD(MODE_ROT_S); break;
        default:
            hipLaunchKernelGGL((rank_rot_kernel<false, false, true>), grid, dim3(256), 0, st, a);
            if (int rc = check_launch("rank_counts_early")) return rc;
            KGE_RD(MODE_ROT_O); break;
    }
#undef KGE_RD
    if (int rc = check_launch("rank_recheck_dist")) return rc;
    hipLaunchKernelGGL(rank_early_merge_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, st, eb.b, n, caller_counts);
    if (int rc = check_launch("rank_early_merge")) return rc;
    *guard_out = eb.b.counter + 1;
    return AMDKGE_OK;
}

}  // namespace kge

using namespace kge;

static int g_rank_kernel = 0;

extern "C" int amdkge_set_rank_kernel(int which) {
    if (which < 0 || which > 3) return set_error(AMDKGE_EINVAL, "set_rank_kernel: 0 = automatic, 1 = VALU tile kernel, 2 = first MFMA kernel, 3 = pipelined fp32 MFMA kernel without the int8 screening pass");
    g_rank_kernel = which;
    return AMDKGE_OK;
}

extern "C" int amdkge_set_rank_early(int on, int check_l1, int check_rot, int cost, int probe) {
    g_early.on = on ? 1 : 0;
    if (probe >= 0) g_early.probe = probe ? 1 : 0;
    if (check_l1 > 0) g_early.check_l1 = check_l1;
    if (check_rot > 0) g_early.check_rot = check_rot;
    if (cost > 0) g_early.cost = cost;
    return AMDKGE_OK;
}

extern "C" int amdkge_set_rank_rotate_fast(int fast) {
    g_rotate_fast = fast ? 1 : 0;
    return AMDKGE_OK;
}

extern "C" int64_t amdkge_rank_workspace_bytes(const amdkge_model* m, int64_t n) {
    if (validate_model(m) != AMDKGE_OK || n < 0) return -1;
    return 1024 + ((n * 4 + 255) / 256) * 256 + n * query_row_floats(m) * 4 + 512 + filter_pair_cap(n) * 8;
}

static int rank_counts_impl(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                            int64_t n, int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                            int32_t* d_counts, void* d_work, void* d_screen, int64_t screen_bytes, void* stream);

extern "C" int amdkge_rank_counts(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                                  int64_t n, int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                                  int32_t* d_counts, void* d_work, void* stream) {
    return rank_counts_impl(m, d_ent, d_rel, d_triples, n, side, d_ent_ids, ent_lo, ent_hi, d_counts, d_work, nullptr, 0, stream);
}

extern "C" int64_t amdkge_rank_screen_workspace_bytes(const amdkge_model* m, int64_t n, int64_t n_cand) {
    if (validate_model(m) != AMDKGE_OK || n < 0 || n_cand < 0) return -1;
    if (mode_of(m->scoring_type, AMDKGE_SIDE_S) != MODE_DOT) {
        // TransE / RotatE: the exact early exit (kge_rank_early.h) -- counters, counts, row flags and the list of handed-over pairs
        // (room for ~3 % of the comparisons; a full list falls back to the plain kernel)
        if (!g_early.on) return 0;
        int64_t pairs = n * n_cand / 32;
        if (pairs < (1 << 18)) pairs = 1 << 18;
        if (pairs > (1ll << 27)) pairs = 1ll << 27;
        return (int64_t)early_fixed_bytes(n, n_cand) + pairs * 8 + 512;
    }
    int64_t pairs = n * n_cand / 32;   // room for ~3 % of the comparisons (typically ~0.2 % are undecided)
    if (pairs < (1 << 20)) pairs = 1 << 20;
    return (int64_t)screen_fixed_bytes(n, n_cand, row_floats(m)) + pairs * 8 + 512;
}

extern "C" int amdkge_rank_counts_screened(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                                           int64_t n, int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                                           int32_t* d_counts, void* d_work, void* d_screen, int64_t screen_bytes, void* stream) {
    return rank_counts_impl(m, d_ent, d_rel, d_triples, n, side, d_ent_ids, ent_lo, ent_hi, d_counts, d_work, d_screen, screen_bytes, stream);
}

static int rank_counts_impl(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                            int64_t n, int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                            int32_t* d_counts, void* d_work, void* d_screen, int64_t screen_bytes, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (side != AMDKGE_SIDE_S && side != AMDKGE_SIDE_O) return set_error(AMDKGE_EINVAL, "rank_counts: side must be AMDKGE_SIDE_S or AMDKGE_SIDE_O");
    if (n < 0 || ent_lo < 0 || ent_hi < ent_lo) return set_error(AMDKGE_EINVAL, "rank_counts: bad sizes");
    if (!d_ent_ids && ent_hi > m->n_ents) return set_error(AMDKGE_EINVAL, "rank_counts: entity range outside the table");
    if (n == 0 || ent_hi == ent_lo) return AMDKGE_OK;
    if (!d_ent || !d_rel || !d_triples || !d_counts || !d_work) return set_error(AMDKGE_EINVAL, "rank_counts: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RankGeom g = geom_of(m, side);
    const Workspace w = carve(d_work, m, n);
    if (int rc = run_prep(m, d_ent, d_rel, d_triples, n, side, g, w, st)) return rc;

    const ModelConst mc = model_const(m);
    CountArgs a{};
    a.ent = d_ent; a.Q = w.Q; a.qpos = w.qpos; a.ent_ids = d_ent_ids; a.counts = d_counts; a.n = n;
    a.ent_lo = ent_lo; a.ent_hi = ent_hi; a.g = g; a.sgn_scale = mc.score_sign * mc.score_scale;
    const int mode = mode_of(m->scoring_type, side);
    const bool rot_exact = (mode == MODE_ROT_O || mode == MODE_ROT_S) && !g_rotate_fast;
    const bool v4 = (rot_exact || g.U % 4 == 0) && (g.eplane % 4 == 0) && (g.K % 4 == 0);
    const int force = g_rank_kernel;   // amdkge_set_rank_kernel (tests): 1 forces the VALU tile kernel, 2 the first MFMA kernel, 3 the pipelined one unscreened
    const bool mfma = (mode == MODE_DOT) && force != 1;
    const int qt = mfma ? MQ : QT, et_ = mfma ? ME : ET;
    const int64_t qtiles = (n + qt - 1) / qt;
    const int64_t etiles = (ent_hi - ent_lo + et_ - 1) / et_;
    // Entity tiles per block: the grid is (query tiles) x (entity splits).  VALU kernel: pick the split that minimises
    // (rounds of resident blocks) x (tiles per block), i.e. the tail of the last round, with a mild bias towards longer
    // blocks (one counter flush per block).  MFMA kernel: the two workgroups resident on a CU share its matrix pipes, so
    // what counts is the work of the busiest CU, ceil(blocks / 256) x (tiles per block + ~3/16 tile of prologue and
    // flush) -- and a CU needs a SUCCESSION of blocks to keep two of them out of phase: measured on C3 (23 x 320 tiles)
    // 5-10 tiles per block 1.96 ms, 15 (two blocks per CU, started together) 2.16 ms, 30 (one block per CU) 2.85 ms.
    // So: at least 4 blocks per CU when the problem is large enough, else the cheapest split with a lone block priced
    // at its measured ~0.6 efficiency.
    const int64_t slots = mfma ? 256 : 2048;   // CUs resp. resident workgroups (8 per CU)
    int64_t tiles_per = 1, best_cost = -1;
    for (int pass = 0; pass < 2 && best_cost < 0; ++pass) {
        for (int64_t tp = 1; tp <= (mfma ? 256 : 4096) && tp <= etiles; ++tp) {
            const int64_t sp = (etiles + tp - 1) / tp;
            if (sp > 65535) continue;
            const int64_t blocks = qtiles * sp;
            if (mfma && pass == 0 && blocks < 4 * slots) continue;
            const int64_t rounds = (blocks + slots - 1) / slots;
            int64_t cost = rounds * (tp * 16 + (mfma ? 3 : 1));
            if (mfma && rounds == 1) cost = cost * 5 / 3;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; tiles_per = tp; }
        }
        if (!mfma) break;
    }
    if (best_cost < 0) return set_error(AMDKGE_EUNSUPPORTED, "rank_counts: entity range too large for one launch; split [ent_lo, ent_hi)");
    // distance models whose probe picked the early-exit kernel: runs of >= 4 tiles amortise its per-block thresholds, and keep the
    // blocks of the fall-back launch behind it (the plain kernel, run only if the hand-over list overflowed) few
    const int64_t mcand_e = ent_hi - ent_lo;
    const bool rot_m = mode == MODE_ROT_O || mode == MODE_ROT_S;
    bool use_early = !mfma && d_screen && g_early.on && force == 0 && v4 && (!rot_m || rot_exact) && a.sgn_scale < 0.f && n >= 64 && mcand_e >= 256 &&
                     g.U >= 64 && mcand_e < 0x7FFFFFFFll && n < 0x7FFFFFFFll && screen_bytes >= (int64_t)early_fixed_bytes(n, mcand_e) + (1 << 16);
    bool probe_measured = false;
    if (use_early)   // the probe: is the exit going to fire on these tables?  (host decision, see early_probe)
        if (int rc = early_probe(mode, d_ent, d_ent_ids, ent_lo, mcand_e, n, g, w, a.sgn_scale, d_screen, (size_t)screen_bytes, &use_early, &probe_measured, st)) return rc;
    if (use_early && tiles_per < 4) {
        // (the same rounds x length cost as above over runs of 4 .. 16 tiles: a run length whose last round is nearly full -- the
        // plain kernel, when the probe picks it, pays for an underfull last round in full: runs of 8 cost it 10 % at C2)
        int64_t best = -1, pick = etiles < 4 ? etiles : 4;
        for (int64_t tp = pick; tp <= 16 && tp <= etiles; ++tp) {
            const int64_t blocks = qtiles * ((etiles + tp - 1) / tp);
            if (blocks > 16 * slots) continue;
            const int64_t cost = ((blocks + slots - 1) / slots) * (tp * 16 + 1);
            if (best < 0 || cost < best) { best = cost; pick = tp; }
        }
        tiles_per = pick;
        while (tiles_per < etiles && (etiles + tiles_per - 1) / tiles_per > 65535) ++tiles_per;   // (the grid's y extent)
    }
    int64_t splits;
    a.ent_per_block = (int)(tiles_per * et_);
    splits = (etiles + tiles_per - 1) / tiles_per;
    const dim3 grid((unsigned)qtiles, (unsigned)splits);
    if (mfma) {
        static PerDeviceOnce attr_done;
        if (attr_done.need()) {
            hipError_t e1 = hipFuncSetAttribute((const void*)rank_count_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MFMA_LDS_BYTES);
            hipError_t e2 = hipFuncSetAttribute((const void*)rank_count_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MFMA_LDS_BYTES);
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)rank_count_mfma_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MFMA_LDS_BYTES);
            if (e1 != hipSuccess || e2 != hipSuccess) return set_error_hip(e1 != hipSuccess ? e1 : e2, "hipFuncSetAttribute(rank_count_mfma)");
            attr_done.done();
        }
        a.qtiles = (int)qtiles; a.splits = (int)splits;
        const int64_t nblk = 8 * ((qtiles + 7) / 8) * splits;
        if (nblk > 0x7FFFFFFFll) return set_error(AMDKGE_EUNSUPPORTED, "rank_counts: too many tiles for one launch; split the triples or the entity range");
        const dim3 grid1((unsigned)nblk);
        const bool pipe = force != 2;
        // ---- int8 screening pass + exact recheck (kge_rank_screen.h) when the caller supplied its workspace: same counts, bit
        //      for bit; the exact kernel below then runs only as the fall-back of an overflowing recheck list ----
        const int64_t mcand = ent_hi - ent_lo;
        if (d_screen && force == 0 && v4 && pipe && g.eplane == 0 && g.U <= 2048 && n >= 128 && mcand >= 512 && sgn_scale_positive(mc) &&
            screen_bytes >= (int64_t)screen_fixed_bytes(n, mcand, g.U) + (1 << 16)) {
            if (int rc = run_screen(m, d_ent, d_ent_ids, ent_lo, mcand, n, g, w, mc, d_counts, d_screen, (size_t)screen_bytes, st)) return rc;
            a.guard = carve_screen(d_screen, (size_t)screen_bytes, n, mcand, g.U).counter + 1;
        }
        if (v4 && pipe) hipLaunchKernelGGL(rank_count_mfma_pipe_kernel, grid1, dim3(256), MFMA_LDS_BYTES, st, a);
        else if (v4) hipLaunchKernelGGL((rank_count_mfma_kernel<true>), grid1, dim3(256), MFMA_LDS_BYTES, st, a);
        else hipLaunchKernelGGL((rank_count_mfma_kernel<false>), grid1, dim3(256), MFMA_LDS_BYTES, st, a);
        return check_launch("rank_counts_mfma");
    }
    // ---- distance models: the exact early exit (kge_rank_early.h) when the caller supplied its workspace: same counts, bit for
    //      bit; the plain kernel below then runs only as the fall-back of an overflowing list ----
    {
        if (use_early) {
            if (int rc = run_early(mode, m, d_ent, d_ent_ids, ent_lo, mcand_e, n, g, w, a, grid, d_screen, (size_t)screen_bytes, &a.guard, probe_measured, st)) return rc;
            a.guard_mode = GUARD_FLAG;   // the plain kernel below: only if the list overflowed
        }
    }
    if (rot_exact) {
        if (!v4) return set_error(AMDKGE_EUNSUPPORTED, "rank_counts: RotatE's exact mode needs the padded stored layout (k_pad = amdkge_padded_k(k)); dense rows with k % 4 != 0 only have the fast mode (amdkge_set_rank_rotate_fast)");
        if (mode == MODE_ROT_S) hipLaunchKernelGGL((rank_rot_kernel<true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((rank_rot_kernel<false, false>), grid, dim3(256), 0, st, a);
        return check_launch("rank_counts_rot");
    }
#define KGE_CNT(MODE) do { if (v4) hipLaunchKernelGGL((rank_count_kernel<MODE, true>), grid, dim3(256), 0, st, a); \
                           else hipLaunchKernelGGL((rank_count_kernel<MODE, false>), grid, dim3(256), 0, st, a); } while (0)
    switch (mode) {
        case MODE_DOT: KGE_CNT(MODE_DOT); break;
        case MODE_L1: KGE_CNT(MODE_L1); break;
        case MODE_L1_SUB: KGE_CNT(MODE_L1_SUB); break;
        case MODE_ROT_O: KGE_CNT(MODE_ROT_O); break;
        default: KGE_CNT(MODE_ROT_S); break;
    }
#undef KGE_CNT
    return check_launch("rank_counts");
}

// filter lookup: one thread per test triple, lower_bound in the sorted (p,o) / (s,p) keys
__global__ void filter_ranges_kernel(const int64_t* keys, const int64_t* start, int64_t n_keys, const int32_t* triples, int64_t n,
                                     int side, int64_t n_ents, int64_t n_rels, int64_t* lo_out, int64_t* hi_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t s = triples[3 * i], p = triples[3 * i + 1], o = triples[3 * i + 2];
    const int64_t q = (side == AMDKGE_SIDE_S) ? p * n_ents + o : s * n_rels + p;
    int64_t a = 0, b = n_keys;
    while (a < b) {
        const int64_t m = (a + b) >> 1;
        if (keys[m] < q) a = m + 1; else b = m;
    }
    const bool hit = a < n_keys && keys[a] == q;
    lo_out[i] = hit ? start[a] : 0;
    hi_out[i] = hit ? start[a + 1] : 0;
}

extern "C" int amdkge_filter_ranges(const int64_t* d_keys, const int64_t* d_start, int64_t n_keys, const int32_t* d_triples,
                                    int64_t n, int32_t side, int64_t n_ents, int64_t n_rels, int64_t* d_lo, int64_t* d_hi,
                                    void* stream) {
    if (side != AMDKGE_SIDE_S && side != AMDKGE_SIDE_O) return set_error(AMDKGE_EINVAL, "filter_ranges: side must be AMDKGE_SIDE_S or AMDKGE_SIDE_O");
    if (n < 0 || n_keys < 0 || n_ents <= 0 || n_rels <= 0) return set_error(AMDKGE_EINVAL, "filter_ranges: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_triples || !d_lo || !d_hi || (n_keys > 0 && (!d_keys || !d_start))) return set_error(AMDKGE_EINVAL, "filter_ranges: NULL pointer");
    hipLaunchKernelGGL(filter_ranges_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_keys, d_start,
                       n_keys, d_triples, n, (int)side, n_ents, n_rels, d_lo, d_hi);
    return check_launch("filter_ranges");
}

extern "C" int amdkge_rank_filter(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                                  int64_t n, int32_t side, const int64_t* d_flt_lo, const int64_t* d_flt_hi,
                                  const int32_t* d_flt_ids, const int32_t* d_subset_pos, int64_t ent_lo, int64_t ent_hi,
                                  int32_t* d_sub, void* d_work, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (side != AMDKGE_SIDE_S && side != AMDKGE_SIDE_O) return set_error(AMDKGE_EINVAL, "rank_filter: side must be AMDKGE_SIDE_S or AMDKGE_SIDE_O");
    if (n < 0 || ent_lo < 0 || ent_hi < ent_lo) return set_error(AMDKGE_EINVAL, "rank_filter: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_ent || !d_rel || !d_triples || !d_flt_lo || !d_flt_hi || !d_sub || !d_work) return set_error(AMDKGE_EINVAL, "rank_filter: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RankGeom g = geom_of(m, side);
    const Workspace w = carve(d_work, m, n);
    if (int rc = run_prep(m, d_ent, d_rel, d_triples, n, side, g, w, st)) return rc;
    const ModelConst mc = model_const(m);
    FilterArgs a{};
    a.ent = d_ent; a.Q = w.Q; a.qpos = w.qpos; a.flt_lo = d_flt_lo; a.flt_hi = d_flt_hi; a.flt_ids = d_flt_ids;
    a.subset_pos = d_subset_pos; a.sub = d_sub; a.n = n; a.ent_lo = ent_lo; a.ent_hi = ent_hi; a.g = g;
    a.sgn_scale = mc.score_sign * mc.score_scale;
    const unsigned grid = (unsigned)((n + 3) / 4);
    const int mode = mode_of(m->scoring_type, side);
    const bool rot_exact = (mode == MODE_ROT_O || mode == MODE_ROT_S) && !g_rotate_fast;
    const bool v4 = (rot_exact || g.U % 4 == 0) && (g.eplane % 4 == 0) && (g.K % 4 == 0);
    if (rot_exact) {
        if (!v4) return set_error(AMDKGE_EUNSUPPORTED, "rank_filter: RotatE's exact mode needs the padded stored layout (k_pad = amdkge_padded_k(k))");
        if (mode == MODE_ROT_S) hipLaunchKernelGGL((rank_filter_kernel<MODE_ROT_S, true, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((rank_filter_kernel<MODE_ROT_O, true, true>), dim3(grid), dim3(256), 0, st, a);
        return check_launch("rank_filter_rot");
    }
    if (mode == MODE_DOT && v4 && g_rank_kernel == 0) {   // (a forced count kernel, amdkge_set_rank_kernel, also keeps round 2's filter pass)
        // contraction models: flat pair list + the coalesced exact-chain kernel; the one-wave-per-query kernel behind it runs
        // only if the list overflowed (device-side flag, no host round trip)
        if (hipError_t e = hipMemsetAsync(w.flt_counter, 0, 8, st)) return set_error_hip(e, "hipMemsetAsync(filter pair counter)");
        hipLaunchKernelGGL(filter_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, w.flt_pairs, w.flt_counter, w.flt_cap);
        if (int rc = check_launch("filter_pairs")) return rc;
        RecheckArgs ra{};
        ra.ent = d_ent; ra.Q = w.Q; ra.qpos = w.qpos; ra.ent_ids = nullptr; ra.ent_lo = 0; ra.U = g.U; ra.K = g.K; ra.QW = g.QW;
        ra.sgn_scale = a.sgn_scale;
        ra.b.counter = w.flt_counter; ra.b.pairs = w.flt_pairs; ra.b.cap = w.flt_cap; ra.b.counts = d_sub;
        static PerDeviceOnce flt_attr;
        if (flt_attr.need()) {
            if (hipError_t e = hipFuncSetAttribute((const void*)rank_recheck_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCK_LDS_BYTES))
                return set_error_hip(e, "hipFuncSetAttribute(rank_recheck<filter>)");
            flt_attr.done();
        }
        const int64_t groups = (w.flt_cap + 63) / 64;
        hipLaunchKernelGGL(rank_recheck_kernel<true>, dim3((unsigned)(groups / 4 < 1024 ? (groups + 3) / 4 : 1024)), dim3(256), RCK_LDS_BYTES, st, ra);
        if (int rc = check_launch("rank_filter_pairs")) return rc;
        a.guard = w.flt_counter + 1;
    }
#define KGE_FLT(MODE) do { if (v4) hipLaunchKernelGGL((rank_filter_kernel<MODE, true>), dim3(grid), dim3(256), 0, st, a); \
                           else hipLaunchKernelGGL((rank_filter_kernel<MODE, false>), dim3(grid), dim3(256), 0, st, a); } while (0)
    switch (mode) {
        case MODE_DOT: KGE_FLT(MODE_DOT); break;
        case MODE_L1: KGE_FLT(MODE_L1); break;
        case MODE_L1_SUB: KGE_FLT(MODE_L1_SUB); break;
        case MODE_ROT_O: KGE_FLT(MODE_ROT_O); break;
        default: KGE_FLT(MODE_ROT_S); break;
    }
#undef KGE_FLT
    return check_launch("rank_filter");
}

extern "C" int amdkge_rank_compose(const int32_t* d_counts, const int32_t* d_sub, int64_t n, int32_t strategy,
                                   int32_t* d_ranks, int64_t rank_stride, void* stream) {
    if (n < 0 || rank_stride < 1) return set_error(AMDKGE_EINVAL, "rank_compose: bad sizes");
    if (strategy < AMDKGE_RANK_WORST || strategy > AMDKGE_RANK_MIDDLE) return set_error(AMDKGE_EINVAL, "rank_compose: unknown ranking strategy");
    if (n == 0) return AMDKGE_OK;
    if (!d_counts || !d_ranks) return set_error(AMDKGE_EINVAL, "rank_compose: NULL pointer");
    hipLaunchKernelGGL(rank_compose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_counts, d_sub, n, strategy, d_ranks, rank_stride);
    return check_launch("rank_compose");
}

// ------------------------------------------------------------------------------------------------
// Discovery: the un-quantised corruption scores themselves, for a bounded chunk of queries, through the
// SAME prep + tile kernels (same rounding points and accumulation chain as the ranks).  Callers stream chunks through
// amdkge_topk_rows (kge_discovery.hip), so the reference's (n, m) score matrix never exists for more than a chunk.
// ------------------------------------------------------------------------------------------------
static int launch_store(int mode, bool v4, CountArgs& a, int64_t n, int64_t m, hipStream_t st) {
    const int64_t qtiles = (n + QT - 1) / QT, etiles = (m + ET - 1) / ET;
    int64_t tiles_per = (etiles * qtiles + 4095) / 4096;   // ~4096 blocks: enough to fill the chip, few enough to amortise the Q reloads
    if (tiles_per < 1) tiles_per = 1;
    const int64_t splits = (etiles + tiles_per - 1) / tiles_per;
    if (splits > 65535 || qtiles > 0x7FFFFFFFll) return set_error(AMDKGE_EUNSUPPORTED, "scores: too many tiles for one launch; split the queries");
    a.ent_per_block = (int)(tiles_per * ET);
    const dim3 grid((unsigned)qtiles, (unsigned)splits);
    if ((mode == MODE_ROT_O || mode == MODE_ROT_S) && !g_rotate_fast) {
        if (!v4) return set_error(AMDKGE_EUNSUPPORTED, "scores: RotatE's exact mode needs the padded stored layout (k_pad = amdkge_padded_k(k))");
        if (mode == MODE_ROT_S) hipLaunchKernelGGL((rank_rot_kernel<true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((rank_rot_kernel<false, true>), grid, dim3(256), 0, st, a);
        return check_launch("corruption_scores_rot");
    }
#define KGE_STORE(MODE) do { if (v4) hipLaunchKernelGGL((rank_count_kernel<MODE, true, true>), grid, dim3(256), 0, st, a); \
                             else hipLaunchKernelGGL((rank_count_kernel<MODE, false, true>), grid, dim3(256), 0, st, a); } while (0)
    switch (mode) {
        case MODE_DOT: KGE_STORE(MODE_DOT); break;
        case MODE_L1: KGE_STORE(MODE_L1); break;
        case MODE_L1_SUB: KGE_STORE(MODE_L1_SUB); break;
        case MODE_ROT_O: KGE_STORE(MODE_ROT_O); break;
        default: KGE_STORE(MODE_ROT_S); break;
    }
#undef KGE_STORE
    return check_launch("corruption_scores");
}

extern "C" int amdkge_corruption_scores(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                                        int64_t n, int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                                        float* d_scores, int64_t ld, void* d_work, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (side != AMDKGE_SIDE_S && side != AMDKGE_SIDE_O) return set_error(AMDKGE_EINVAL, "corruption_scores: side must be AMDKGE_SIDE_S or AMDKGE_SIDE_O");
    if (n < 0 || ent_lo < 0 || ent_hi < ent_lo || ld < ent_hi - ent_lo) return set_error(AMDKGE_EINVAL, "corruption_scores: bad sizes");
    if (!d_ent_ids && ent_hi > m->n_ents) return set_error(AMDKGE_EINVAL, "corruption_scores: entity range outside the table");
    if (n == 0 || ent_hi == ent_lo) return AMDKGE_OK;
    if (!d_ent || !d_rel || !d_triples || !d_scores || !d_work) return set_error(AMDKGE_EINVAL, "corruption_scores: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const RankGeom g = geom_of(m, side);
    const Workspace w = carve(d_work, m, n);
    if (int rc = run_prep(m, d_ent, d_rel, d_triples, n, side, g, w, st)) return rc;
    const ModelConst mc = model_const(m);
    CountArgs a{};
    a.ent = d_ent; a.Q = w.Q; a.qpos = w.qpos; a.ent_ids = d_ent_ids; a.n = n; a.ent_lo = ent_lo; a.ent_hi = ent_hi; a.g = g;
    a.sgn_scale = mc.score_sign * mc.score_scale; a.scores = d_scores; a.ld = ld;
    const int mode = mode_of(m->scoring_type, side);
    const bool rot_exact = (mode == MODE_ROT_O || mode == MODE_ROT_S) && !g_rotate_fast;
    const bool v4 = (rot_exact || g.U % 4 == 0) && (g.eplane % 4 == 0) && (g.K % 4 == 0);
    return launch_store(mode, v4, a, n, ent_hi - ent_lo, st);
}

extern "C" int amdkge_row_dots(const float* d_q, int64_t n, const float* d_table, int32_t row_floats, const int32_t* d_ent_ids,
                               int64_t ent_lo, int64_t ent_hi, float* d_out, int64_t ld, void* stream) {
    if (n < 0 || row_floats < 1 || ent_lo < 0 || ent_hi < ent_lo || ld < ent_hi - ent_lo) return set_error(AMDKGE_EINVAL, "row_dots: bad sizes");
    if (n == 0 || ent_hi == ent_lo) return AMDKGE_OK;
    if (!d_q || !d_table || !d_out) return set_error(AMDKGE_EINVAL, "row_dots: NULL pointer");
    CountArgs a{};
    a.ent = d_table; a.Q = d_q; a.qpos = nullptr; a.ent_ids = d_ent_ids; a.n = n; a.ent_lo = ent_lo; a.ent_hi = ent_hi;
    a.g = RankGeom{row_floats, 0, 0, row_floats, row_floats, 1.f};
    a.sgn_scale = 1.f; a.scores = d_out; a.ld = ld;
    const bool v4 = row_floats % 4 == 0 && (((uintptr_t)d_q | (uintptr_t)d_table) & 15) == 0;
    return launch_store(MODE_DOT, v4, a, n, ent_hi - ent_lo, (hipStream_t)stream);
}
