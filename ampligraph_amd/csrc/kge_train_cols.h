// COLUMN-SHARDED train step: every GPU holds K / W units of EVERY row -- for the complex models
// the re and im slices of the same units, stored as the row of a model with k' = k / W units -- and processes ALL positives of the
// global batch on its slice.  All five scores are sums over units (TransE.py:51-53, DistMult.py:48, ComplEx.py:58-62, HolE.py:45,
// RotatE.py:100-104), so the only thing a rank lacks is the other ranks' partial sums:
//   A. cols_scores_kernel : partial score sums of the positives and of their eta corruptions (the same Philox draws as one GPU),
//                           B (1 + eta) floats -- the ONLY exchange of the step: one all-reduce (ncclAllReduce over xGMI;
//                           6.7 MB at B = 80 000, eta = 20, whatever the table size), issued by the caller between A and B;
//   B. cols_loss_kernel   : Loss.__call__ on the complete scores (loss_functions.py:185-225): loss value, dL/dscore in place;
//   C. cols_stage_kernel  : the backward pass on the slice with the coefficients GIVEN -- d score / d row[cols] needs nothing remote:
//                           gradient rows of the positive's own s / o, side rows, one bucket entry per replaced row: exactly what the
//                           forward kernel of the owner-computes pair (kge_train_kernel.h, STAGE) hands to tile_backward_kernel,
//                           which then runs UNCHANGED on the slice table (accumulate in LDS, optimizer, regulariser -- all
//                           element-wise, hence local).
// What replaces the reference: ScoringBasedEmbeddingModel.train_step (ScoringBasedEmbeddingModel.py:370-429) on one global batch;
// the reference has no multi-device path.  Geometry: the slices are NARROW by construction (C2 at W = 8: 50 units = 13 quads per
// half), so a positive is handled by a GROUP of G = 16 / 32 / 64 lanes (one quad of each half per lane), four / two / one positives
// per wave -- a whole wave per 13-quad row would idle 80 % of its lanes.  The slice table (C2 at W = 8: 14 505 x 416 B = 6 MB) is
// L2-resident on every XCD, which is where the design gets its speed from (scripts/xcd_slice_bench.hip: 15.1 TB/s of gathers).
// Included by kge_train_tiled.hip (shares TrainArgs / StageEntry / tile_of_row with the forward kernel).
#pragma once

namespace kge {

template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G == 64) return wave_sum(v);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // (stays inside the aligned group of G lanes)
    return v;
}

__device__ __forceinline__ void wave_lds_sync() {   // LDS operations of one wave complete in order: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct ColsArgs {
    TrainArgs t;          // tables, triples, sampling, geometry, staging outputs (see kge_train_kernel.h)
    float* scores;        // A: out, partial sums [B] positives then [eta][B] corruptions; C: in, dL/dscore in the same layout
};

constexpr int COLS_PF = 4;   // replacement rows in flight per group

template <int MODEL, int G>
__device__ __forceinline__ void cols_load_spo(const TrainArgs& a, int ps, int pp, int po, int qoff, float (&s)[4][ModelTraits<MODEL>::NC],
                                              float (&p)[4][ModelTraits<MODEL>::NC], float (&o)[4][ModelTraits<MODEL>::NC]) {
    constexpr int NC = ModelTraits<MODEL>::NC;
#pragma unroll
    for (int h = 0; h < NC; ++h) {
        const fvec<4> vs = ldg<4>(a.ent + (int64_t)ps * a.K + qoff + h * a.k);
        const fvec<4> vp = ldg<4>(a.rel + (int64_t)pp * a.K + qoff + h * a.k);
        const fvec<4> vo = ldg<4>(a.ent + (int64_t)po * a.K + qoff + h * a.k);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[u][h] = vs.v[u]; p[u][h] = vp.v[u]; o[u][h] = vo.v[u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) prep_rel<MODEL>(a.mc, p[u]);
}

// the corruption draws of one positive into the group's LDS arrays (CorruptionGenerationLayerTrain.py:35-94; the same Philox rows
// as one GPU: keyed by the global corruption row)
template <int G>
__device__ __forceinline__ void cols_draws(const TrainArgs& a, int64_t i, int ps, int gl, int* sh_keep, int* sh_repl) {
    for (int j = gl; j < a.eta; j += G) {
        int keep, repl;
        if (a.neg_override) {
            const int64_t r = (int64_t)j * a.B + i;
            const int ns = a.neg_override[3 * r + 0], no = a.neg_override[3 * r + 2];
            keep = (ns == ps) ? 1 : 0;
            repl = keep ? no : ns;
        } else {
            draw_corruption(a.sc, i, j, keep, repl);
        }
        sh_keep[j] = keep;
        sh_repl[j] = repl;
    }
}

// ---- A: partial scores ---------------------------------------------------------------------------------------------------------
template <int MODEL, int G>
__global__ __launch_bounds__(256) void cols_scores_kernel(ColsArgs ca) {
    const TrainArgs& a = ca.t;
    constexpr int NC = ModelTraits<MODEL>::NC;
    constexpr int GPB = 256 / G;   // positives per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, grp = tid / G, gl = tid % G;
    const int64_t i_raw = (int64_t)blockIdx.x * GPB + grp;
    const bool active = i_raw < a.B;
    const int64_t i = active ? i_raw : a.B - 1;
    int* sh_keep = reinterpret_cast<int*>(smem) + (size_t)grp * 2 * a.eta;
    int* sh_repl = sh_keep + a.eta;
    const int ps = a.triples[3 * i + 0], pp = a.triples[3 * i + 1], po = a.triples[3 * i + 2];
    cols_draws<G>(a, i, ps, gl, sh_keep, sh_repl);
    const bool qok = gl < a.nq;
    const int qoff = (qok ? gl : 0) * 4;
    float s[4][NC], p[4][NC], o[4][NC];
    cols_load_spo<MODEL, G>(a, ps, pp, po, qoff, s, p, o);
    wave_lds_sync();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += score_unit<MODEL>(s[u], p[u], o[u]);
    const float P = group_sum<G>(qok ? acc : 0.f);
    if (active && gl == 0) ca.scores[i] = P;
    float* neg = ca.scores + a.B;
    for (int j0 = 0; j0 < a.eta; j0 += COLS_PF) {
        float e[COLS_PF][4][NC];
        int keepv[COLS_PF];
#pragma unroll
        for (int f = 0; f < COLS_PF; ++f) {
            const int j = min(j0 + f, a.eta - 1);   // past the end: the last row again, result unused
            keepv[f] = sh_keep[j];
            const float* re = a.ent + (int64_t)sh_repl[j] * a.K + qoff;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<4> ve = ldg<4>(re + h * a.k);
#pragma unroll
                for (int u = 0; u < 4; ++u) e[f][u][h] = ve.v[u];
            }
        }
#pragma unroll
        for (int f = 0; f < COLS_PF; ++f) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) t += keepv[f] ? score_unit<MODEL>(s[u], p[u], e[f][u]) : score_unit<MODEL>(e[f][u], p[u], o[u]);
            const float n = group_sum<G>(qok ? t : 0.f);
            if (active && gl == 0 && j0 + f < a.eta) neg[(int64_t)(j0 + f) * a.B + i] = n;
        }
    }
}

// ---- B: loss on the complete scores ----------------------------------------------------------------------------------------------
// One THREAD per positive: its 1 + eta scores sit B floats apart (layout j * B + i), so the lanes of a wave read and write
// consecutive floats, and every lane evaluates a whole Loss.__call__ (loss_functions.py:285-308,359-382,441-464,539-574,629-654; the
// arithmetic of loss_and_dscore, kge_train_kernel.h, with the sums taken serially) -- the first version gave a positive a whole wave
// and used 21 of its 64 lanes: 82 us at B = 80 000, eta = 20 (profiles/r05c_cols8_kernel_stats.csv) for 6.7 MB of data.
// scores <- dL/dscore in place; the loss value into ONE atomic per block.
__global__ __launch_bounds__(256) void cols_loss_kernel(float* __restrict__ scores, int64_t B, int eta, amdkge_loss L, float sgn_scale, double* loss_sum) {
    __shared__ double s_loss[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* neg = scores + B;
    const float feta = (float)eta;
    double tot = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < B; i += (int64_t)gridDim.x * 256) {
        const float P = sgn_scale * scores[i];   // the reference's rounding: reduce_sum, then negate (TransE / RotatE) or scale (HolE)
        float* nj = neg + i;                      // corruption j at nj[j * B]
        float red = L.reduction_mean ? feta : 1.f, per, dP;
        switch (L.kind) {
            case AMDKGE_LOSS_PAIRWISE: {
                float acc = 0.f, cnt = 0.f;
                for (int j = 0; j < eta; ++j) {
                    const float h = L.margin - P + sgn_scale * nj[(int64_t)j * B];
                    const bool act = h >= 0.f;
                    acc += hinge_nan(h);
                    cnt += act ? 1.f : 0.f;
                    nj[(int64_t)j * B] = act ? 1.f / red : masked_zero(h);
                }
                per = acc / red;
                dP = -cnt / red;
            } break;
            case AMDKGE_LOSS_NLL: {
                if (L.reduction_mean) red = 2.f * feta;
                const bool inP = (P >= -75.f) && (P <= 75.f);
                const float Pc = clip_exp(P);
                float acc = 0.f;
                for (int j = 0; j < eta; ++j) {
                    const float n = sgn_scale * nj[(int64_t)j * B];
                    const bool in = (n >= -75.f) && (n <= 75.f);
                    const float nc = clip_exp(n);
                    acc += logf(1.f + expf(nc));
                    nj[(int64_t)j * B] = in ? sigmoidf(nc) / red : masked_zero(n);
                }
                per = (feta * logf(1.f + expf(-Pc)) + acc) / red;
                dP = inP ? -feta * sigmoidf(-Pc) / red : 0.f;
            } break;
            case AMDKGE_LOSS_ABSOLUTE_MARGIN: {
                float acc = 0.f;
                for (int j = 0; j < eta; ++j) {
                    const float h = L.margin + sgn_scale * nj[(int64_t)j * B];
                    acc += hinge_nan(h);
                    nj[(int64_t)j * B] = (h >= 0.f) ? 1.f / red : masked_zero(h);
                }
                per = (acc - feta * P) / red;
                dP = -feta / red;
            } break;
            case AMDKGE_LOSS_SELF_ADVERSARIAL: {
                float mx = -INFINITY;
                for (int j = 0; j < eta; ++j) mx = fmaxf(mx, L.alpha * (sgn_scale * nj[(int64_t)j * B]));
                float se = 0.f;
                for (int j = 0; j < eta; ++j) se += expf(L.alpha * (sgn_scale * nj[(int64_t)j * B]) - mx);
                float lbar = 0.f;
                for (int j = 0; j < eta; ++j) {
                    const float n = sgn_scale * nj[(int64_t)j * B];
                    lbar += expf(L.alpha * n - mx) / se * log_sigmoid(-n - L.margin);
                }
                for (int j = 0; j < eta; ++j) {
                    const float n = sgn_scale * nj[(int64_t)j * B];
                    const float w = expf(L.alpha * n - mx) / se;
                    const float ell = log_sigmoid(-n - L.margin);
                    nj[(int64_t)j * B] = computed_zero((w * sigmoidf(n + L.margin) - L.alpha * w * (ell - lbar)) / red, n);
                }
                per = -log_sigmoid(L.margin + P) - lbar / red;
                dP = -sigmoidf(-(L.margin + P));
            } break;
            default: {   // AMDKGE_LOSS_MULTICLASS_NLL
                const bool inP = (P >= -75.f) && (P <= 75.f);
                const float eP = expf(clip_exp(P));
                float acc = 0.f;
                for (int j = 0; j < eta; ++j) acc += expf(clip_exp(sgn_scale * nj[(int64_t)j * B]));
                const float Z = acc / red + eP;
                for (int j = 0; j < eta; ++j) {
                    const float n = sgn_scale * nj[(int64_t)j * B];
                    const bool in = (n >= -75.f) && (n <= 75.f);
                    nj[(int64_t)j * B] = in ? expf(clip_exp(n)) / Z / red : masked_zero(n);
                }
                per = -logf(eP / Z);
                dP = inP ? -1.f + eP / Z : 0.f;
            } break;
        }
        scores[i] = dP;
        tot += (double)per;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if (lane == 0) s_loss[wv] = tot;
    __syncthreads();
    if (tid == 0 && loss_sum) {
        const double t = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        if (t != 0.0) atomicAdd(loss_sum, t);
    }
}

// ---- C: backward on the slice with given coefficients, staged for tile_backward_kernel ---------------------------------------------
template <int MODEL, int G>
__global__ __launch_bounds__(256) void cols_stage_kernel(ColsArgs ca) {
    const TrainArgs& a = ca.t;
    constexpr int NC = ModelTraits<MODEL>::NC;
    constexpr int GPB = 256 / G;
    constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, grp = tid / G, gl = tid % G, lane = tid & 63;
    const int eta = a.eta;
    const int64_t i_raw = (int64_t)blockIdx.x * GPB + grp;
    const bool active = i_raw < a.B;
    const int64_t i = active ? i_raw : a.B - 1;
    int* sh_keep = reinterpret_cast<int*>(smem) + (size_t)grp * 2 * eta;
    int* sh_repl = sh_keep + eta;
    float* sh_row = reinterpret_cast<float*>(smem + (size_t)GPB * 2 * eta * 4) + (size_t)grp * a.K;   // the group's relation-gradient row, transposed
    int* sh_rel = reinterpret_cast<int*>(smem + (size_t)GPB * 2 * eta * 4 + (size_t)GPB * a.K * 4);     // [GPB] relation id (-1: inactive)
    const int ps = a.triples[3 * i + 0], pp = a.triples[3 * i + 1], po = a.triples[3 * i + 2];
    cols_draws<G>(a, i, ps, gl, sh_keep, sh_repl);
    if (gl == 0) sh_rel[grp] = active ? pp : -1;
    const bool qok = gl < a.nq;
    const int qoff = (qok ? gl : 0) * 4;
    float s[4][NC], p[4][NC], o[4][NC];
    cols_load_spo<MODEL, G>(a, ps, pp, po, qoff, s, p, o);
    float pad1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pad1[u] = (MODEL == AMDKGE_ROTATE && qoff + u >= a.k_live) ? 1.f : 0.f;
    const float sgn_scale = a.mc.score_sign * a.mc.score_scale;
    // side rows for the owner kernel: the staging protocol of train_fwdbwd_kernel<STAGE> (kge_train_kernel.h)
    if (active && qok) {
        float va[NC][4], vb[NC][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (TRILINEAR) {
                float ds[NC], dp[NC], dd[NC];
                grad_unit<MODEL>(s[u], p[u], o[u], 1.f, ds, dp, dd);
#pragma unroll
                for (int h = 0; h < NC; ++h) { va[h][u] = dd[h]; vb[h][u] = ds[h]; }
            } else if constexpr (MODEL == AMDKGE_ROTATE) {
                const float cs = p[u][0], sn = p[u][1];   // A = s o r, B = o o conj(r)
                va[0][u] = s[u][0] * cs - s[u][1] * sn; va[1][u] = s[u][0] * sn + s[u][1] * cs;
                vb[0][u] = o[u][0] * cs + o[u][1] * sn; vb[1][u] = o[u][1] * cs - o[u][0] * sn;
            } else {
#pragma unroll
                for (int h = 0; h < NC; ++h) { va[h][u] = s[u][h]; vb[h][u] = o[u][h]; }
            }
        }
        float* qa = a.stage_rows + ((int64_t)i * a.ns + 2) * a.K + qoff;
        float* qb = a.stage_rows + ((int64_t)i * a.ns + 3) * a.K + qoff;
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            *reinterpret_cast<float4*>(qa + h * a.k) = make_float4(va[h][0], va[h][1], va[h][2], va[h][3]);
            *reinterpret_cast<float4*>(qb + h * a.k) = make_float4(vb[h][0], vb[h][1], vb[h][2], vb[h][3]);
        }
    }
    wave_lds_sync();
    // ---- gradients of the positive's own rows: the positive, then every corruption, coefficients as given ----
    const float* cneg = ca.scores + a.B;
    float gs[4][NC], gp[4][NC], go[4][NC];
    {
        const float dP = ca.scores[i] * sgn_scale;
#pragma unroll
        for (int u = 0; u < 4; ++u) grad_unit<MODEL>(s[u], p[u], o[u], dP, gs[u], gp[u], go[u], pad1[u]);
    }
    for (int j0 = 0; j0 < eta; j0 += COLS_PF) {
        float e[COLS_PF][4][NC];
        int keepv[COLS_PF];
        float gv[COLS_PF];
#pragma unroll
        for (int f = 0; f < COLS_PF; ++f) {
            const int j = min(j0 + f, eta - 1);
            keepv[f] = sh_keep[j];
            gv[f] = (j0 + f < eta) ? cneg[(int64_t)j * a.B + i] * sgn_scale : 0.f;
            const float* re = a.ent + (int64_t)sh_repl[j] * a.K + qoff;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<4> ve = ldg<4>(re + h * a.k);
#pragma unroll
                for (int u = 0; u < 4; ++u) e[f][u][h] = ve.v[u];
            }
        }
#pragma unroll
        for (int f = 0; f < COLS_PF; ++f) {
            if (j0 + f >= eta) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float ds[NC], dp[NC], dd[NC];
                if (keepv[f]) {   // (s, p, e): object replaced
                    grad_unit<MODEL>(s[u], p[u], e[f][u], gv[f], ds, dp, dd, pad1[u]);
#pragma unroll
                    for (int h = 0; h < NC; ++h) { gs[u][h] += ds[h]; gp[u][h] += dp[h]; }
                } else {          // (e, p, o): subject replaced
                    grad_unit<MODEL>(e[f][u], p[u], o[u], gv[f], ds, dp, dd, pad1[u]);
#pragma unroll
                    for (int h = 0; h < NC; ++h) { go[u][h] += dd[h]; gp[u][h] += dp[h]; }
                }
            }
        }
    }
    // ---- one entry per row gradient that lands in the entity table, into the bucket of the owning tile (as the forward kernel) ----
    if (active)
        for (int j = gl; j < eta + 2; j += G) {
            uint32_t dest, role;
            float g, coeff = 1.f;
            if (j < eta) { dest = (uint32_t)sh_repl[j]; role = sh_keep[j] ? 0u : 1u; coeff = cneg[(int64_t)j * a.B + i]; g = coeff * sgn_scale; }
            else { dest = (uint32_t)(j == eta ? ps : po); role = (j == eta) ? 2u : 3u; g = 1.f; }
            if (!entry_wanted(coeff, g)) continue;   // (no entry below the smallest normal number, masked zeros of non-finite scores kept: see the forward kernel)
            uint32_t tile, local;
            tile_of_row(dest, (uint32_t)a.st_n_tiles, (uint32_t)a.st_rb, tile, local);
            StageEntry en{(uint32_t)i, role | (local << 2), g, dest};
            const int slotpos = atomicAdd(a.st_counters + (size_t)tile * 32, 1);
            if (slotpos < a.st_cap) {
                a.st_lists[(size_t)tile * a.st_cap + slotpos] = en;
            } else {
                const int op = atomicAdd(a.st_counters + (size_t)a.st_n_tiles * 32, 1);
                if (op < a.st_ovf_cap) a.st_ovf[op] = en;
            }
        }
    // ---- the positive's own s / o gradient rows: staged (roles 2, 3) ----
    if (active && qok) {
        float* ps_ = a.stage_rows + ((int64_t)i * a.ns + 0) * a.K + qoff;
        float* po_ = a.stage_rows + ((int64_t)i * a.ns + 1) * a.K + qoff;
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            *reinterpret_cast<float4*>(ps_ + h * a.k) = make_float4(gs[0][h], gs[1][h], gs[2][h], gs[3][h]);
            *reinterpret_cast<float4*>(po_ + h * a.k) = make_float4(go[0][h], go[1][h], go[2][h], go[3][h]);
        }
    }
    // ---- relation rows: few and hot -> atomic row-adds, every wave instruction on consecutive floats (rows transposed through LDS,
    //      the wave's groups taken one after the other by all 64 lanes) ----
    const float rmul = (MODEL == AMDKGE_ROTATE) ? 1.f / a.mc.phase_div : 1.f;   // RotatE: d/dtheta = d/dphi / phase_div, second half 0
    if (qok) {
#pragma unroll
        for (int h = 0; h < NC; ++h)
            *reinterpret_cast<float4*>(sh_row + qoff + h * a.k) = make_float4(gp[0][h] * rmul, gp[1][h] * rmul, gp[2][h] * rmul, gp[3][h] * rmul);
    }
    wave_lds_sync();
    const int nfloats = (MODEL == AMDKGE_ROTATE) ? a.k : a.K;
    const int g0 = (tid >> 6) * (64 / G);   // first group of this wave
#pragma unroll
    for (int q = 0; q < 64 / G; ++q) {
        const int r = sh_rel[g0 + q];
        if (r < 0) continue;
        const float* src = reinterpret_cast<const float*>(smem + (size_t)GPB * 2 * eta * 4) + (size_t)(g0 + q) * a.K;
        float* grow = a.g_rel + (int64_t)r * a.K;
        for (int idx = lane; idx < nfloats; idx += KGE_WAVE) atomic_add_f32(grow + idx, src[idx]);
    }
}

__host__ inline size_t cols_scores_lds(int G, int eta) { return (size_t)(256 / G) * 2 * eta * 4; }
__host__ inline size_t cols_stage_lds(int G, int eta, int K) { return (size_t)(256 / G) * (2 * (size_t)eta * 4 + (size_t)K * 4 + 4); }

}  // namespace kge
