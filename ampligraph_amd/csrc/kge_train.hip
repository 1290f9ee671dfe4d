// Fused training forward+backward for gfx950: lookup (a1) + negative sampling (a3) + scoring
// (a4-a8) + loss (a12-a16) + gradient scatter, one launch per batch.  Replaces the TF graph of
// ScoringBasedEmbeddingModel.train_step (/root/reference/ampligraph/latent_features/models/
// ScoringBasedEmbeddingModel.py:370-429) up to (not including) optimizer.apply_gradients.
//
// Mapping: one "slot" of W wave64s owns one positive and its eta corruptions.  Each lane owns CH
// "quads" of VEC consecutive units of the embedding row; the positive's s, p, o quads and their
// gradient accumulators stay in VGPRs for the whole slot, replacement rows are streamed (16-byte
// coalesced loads) twice: once for the scores, once (L2-hot) for the gradients.  Score reductions
// are DPP wave reductions (+ one LDS hop when W > 1); the loss and dL/dscore of all eta corruptions
// are evaluated by the wave itself, so nothing but row gradients leaves the CU.  Row gradients are
// accumulated in registers per distinct row (s, p, o: one atomic row-add each per positive; every
// replacement row: one atomic row-add) with hardware fp32 atomics into the dense gradient buffers.
#include "kge_device.h"
#include "kge_host.h"
#include <stdlib.h>

namespace kge {

struct TrainArgs {
    const float* ent;
    const float* rel;
    const int32_t* triples;
    const int32_t* neg_override;
    float* g_ent;
    float* g_rel;
    double* loss_sum;
    float* pos_scores;
    float* neg_scores;
    int64_t B;
    int eta;
    int k;       // user k (units per half for complex models)
    int K;       // floats per row
    int nq;      // quads per row ( = units / VEC )
    SampleCfg sc;
    ModelConst mc;
    amdkge_loss loss;
    int dbg;     // development ablation flags (env AMDKGE_DEBUG): 1 no neg-row atomics, 2 no s/p/o atomics, 4 no pass 2, 8 workgroup-scope atomics
};

__device__ __forceinline__ float log_sigmoid(float x) {
    // -softplus(-x), stable on both tails
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) {
    return 1.f / (1.f + expf(-x));
}

// Loss.__call__ for one positive: neg scores in `sn[0..eta)` (LDS) are replaced by dL/dneg.
// Returns per-sample loss and dL/dpos.  Executed by one whole wave (all lanes get the results).
__device__ __forceinline__ void loss_and_dscore(const amdkge_loss& L, float P, float* sn, int eta, int lane,
                                                float& per, float& dP) {
    const float feta = (float)eta;
    float red = L.reduction_mean ? feta : 1.f;
    switch (L.kind) {
        case AMDKGE_LOSS_PAIRWISE: {  // loss_functions.py:302-308
            float acc = 0.f, cnt = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float h = L.margin - P + sn[j];
                const bool act = h >= 0.f;
                acc += fmaxf(h, 0.f);
                cnt += act ? 1.f : 0.f;
                sn[j] = act ? 1.f / red : 0.f;
            }
            per = wave_sum(acc) / red;
            dP = -wave_sum(cnt) / red;
        } break;
        case AMDKGE_LOSS_NLL: {  // :376-382 (clip at :60-66)
            if (L.reduction_mean) red = 2.f * feta;
            const bool inP = (P >= -75.f) && (P <= 75.f);
            const float Pc = fminf(fmaxf(P, -75.f), 75.f);
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const bool in = (n >= -75.f) && (n <= 75.f);
                const float nc = fminf(fmaxf(n, -75.f), 75.f);
                acc += logf(1.f + expf(nc));
                sn[j] = in ? sigmoidf(nc) / red : 0.f;
            }
            per = (feta * logf(1.f + expf(-Pc)) + wave_sum(acc)) / red;
            dP = inP ? -feta * sigmoidf(-Pc) / red : 0.f;
        } break;
        case AMDKGE_LOSS_ABSOLUTE_MARGIN: {  // :458-464
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float h = L.margin + sn[j];
                acc += fmaxf(h, 0.f);
                sn[j] = (h >= 0.f) ? 1.f / red : 0.f;
            }
            per = (wave_sum(acc) - feta * P) / red;
            dP = -feta / red;
        } break;
        case AMDKGE_LOSS_SELF_ADVERSARIAL: {  // :556-574 (softmax NOT stop-gradiented)
            float mx = -INFINITY;
            for (int j = lane; j < eta; j += KGE_WAVE) mx = fmaxf(mx, L.alpha * sn[j]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) se += expf(L.alpha * sn[j] - mx);
            se = wave_sum(se);
            float lb = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float w = expf(L.alpha * sn[j] - mx) / se;
                lb += w * log_sigmoid(-sn[j] - L.margin);
            }
            const float lbar = wave_sum(lb);
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const float w = expf(L.alpha * n - mx) / se;
                const float ell = log_sigmoid(-n - L.margin);
                sn[j] = (w * sigmoidf(n + L.margin) - L.alpha * w * (ell - lbar)) / red;
            }
            per = -log_sigmoid(L.margin + P) - lbar / red;
            dP = -sigmoidf(-(L.margin + P));
        } break;
        default: {  // AMDKGE_LOSS_MULTICLASS_NLL :647-654
            const bool inP = (P >= -75.f) && (P <= 75.f);
            const float eP = expf(fminf(fmaxf(P, -75.f), 75.f));
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) acc += expf(fminf(fmaxf(sn[j], -75.f), 75.f));
            const float Z = wave_sum(acc) / red + eP;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const bool in = (n >= -75.f) && (n <= 75.f);
                sn[j] = in ? expf(fminf(fmaxf(n, -75.f), 75.f)) / Z / red : 0.f;
            }
            per = -logf(eP / Z);
            dP = inP ? -1.f + eP / Z : 0.f;
        } break;
    }
}

__host__ __device__ inline size_t slot_lds_bytes(int eta, int W) {
    // neg[eta+1], repl[eta+1], keep[eta+1], part[W][eta+1] (W>1), rounded to 8 bytes
    const size_t b = (size_t)(eta + 1) * (3 + (W > 1 ? W : 0)) * 4;
    return (b + 7) & ~(size_t)7;
}

template <int W>
__device__ __forceinline__ void slot_sync() {
    if constexpr (W == 1) {
        // single-wave slot: LDS ops of one wave complete in order; only stop compiler reordering
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// LDS layout per slot: float neg[eta+1] (index eta = the positive); int repl[eta]; int keep[eta];
// float part[W][eta+1] (W>1 only).  Tail of the block: double blockloss[SLOTS].
template <int MODEL, int VEC, int W, int CH>
__global__ __launch_bounds__(256) void train_fwdbwd_kernel(TrainArgs a) {
    using T = ModelTraits<MODEL>;
    constexpr int NC = T::NC;
    constexpr int SLOTS = 4 / W;          // positives per 256-thread block
    constexpr int TS = KGE_WAVE * W;      // threads per slot
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int slot = tid / TS;
    const int ts = tid % TS;              // thread index inside the slot
    const int lane = tid & 63;
    const int wv = ts >> 6;               // wave index inside the slot
    const int eta = a.eta;
    const int64_t i_raw = (int64_t)blockIdx.x * SLOTS + slot;
    const bool active = i_raw < a.B;
    const int64_t i = active ? i_raw : (a.B - 1);  // tail slots recompute the last positive, write nothing

    const int e1 = eta + 1;
    const size_t per_slot = slot_lds_bytes(eta, W);
    char* base = smem + (size_t)slot * per_slot;
    float* sh_neg = reinterpret_cast<float*>(base);
    int* sh_repl = reinterpret_cast<int*>(base + (size_t)e1 * 4);
    int* sh_keep = reinterpret_cast<int*>(base + (size_t)e1 * 8);
    float* sh_part = reinterpret_cast<float*>(base + (size_t)e1 * 12);
    double* sh_loss = reinterpret_cast<double*>(smem + (size_t)SLOTS * per_slot);

    const int ps = a.triples[3 * i + 0], pp = a.triples[3 * i + 1], po = a.triples[3 * i + 2];

    // ---- negatives of this positive (a3) ------------------------------------------------------
    for (int j = ts; j < eta; j += TS) {
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

    // ---- resident quads of s, p, o ------------------------------------------------------------
    const float* rs = a.ent + (int64_t)ps * a.K;
    const float* rp = a.rel + (int64_t)pp * a.K;
    const float* ro = a.ent + (int64_t)po * a.K;
    float s[CH][VEC][NC], p[CH][VEC][NC], o[CH][VEC][NC];
    bool qok[CH];
    int qoff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int q = ts + c * TS;
        qok[c] = q < a.nq;
        qoff[c] = (qok[c] ? q : 0) * VEC;
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            const fvec<VEC> vs = ldg<VEC>(rs + qoff[c] + h * a.k);
            const fvec<VEC> vp = ldg<VEC>(rp + qoff[c] + h * a.k);
            const fvec<VEC> vo = ldg<VEC>(ro + qoff[c] + h * a.k);
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                s[c][u][h] = vs.v[u]; p[c][u][h] = vp.v[u]; o[c][u][h] = vo.v[u];
            }
        }
#pragma unroll
        for (int u = 0; u < VEC; ++u) prep_rel<MODEL>(a.mc, p[c][u]);
    }
    slot_sync<W>();

    const float sgn_scale = a.mc.score_sign * a.mc.score_scale;

    // ---- pass 1: scores (positive as j == -1) -------------------------------------------------
    for (int j = -1; j < eta; ++j) {
        const int keep = (j < 0) ? 1 : sh_keep[j];
        const int64_t er = (j < 0) ? (int64_t)po : (int64_t)sh_repl[j];
        const float* re = a.ent + er * a.K;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float e[VEC][NC];
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<VEC> ve = ldg<VEC>(re + qoff[c] + h * a.k);
#pragma unroll
                for (int u = 0; u < VEC; ++u) e[u][h] = ve.v[u];
            }
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < VEC; ++u)
                acc += keep ? score_unit<MODEL>(s[c][u], p[c][u], e[u]) : score_unit<MODEL>(e[u], p[c][u], o[c][u]);
            part += qok[c] ? acc : 0.f;
        }
        const float tot = wave_sum(part);
        const int jj = (j < 0) ? eta : j;
        if constexpr (W > 1) {
            if (lane == 0) sh_part[wv * e1 + jj] = tot;   // cross-wave hop, summed after the loop
        } else {
            // reference rounding: reduce_sum, then negate (TransE/RotatE) or scale (HolE)
            if (lane == 0) sh_neg[jj] = sgn_scale * tot;
        }
    }
    if constexpr (W > 1) {
        __syncthreads();
        for (int j = ts; j < e1; j += TS) {
            float t2 = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) t2 += sh_part[w * e1 + j];
            sh_neg[j] = sgn_scale * t2;
        }
    }
    slot_sync<W>();
    const float P = sh_neg[eta];

    if (active && a.neg_scores)
        for (int j = ts; j < eta; j += TS) a.neg_scores[(int64_t)j * a.B + i] = sh_neg[j];
    if (active && a.pos_scores && ts == 0) a.pos_scores[i] = P;
    if constexpr (W > 1) __syncthreads();  // sh_neg is rewritten below by wave 0

    // ---- loss + dL/dscore (a12-a16): wave 0 of the slot, results through LDS -------------------
    float per = 0.f, dP = 0.f;
    if (W == 1 || wv == 0) loss_and_dscore(a.loss, P, sh_neg, eta, lane, per, dP);
    if constexpr (W > 1) {
        if (wv == 0 && lane == 0) sh_part[0] = dP;
        __syncthreads();
        dP = sh_part[0];
    } else {
        slot_sync<W>();
    }
    if (ts == 0) sh_loss[slot] = active ? (double)per : 0.0;

    // ---- pass 2: gradients -------------------------------------------------------------------
    float gs[CH][VEC][NC], gp[CH][VEC][NC], go[CH][VEC][NC];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            float ds[NC], dp[NC], dd[NC];
            grad_unit<MODEL>(s[c][u], p[c][u], o[c][u], dP * sgn_scale, ds, dp, dd);
#pragma unroll
            for (int h = 0; h < NC; ++h) { gs[c][u][h] = ds[h]; gp[c][u][h] = dp[h]; go[c][u][h] = dd[h]; }
        }

    // Row-gradient emit.  fp32 atomics retire per 128-byte line (measured ~10.4 G line-ops/s on MI355X,
    // scripts/atomic_bench.hip), so every atomic wave-instruction must cover 64 CONSECUTIVE floats.
    // VEC == 1: the lane's units already are consecutive across lanes.  VEC == 4 (16-byte loads): the row
    // is transposed through a per-wave LDS staging row (ds_write_b128 -> ds_read_b32) first.
    float* stage = reinterpret_cast<float*>(smem + (size_t)SLOTS * per_slot + SLOTS * sizeof(double)) + (size_t)(tid >> 6) * a.K;
    auto emit_row = [&](float* grow, const float (&gr)[CH][VEC][NC], int nfloats, float mul) {
        if constexpr (VEC == 1) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < NC; ++h)
                    if (qok[c] && qoff[c] + h * a.k < nfloats) atomic_add_f32(grow + qoff[c] + h * a.k, gr[c][0][h] * mul);
        } else {
            static_assert(VEC == 1 || W == 1, "LDS-transposed emit is per wave");
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < NC; ++h)
                    if (qok[c])
                        *reinterpret_cast<float4*>(stage + qoff[c] + h * a.k) =
                            make_float4(gr[c][0][h] * mul, gr[c][1][h] * mul, gr[c][2][h] * mul, gr[c][3][h] * mul);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int idx = lane; idx < nfloats; idx += KGE_WAVE) atomic_add_f32(grow + idx, stage[idx]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };

    const bool do_neg_atomics = active && !(a.dbg & 1);
    for (int j = 0; j < ((a.dbg & 4) ? 0 : eta); ++j) {
        const int keep = sh_keep[j];
        const int64_t er = (int64_t)sh_repl[j];
        const float g = sh_neg[j] * sgn_scale;
        const float* re = a.ent + er * a.K;
        float gr[CH][VEC][NC];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float e[VEC][NC];
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<VEC> ve = ldg<VEC>(re + qoff[c] + h * a.k);
#pragma unroll
                for (int u = 0; u < VEC; ++u) e[u][h] = ve.v[u];
            }
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                float ds[NC], dp[NC], dd[NC];
                if (keep) {   // (s, p, e): object replaced
                    grad_unit<MODEL>(s[c][u], p[c][u], e[u], g, ds, dp, dd);
#pragma unroll
                    for (int h = 0; h < NC; ++h) { gs[c][u][h] += ds[h]; gp[c][u][h] += dp[h]; gr[c][u][h] = dd[h]; }
                } else {      // (e, p, o): subject replaced
                    grad_unit<MODEL>(e[u], p[c][u], o[c][u], g, ds, dp, dd);
#pragma unroll
                    for (int h = 0; h < NC; ++h) { go[c][u][h] += dd[h]; gp[c][u][h] += dp[h]; gr[c][u][h] = ds[h]; }
                }
            }
        }
        if (do_neg_atomics) emit_row(a.g_ent + er * a.K, gr, a.K, 1.f);
    }

    // ---- per-block loss: one fp64 atomic -------------------------------------------------------
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) t += sh_loss[q];
        atomicAdd(a.loss_sum, t);
    }

    // ---- one atomic row-add per resident row ---------------------------------------------------
    if (active && !(a.dbg & 2)) {
        emit_row(a.g_ent + (int64_t)ps * a.K, gs, a.K, 1.f);
        emit_row(a.g_ent + (int64_t)po * a.K, go, a.K, 1.f);
        if constexpr (MODEL == AMDKGE_ROTATE) {
            // d/dtheta = d/dphi / phase_div ; the second half of the relation row gets no gradient
            emit_row(a.g_rel + (int64_t)pp * a.K, gp, a.k, 1.f / a.mc.phase_div);
        } else {
            emit_row(a.g_rel + (int64_t)pp * a.K, gp, a.K, 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stand-alone sampler (materialises what the fused kernel draws; a3)
// ------------------------------------------------------------------------------------------------
__global__ void sample_kernel(const int32_t* triples, int64_t B, int eta, SampleCfg sc, int32_t* out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * eta) return;
    const int j = (int)(r / B);
    const int64_t i = r % B;
    int keep, repl;
    draw_corruption(sc, i, j, keep, repl);
    const int s = triples[3 * i], p = triples[3 * i + 1], o = triples[3 * i + 2];
    out[3 * r + 0] = keep ? s : repl;
    out[3 * r + 1] = p;
    out[3 * r + 2] = keep ? repl : o;
}

template <int MODEL, int VEC, int W>
static int launch_train_w(const TrainArgs& a, int CH, hipStream_t st) {
    const int slots = 4 / W;
    const size_t shmem = (size_t)slots * slot_lds_bytes(a.eta, W) + slots * sizeof(double) + (VEC == 4 ? 4 * (size_t)a.K * 4 : 0);
    if (shmem > 64 * 1024) return set_error(AMDKGE_EUNSUPPORTED, "train: eta too large for the LDS score buffer");
    const unsigned grid = (unsigned)((a.B + slots - 1) / slots);
#define KGE_LAUNCH(CC) hipLaunchKernelGGL((train_fwdbwd_kernel<MODEL, VEC, W, CC>), dim3(grid), dim3(256), shmem, st, a)
    switch (CH) {
        case 1: KGE_LAUNCH(1); break;
        case 2: KGE_LAUNCH(2); break;
        case 4: KGE_LAUNCH(4); break;
        default: KGE_LAUNCH(8); break;
    }
#undef KGE_LAUNCH
    return check_launch("train_fwdbwd");
}

// Slot geometry.  A lane owns CH "quads" of VEC consecutive units at quad index ts + c*64*W, so with
// VEC == 1 every wave-instruction (load or atomic) covers 64 consecutive floats = two full 128-byte
// lines.  Measured on MI355X (scripts/atomic_bench.hip): fp32 atomics retire ~10.4 G cache-line
// operations/s whatever the number of dwords per line, so the 16-byte-per-lane layout (8 active
// dwords per line) makes the gradient scatter 4.4x slower than the lane-contiguous one.
template <int MODEL, int VEC>
static int launch_train_mv(const TrainArgs& a, hipStream_t st) {
    int W = 1;
    while (W < 4 && (a.nq + 64 * W - 1) / (64 * W) > 8) W *= 2;
    int ch = (a.nq + 64 * W - 1) / (64 * W);
    if (ch > 8) return set_error(AMDKGE_EUNSUPPORTED, "train: embedding row too long for the compiled slot geometries (k > 2048)");
    int CH = 1;
    while (CH < ch) CH *= 2;
    if (W == 1) return launch_train_w<MODEL, VEC, 1>(a, CH, st);
    if (W == 2) return launch_train_w<MODEL, VEC, 2>(a, CH, st);
    return launch_train_w<MODEL, VEC, 4>(a, CH, st);
}

template <int MODEL>
static int launch_train_m(TrainArgs& a, hipStream_t st) {
    const int units = a.k;  // units per row: k floats (real models) or k complex pairs
    // 16-byte loads + LDS-transposed scatter when one wave covers the row with <= 2 quads per lane
    if (units % 4 == 0 && units <= 512 && !(a.dbg & 16)) { a.nq = units / 4; return launch_train_w<MODEL, 4, 1>(a, a.nq <= 64 ? 1 : 2, st); }
    a.nq = units;
    return launch_train_mv<MODEL, 1>(a, st);
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_sample_corruptions(const int32_t* d_triples, int64_t B, int32_t eta, int64_t sample_base,
                                         int64_t sample_range, uint64_t seed, uint64_t step, int64_t row_offset,
                                         int64_t b_global, int32_t* d_out, void* stream) {
    if (B < 0 || eta < 0 || !d_out || (B > 0 && !d_triples)) return set_error(AMDKGE_EINVAL, "sample: bad arguments");
    if (sample_range <= 0 || sample_range > 0xFFFFFFFFll) return set_error(AMDKGE_EINVAL, "sample: sample_range must be in [1, 2^32)");
    if (B == 0 || eta == 0) return AMDKGE_OK;
    SampleCfg sc{sample_base, (uint32_t)sample_range, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                 (uint32_t)(step >> 32), row_offset, b_global > 0 ? b_global : B};
    const int64_t n = B * eta;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_triples, B, eta, sc, d_out);
    return check_launch("sample_corruptions");
}

extern "C" int amdkge_train_fwdbwd(const amdkge_model* m, const amdkge_loss* loss, const float* d_ent, const float* d_rel,
                                   const int32_t* d_triples, int64_t B, int32_t eta, int64_t sample_base,
                                   int64_t sample_range, uint64_t seed, uint64_t step, int64_t row_offset,
                                   int64_t b_global, const int32_t* d_neg_override, float* d_grad_ent,
                                   float* d_grad_rel, double* d_loss_sum, float* d_pos_scores, float* d_neg_scores,
                                   void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (!loss || loss->kind < 0 || loss->kind > AMDKGE_LOSS_MULTICLASS_NLL) return set_error(AMDKGE_EINVAL, "train: unknown loss kind");
    if (!d_ent || !d_rel || !d_grad_ent || !d_grad_rel || !d_loss_sum) return set_error(AMDKGE_EINVAL, "train: null table / gradient / loss pointer");
    if (B < 0 || eta < 1) return set_error(AMDKGE_EINVAL, "train: B must be >= 0 and eta >= 1");
    if (B == 0) return AMDKGE_OK;
    if (!d_triples) return set_error(AMDKGE_EINVAL, "train: null triples");
    if (!d_neg_override && (sample_range <= 0 || sample_range > 0xFFFFFFFFll || sample_base < 0 ||
                            sample_base + sample_range > m->n_ents))
        return set_error(AMDKGE_EINVAL, "train: sampling range outside the entity table");
    TrainArgs a{};
    a.ent = d_ent; a.rel = d_rel; a.triples = d_triples; a.neg_override = d_neg_override;
    a.g_ent = d_grad_ent; a.g_rel = d_grad_rel; a.loss_sum = d_loss_sum;
    a.pos_scores = d_pos_scores; a.neg_scores = d_neg_scores;
    a.B = B; a.eta = eta; a.k = m->k; a.K = internal_k_of(m->scoring_type, m->k);
    a.sc = SampleCfg{sample_base, (uint32_t)sample_range, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                     (uint32_t)(step >> 32), row_offset, b_global > 0 ? b_global : B};
    a.mc = model_const(m);
    a.loss = *loss;
    { const char* e = getenv("AMDKGE_DEBUG"); a.dbg = e ? atoi(e) : 0; }
    hipStream_t st = (hipStream_t)stream;
    switch (m->scoring_type) {
        case AMDKGE_TRANSE: return launch_train_m<AMDKGE_TRANSE>(a, st);
        case AMDKGE_DISTMULT: return launch_train_m<AMDKGE_DISTMULT>(a, st);
        case AMDKGE_COMPLEX: return launch_train_m<AMDKGE_COMPLEX>(a, st);
        case AMDKGE_HOLE: return launch_train_m<AMDKGE_COMPLEX>(a, st);  // HolE = ComplEx * fp32(2/k), via ModelConst
        default: return launch_train_m<AMDKGE_ROTATE>(a, st);
    }
}
