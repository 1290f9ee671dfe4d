// predict(): fused lookup + pointwise score, one wave64 per triple, 16-byte coalesced row reads,
// DPP wave reduction.  Replaces EmbeddingLookupLayer.call + <Model>._compute_scores
// (/root/reference/ampligraph/latent_features/layers/encoding/EmbeddingLookupLayer.py:307-342,
//  layers/scoring/{TransE.py:37,DistMult.py:34,ComplEx.py:39,HolE.py:31,RotatE.py:62}).
#include "kge_host.h"

namespace kge {

template <int MODEL, int VEC>
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ ent, const float* __restrict__ rel,
                                                    const int32_t* __restrict__ triples, int64_t n, int k, int K,
                                                    int nq, ModelConst mc, float* __restrict__ out) {
    constexpr int NC = ModelTraits<MODEL>::NC;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* rs = ent + (int64_t)triples[3 * i + 0] * K;
    const float* rp = rel + (int64_t)triples[3 * i + 1] * K;
    const float* ro = ent + (int64_t)triples[3 * i + 2] * K;
    float part = 0.f;
    for (int q = lane; q < nq; q += KGE_WAVE) {
        float s[VEC][NC], p[VEC][NC], o[VEC][NC];
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            const fvec<VEC> vs = ldg<VEC>(rs + q * VEC + h * k);
            const fvec<VEC> vp = ldg<VEC>(rp + q * VEC + h * k);
            const fvec<VEC> vo = ldg<VEC>(ro + q * VEC + h * k);
#pragma unroll
            for (int u = 0; u < VEC; ++u) { s[u][h] = vs.v[u]; p[u][h] = vp.v[u]; o[u][h] = vo.v[u]; }
        }
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            prep_rel_exact<MODEL>(mc, p[u]);   // RotatE: the declared correctly rounded cos / sin (same values as evaluate)
            part += score_unit<MODEL>(s[u], p[u], o[u]);
        }
    }
    const float tot = wave_sum(part);
    if (lane == 0) out[i] = mc.score_sign * mc.score_scale * tot;
}

template <int MODEL>
static int launch_score(const float* ent, const float* rel, const int32_t* tr, int64_t n, int k, int K, ModelConst mc,
                        float* out, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (k % 4 == 0) hipLaunchKernelGGL((score_kernel<MODEL, 4>), dim3(grid), dim3(256), 0, st, ent, rel, tr, n, k, K, k / 4, mc, out);
    else if (k % 2 == 0) hipLaunchKernelGGL((score_kernel<MODEL, 2>), dim3(grid), dim3(256), 0, st, ent, rel, tr, n, k, K, k / 2, mc, out);
    else hipLaunchKernelGGL((score_kernel<MODEL, 1>), dim3(grid), dim3(256), 0, st, ent, rel, tr, n, k, K, k, mc, out);
    return check_launch("score");
}

int score_dispatch(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples, int64_t n,
                   float* d_scores, hipStream_t st) {
    // stored layout: padding units score exactly 0 in every model (RotatE: sqrt(0)), so the kernels just see ks units
    const int ks = stored_k(m), K = row_floats(m);
    const ModelConst mc = model_const(m);
    switch (m->scoring_type) {
        case AMDKGE_TRANSE: return launch_score<AMDKGE_TRANSE>(d_ent, d_rel, d_triples, n, ks, K, mc, d_scores, st);
        case AMDKGE_DISTMULT: return launch_score<AMDKGE_DISTMULT>(d_ent, d_rel, d_triples, n, ks, K, mc, d_scores, st);
        case AMDKGE_COMPLEX:
        case AMDKGE_HOLE: return launch_score<AMDKGE_COMPLEX>(d_ent, d_rel, d_triples, n, ks, K, mc, d_scores, st);
        default: return launch_score<AMDKGE_ROTATE>(d_ent, d_rel, d_triples, n, ks, K, mc, d_scores, st);
    }
}

// calibrate(): one evaluation of the Platt-scaling objective and its gradient for a batch of scores.
// Replaces CalibrationLayer.call(training=1) (/root/reference/ampligraph/latent_features/layers/calibration/
// calibrate.py:78-129) + the tape gradient of ScoringBasedEmbeddingModel.calibrate (:2108-2121):
//   logit = -(w*s + b); loss = mean_i weight_i * (max(x,0) - x*z_i + log(1+exp(-|x|)))   (sigmoid_cross_entropy_with_logits)
//   z = label_pos | label_neg, weight = weight_pos | weight_neg by the side the score came from.
// out[0] += loss, out[1] += dloss/dw, out[2] += dloss/db   (fp64 accumulators)
__global__ __launch_bounds__(256) void platt_kernel(const float* __restrict__ sp, int64_t np_, const float* __restrict__ sn,
                                                    int64_t nn, float w, float b, float zp, float zn, float wp, float wn,
                                                    double* out) {
    const int64_t n = np_ + nn;
    float l = 0.f, gw = 0.f, gb = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool pos = i < np_;
        const float s = pos ? sp[i] : sn[i - np_];
        const float z = pos ? zp : zn, wt = pos ? wp : wn;
        const float x = -(w * s + b);
        l += wt * (fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))));
        const float d = wt * (1.f / (1.f + expf(-x)) - z);   // dloss_i/dx
        gw += -s * d;
        gb += -d;
    }
    const float inv = 1.f / (float)n;
    l = wave_sum(l); gw = wave_sum(gw); gb = wave_sum(gb);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, (double)(l * inv));
        atomicAdd(out + 1, (double)(gw * inv));
        atomicAdd(out + 2, (double)(gb * inv));
    }
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_platt_step(const float* d_scores_pos, int64_t n_pos, const float* d_scores_neg, int64_t n_neg, float w,
                                 float b, float label_pos, float label_neg, float weight_pos, float weight_neg,
                                 double* d_out3, void* stream) {
    if (n_pos < 0 || n_neg < 0) return set_error(AMDKGE_EINVAL, "platt_step: negative size");
    if (n_pos + n_neg == 0) return AMDKGE_OK;
    if ((n_pos > 0 && !d_scores_pos) || (n_neg > 0 && !d_scores_neg) || !d_out3) return set_error(AMDKGE_EINVAL, "platt_step: NULL pointer");
    const int64_t n = n_pos + n_neg;
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(platt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_scores_pos, n_pos, d_scores_neg, n_neg, w, b,
                       label_pos, label_neg, weight_pos, weight_neg, d_out3);
    return check_launch("platt_step");
}

extern "C" int amdkge_score(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples,
                            int64_t n, float* d_scores, void* stream) {
    if (int rc = validate_model(m)) return rc;
    if (n < 0) return set_error(AMDKGE_EINVAL, "score: n must be >= 0");
    if (n == 0) return AMDKGE_OK;
    if (!d_ent || !d_rel || !d_triples || !d_scores) return set_error(AMDKGE_EINVAL, "score: NULL pointer");
    return score_dispatch(m, d_ent, d_rel, d_triples, n, d_scores, (hipStream_t)stream);
}
