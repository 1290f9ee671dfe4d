// Element-wise optimizer + regulariser arithmetic shared by the dense sweep (kge_opt.hip) and the
// owner-computes backward (kge_train_tiled.hip).  Keras *legacy* update rules, see kge_opt.hip.
#pragma once
#include "kge_host.h"

namespace kge {

struct OptArgs {
    float* x;
    float* g;
    float* s0;
    float* s1;
    int64_t n;
    double* reg_loss;
    float lr, lr_t, beta1, beta2, omb1, omb2, eps, lam;
    int kind, reg_p;
};

__device__ __forceinline__ float ipowf(float a, int p) {
    float r = 1.f;
    for (int i = 0; i < p; ++i) r *= a;
    return r;
}

template <int KIND>
__device__ __forceinline__ void opt_elem(const OptArgs& a, float& x, float g, float& s0, float& s1, float& reg_acc) {
    if (a.lam != 0.f) {
        const float ax = fabsf(x);
        // lambda * sum |x|^p ; d/dx = lambda * p * |x|^(p-1) * sign(x)
        reg_acc += ipowf(ax, a.reg_p);
        const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
        g += a.lam * (float)a.reg_p * ipowf(ax, a.reg_p - 1) * sg;
    }
    if constexpr (KIND == AMDKGE_OPT_ADAM) {
        s0 = s0 * a.beta1 + g * a.omb1;
        s1 = s1 * a.beta2 + (g * g) * a.omb2;
        x -= (a.lr_t * s0) / (sqrtf(s1) + a.eps);
    } else if constexpr (KIND == AMDKGE_OPT_ADAGRAD) {
        s0 += g * g;
        x -= a.lr * g / (sqrtf(s0) + a.eps);
    } else {
        x -= a.lr * g;
    }
}


// host-side: fill the derived fields of OptArgs from the ABI descriptor
inline void fill_opt_args(OptArgs& a, const amdkge_opt* opt) {
    a.lr = opt->lr; a.beta1 = opt->beta1; a.beta2 = opt->beta2; a.eps = opt->epsilon;
    a.omb1 = (float)(1.0 - (double)opt->beta1);   // python: 1 - beta_1, cast to fp32 like the TF constant
    a.omb2 = (float)(1.0 - (double)opt->beta2);
    a.lam = opt->reg_lambda; a.kind = opt->kind; a.reg_p = opt->reg_p;
    const double t = (double)opt->iteration;
    a.lr_t = (float)((double)opt->lr * sqrt(1.0 - pow((double)opt->beta2, t)) / (1.0 - pow((double)opt->beta1, t)));
}

inline int validate_opt(const amdkge_opt* opt) {
    if (!opt) return set_error(AMDKGE_EINVAL, "NULL optimizer descriptor");
    if (opt->kind < AMDKGE_OPT_SGD || opt->kind > AMDKGE_OPT_ADAM) return set_error(AMDKGE_EINVAL, "unknown optimizer kind");
    if (opt->iteration < 1) return set_error(AMDKGE_EINVAL, "optimizer iteration is 1-based");
    if (opt->reg_lambda != 0.f && opt->reg_p < 1) return set_error(AMDKGE_EINVAL, "regulariser p must be >= 1");
    return AMDKGE_OK;
}

}  // namespace kge
