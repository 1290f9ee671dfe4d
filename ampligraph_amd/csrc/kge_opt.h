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
    // second LP term of the same table (Keras l1_l2): gradient weight lam2, value weight relative to the first term
    // (the sweeps accumulate sum |x|^p + ratio2 * |x|^p2 and scale by lam once at the end); lam2 == 0: none
    float lam2, ratio2;
    int reg_p2;
    int lazy, row_floats;   // touched-rows mode (include/amdkge.h, amdkge_opt.lazy)
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
        if (a.lam2 != 0.f) {
            reg_acc += a.ratio2 * ipowf(ax, a.reg_p2);
            g += a.lam2 * (float)a.reg_p2 * ipowf(ax, a.reg_p2 - 1) * sg;
        }
    }
    if constexpr (KIND == AMDKGE_OPT_ADAM) {
        s0 = s0 * a.beta1 + g * a.omb1;
        s1 = s1 * a.beta2 + (g * g) * a.omb2;
        x -= (a.lr_t * s0) / (sqrtf(s1) + a.eps);
    } else if constexpr (KIND == AMDKGE_OPT_ADAGRAD) {
        s0 += g * g;
        x -= a.lr * g / (sqrtf(s0) + a.eps);
    } else if constexpr (KIND == AMDKGE_OPT_MOMENTUM) {      // ResourceApplyKerasMomentum
        s0 = s0 * a.beta1 - a.lr * g;
        x += (a.beta2 != 0.f) ? (s0 * a.beta1 - a.lr * g) : s0;
    } else if constexpr (KIND == AMDKGE_OPT_RMSPROP) {       // RMSprop._resource_apply_dense, momentum == 0
        s0 += (g * g - s0) * a.omb1;
        x -= a.lr * g / (sqrtf(s0) + a.eps);
    } else if constexpr (KIND == AMDKGE_OPT_RMSPROP_MOM) {   // ResourceApplyRMSProp
        s0 += (g * g - s0) * a.omb1;
        s1 = s1 * a.beta2 + (a.lr * g) / sqrtf(s0 + a.eps);
        x -= s1;
    } else if constexpr (KIND == AMDKGE_OPT_ADADELTA) {      // ResourceApplyAdadelta
        s0 = s0 * a.beta1 + (g * g) * a.omb1;
        const float u = sqrtf(s1 + a.eps) * (1.f / sqrtf(s0 + a.eps)) * g;
        x -= u * a.lr;
        s1 = s1 * a.beta1 + (u * u) * a.omb1;
    } else if constexpr (KIND == AMDKGE_OPT_ADAMAX) {        // ResourceApplyAdaMax
        s0 += (g - s0) * a.omb1;
        s1 = fmaxf(a.beta2 * s1, fabsf(g));
        x -= a.lr_t * s0 / (s1 + a.eps);
    } else {
        x -= a.lr * g;
    }
}

// number of optimizer state tensors a kind keeps per table
__host__ __device__ constexpr int opt_nslots(int kind) {
    return kind == AMDKGE_OPT_SGD ? 0
         : (kind == AMDKGE_OPT_ADAGRAD || kind == AMDKGE_OPT_MOMENTUM || kind == AMDKGE_OPT_RMSPROP) ? 1 : 2;
}

// KGE_OPT_DISPATCH(kind, F): run F(KIND) with KIND a compile-time constant
#define KGE_OPT_DISPATCH(kind, F)                                              \
    switch (kind) {                                                            \
        case AMDKGE_OPT_ADAM: F(AMDKGE_OPT_ADAM); break;                       \
        case AMDKGE_OPT_ADAGRAD: F(AMDKGE_OPT_ADAGRAD); break;                 \
        case AMDKGE_OPT_MOMENTUM: F(AMDKGE_OPT_MOMENTUM); break;               \
        case AMDKGE_OPT_RMSPROP: F(AMDKGE_OPT_RMSPROP); break;                 \
        case AMDKGE_OPT_RMSPROP_MOM: F(AMDKGE_OPT_RMSPROP_MOM); break;         \
        case AMDKGE_OPT_ADADELTA: F(AMDKGE_OPT_ADADELTA); break;               \
        case AMDKGE_OPT_ADAMAX: F(AMDKGE_OPT_ADAMAX); break;                   \
        default: F(AMDKGE_OPT_SGD); break;                                     \
    }


// Grid-stride dense sweep over a.n elements (float4 body + scalar tail): optimizer + regulariser + gradient
// reset.  `first` = this thread's index among `stride` cooperating threads.  Returns the thread's sum |x|^p.
template <int KIND>
__device__ __forceinline__ float opt_sweep(const OptArgs& a, int64_t first, int64_t stride) {
    const int64_t n4 = a.n >> 2;
    float reg_acc = 0.f;
    float4* x4 = reinterpret_cast<float4*>(a.x);
    float4* g4 = reinterpret_cast<float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.s0);
    float4* v4 = reinterpret_cast<float4*>(a.s1);
    for (int64_t i = first; i < n4; i += stride) {
        float4 x = x4[i], g = g4[i];
        float4 m = make_float4(0, 0, 0, 0), v = make_float4(0, 0, 0, 0);
        if constexpr (opt_nslots(KIND) >= 1) m = m4[i];
        if constexpr (opt_nslots(KIND) == 2) v = v4[i];
        opt_elem<KIND>(a, x.x, g.x, m.x, v.x, reg_acc);
        opt_elem<KIND>(a, x.y, g.y, m.y, v.y, reg_acc);
        opt_elem<KIND>(a, x.z, g.z, m.z, v.z, reg_acc);
        opt_elem<KIND>(a, x.w, g.w, m.w, v.w, reg_acc);
        x4[i] = x;
        g4[i] = make_float4(0, 0, 0, 0);
        if constexpr (opt_nslots(KIND) >= 1) m4[i] = m;
        if constexpr (opt_nslots(KIND) == 2) v4[i] = v;
    }
    // scalar tail (n % 4)
    for (int64_t i = (n4 << 2) + first; i < a.n; i += stride) {
        float x = a.x[i], g = a.g[i], m = 0.f, v = 0.f;
        if constexpr (opt_nslots(KIND) >= 1) m = a.s0[i];
        if constexpr (opt_nslots(KIND) == 2) v = a.s1[i];
        opt_elem<KIND>(a, x, g, m, v, reg_acc);
        a.x[i] = x;
        a.g[i] = 0.f;
        if constexpr (opt_nslots(KIND) >= 1) a.s0[i] = m;
        if constexpr (opt_nslots(KIND) == 2) a.s1[i] = v;
    }
    return reg_acc;
}

// Touched-rows sweep: one wave per row (grid-stride over rows).  The wave first reads the gradient row and ORs "non-zero";
// an all-zero row is left alone -- x, the slots and the regulariser are not even read.  A touched row is re-read from
// cache and goes through the same opt_elem as the dense sweep, and its gradient row is reset.
template <int KIND>
__device__ __forceinline__ float opt_sweep_rows(const OptArgs& a, int64_t first_row, int64_t row_stride, int lane) {
    const int nq = a.row_floats >> 2;
    const int64_t rows = a.n / a.row_floats;
    float reg_acc = 0.f;
    for (int64_t r = first_row; r < rows; r += row_stride) {
        const int64_t base = r * (int64_t)nq;
        const float4* g4 = reinterpret_cast<const float4*>(a.g) + base;
        bool nz = false;
        for (int q = lane; q < nq; q += KGE_WAVE) {
            const float4 g = g4[q];
            nz |= (g.x != 0.f) | (g.y != 0.f) | (g.z != 0.f) | (g.w != 0.f);
        }
        if (!__ballot(nz)) continue;
        float4* x4 = reinterpret_cast<float4*>(a.x) + base;
        float4* m4 = reinterpret_cast<float4*>(a.s0) + base;
        float4* v4 = reinterpret_cast<float4*>(a.s1) + base;
        float4* gw4 = reinterpret_cast<float4*>(a.g) + base;
        for (int q = lane; q < nq; q += KGE_WAVE) {
            float4 x = x4[q], g = gw4[q];
            float4 m = make_float4(0, 0, 0, 0), v = make_float4(0, 0, 0, 0);
            if constexpr (opt_nslots(KIND) >= 1) m = m4[q];
            if constexpr (opt_nslots(KIND) == 2) v = v4[q];
            opt_elem<KIND>(a, x.x, g.x, m.x, v.x, reg_acc);
            opt_elem<KIND>(a, x.y, g.y, m.y, v.y, reg_acc);
            opt_elem<KIND>(a, x.z, g.z, m.z, v.z, reg_acc);
            opt_elem<KIND>(a, x.w, g.w, m.w, v.w, reg_acc);
            x4[q] = x;
            gw4[q] = make_float4(0, 0, 0, 0);
            if constexpr (opt_nslots(KIND) >= 1) m4[q] = m;
            if constexpr (opt_nslots(KIND) == 2) v4[q] = v;
        }
    }
    return reg_acc;
}

// regulariser terms of one table -> OptArgs.  The kernels key everything on lam != 0, so a lone second term becomes the first.
inline void set_reg_terms(OptArgs& a, int p1, float lam1, int p2, float lam2) {
    if (lam1 == 0.f && lam2 != 0.f) { p1 = p2; lam1 = lam2; lam2 = 0.f; }
    a.lam = lam1; a.reg_p = p1; a.lam2 = lam2; a.reg_p2 = p2;
    a.ratio2 = (lam2 != 0.f) ? (float)((double)lam2 / (double)lam1) : 0.f;
}

// host-side: fill the derived fields of OptArgs from the ABI descriptor
inline void fill_opt_args(OptArgs& a, const amdkge_opt* opt) {
    a.lr = opt->lr; a.beta1 = opt->beta1; a.beta2 = opt->beta2; a.eps = opt->epsilon;
    a.omb1 = (float)(1.0 - (double)opt->beta1);   // python: 1 - beta_1, cast to fp32 like the TF constant
    a.omb2 = (float)(1.0 - (double)opt->beta2);
    a.kind = opt->kind;
    set_reg_terms(a, opt->reg_p, opt->reg_lambda, opt->reg2_p, opt->reg2_lambda);
    a.lazy = opt->lazy ? 1 : 0; a.row_floats = opt->row_floats;
    const double t = (double)opt->iteration;
    if (opt->kind == AMDKGE_OPT_ADAMAX) a.lr_t = (float)((double)opt->lr / (1.0 - pow((double)opt->beta1, t)));
    else a.lr_t = (float)((double)opt->lr * sqrt(1.0 - pow((double)opt->beta2, t)) / (1.0 - pow((double)opt->beta1, t)));
}

inline int validate_opt(const amdkge_opt* opt) {
    if (!opt) return set_error(AMDKGE_EINVAL, "NULL optimizer descriptor");
    if (opt->kind < AMDKGE_OPT_SGD || opt->kind > AMDKGE_OPT_ADAMAX) return set_error(AMDKGE_EINVAL, "unknown optimizer kind");
    if (opt->iteration < 1) return set_error(AMDKGE_EINVAL, "optimizer iteration is 1-based");
    if (opt->reg_lambda != 0.f && opt->reg_p < 1) return set_error(AMDKGE_EINVAL, "regulariser p must be >= 1");
    if (opt->reg2_lambda != 0.f && opt->reg2_p < 1) return set_error(AMDKGE_EINVAL, "regulariser p (second term) must be >= 1");
    if (opt->rel_reg2_lambda != 0.f && opt->rel_reg2_p < 1) return set_error(AMDKGE_EINVAL, "relation regulariser p (second term) must be >= 1");
    if (opt->rel_reg_p < 0) return set_error(AMDKGE_EINVAL, "rel_reg_p must be 0 (= reg_p) or >= 1");
    if (opt->lazy && (opt->row_floats < 4 || opt->row_floats % 4 != 0))
        return set_error(AMDKGE_EINVAL, "lazy optimizer mode needs row_floats = amdkge_row_floats(model) of a padded layout (multiple of 4)");
    return AMDKGE_OK;
}

}  // namespace kge
