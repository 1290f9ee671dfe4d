// Dense optimizer sweep fused with the whole-table LP regulariser and the gradient-buffer reset.
// Replaces OptimizerWrapper.minimize -> Keras *legacy* apply_gradients
// (/root/reference/ampligraph/latent_features/optimizers.py:136-168; update rules live in the
// third-party tensorflow==2.15 wheel, keras/optimizers/legacy/{adam,adagrad,gradient_descent,rmsprop,adadelta,adamax}.py)
// and LP_regularizer (regularizers.py:35-37).  Non-lazy like the reference: every row's slots
// decay and every row moves each step.  HBM-bound: reads x,m,v,g and writes x,m,v,g(=0), 16 B/lane.
#include "kge_opt.h"

namespace kge {

template <int KIND>
__global__ __launch_bounds__(256) void opt_kernel(OptArgs a) {
    const float reg_acc = opt_sweep<KIND>(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    if (a.reg_loss && a.lam != 0.f) {
        __shared__ float red[4];
        const float w = wave_sum(reg_acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.reg_loss, (double)a.lam * ((double)red[0] + red[1] + red[2] + red[3]));
    }
}

// touched-rows variant (amdkge_opt.lazy): one wave per row, see opt_sweep_rows
template <int KIND>
__global__ __launch_bounds__(256) void opt_rows_kernel(OptArgs a) {
    const float reg_acc = opt_sweep_rows<KIND>(a, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4, threadIdx.x & 63);
    if (a.reg_loss && a.lam != 0.f) {
        __shared__ float red[4];
        const float w = wave_sum(reg_acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.reg_loss, (double)a.lam * ((double)red[0] + red[1] + red[2] + red[3]));
    }
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_opt_step(const amdkge_opt* opt, float* d_x, float* d_grad, float* d_slot0, float* d_slot1,
                               int64_t n_elems, double* d_reg_loss, void* stream) {
    if (int rc = validate_opt(opt)) return rc;
    if (n_elems < 0) return set_error(AMDKGE_EINVAL, "opt_step: n_elems must be >= 0");
    if (n_elems == 0) return AMDKGE_OK;
    if (!d_x || !d_grad) return set_error(AMDKGE_EINVAL, "opt_step: NULL table / gradient pointer");
    if (opt_nslots(opt->kind) >= 1 && !d_slot0) return set_error(AMDKGE_EINVAL, "opt_step: optimizer slot 0 is NULL");
    if (opt_nslots(opt->kind) == 2 && !d_slot1) return set_error(AMDKGE_EINVAL, "opt_step: optimizer slot 1 is NULL");
    if ((((uintptr_t)d_x | (uintptr_t)d_grad | (uintptr_t)d_slot0 | (uintptr_t)d_slot1) & 15) != 0)
        return set_error(AMDKGE_EINVAL, "opt_step: buffers must be 16-byte aligned");
    OptArgs a{};
    a.x = d_x; a.g = d_grad; a.s0 = d_slot0; a.s1 = d_slot1; a.n = n_elems; a.reg_loss = d_reg_loss;
    fill_opt_args(a, opt);
    hipStream_t st = (hipStream_t)stream;
    if (a.lazy) {
        if (n_elems % a.row_floats != 0) return set_error(AMDKGE_EINVAL, "opt_step: lazy mode sweeps whole rows (n_elems must be a multiple of row_floats)");
        const int64_t rows = n_elems / a.row_floats;
        unsigned gridr = (unsigned)((rows + 3) / 4 < 256 * 16 ? (rows + 3) / 4 : 256 * 16);
#define KGE_OPT_LAUNCH_ROWS(KIND) hipLaunchKernelGGL(opt_rows_kernel<KIND>, dim3(gridr), dim3(256), 0, st, a)
        KGE_OPT_DISPATCH(opt->kind, KGE_OPT_LAUNCH_ROWS)
#undef KGE_OPT_LAUNCH_ROWS
        return check_launch("opt_step(lazy)");
    }
    const int64_t n4 = (n_elems + 3) / 4;
    unsigned grid = (unsigned)((n4 + 255) / 256);
    if (grid > 2048) grid = 2048;   // 256 CUs x 8 blocks, grid-stride beyond
#define KGE_OPT_LAUNCH(KIND) hipLaunchKernelGGL(opt_kernel<KIND>, dim3(grid), dim3(256), 0, st, a)
    KGE_OPT_DISPATCH(opt->kind, KGE_OPT_LAUNCH)
#undef KGE_OPT_LAUNCH
    return check_launch("opt_step");
}
