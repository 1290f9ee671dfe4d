// Discovery helpers on the device: per-row top-k selection and squared row norms.  Together with
// amdkge_corruption_scores / amdkge_row_dots (kge_rank.hip) they replace the host side of
// /root/reference/ampligraph/discovery/discovery.py:985-1168 (query_topn: one STRING triple per candidate through
// model.predict, then np.argsort) and :1171-1244 (find_nearest_neighbours: sklearn NearestNeighbors on the host).
#include "kge_host.h"

namespace kge {

constexpr int TOPK_MAX = 1024;          // largest k
constexpr int TOPK_BUF = 2 * TOPK_MAX;  // LDS candidates: the current best TOPK_MAX (sorted) + a staging half

// order-preserving map fp32 -> uint32 (larger float <=> larger key); NaN sorts below everything
__device__ __forceinline__ uint32_t sortable(float v) {
    if (v != v) return 0u;
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unsortable(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

// bitonic sort of buf[0 .. TOPK_BUF) in DESCENDING order by the 64-bit key (256 threads)
__device__ __forceinline__ void sort_desc(unsigned long long* buf, int tid) {
    for (int k = 2; k <= TOPK_BUF; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < TOPK_BUF; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const bool desc = (i & k) == 0;
                    const unsigned long long a = buf[i], b = buf[p];
                    if (desc ? (a < b) : (a > b)) { buf[i] = b; buf[p] = a; }
                }
            }
            __syncthreads();
        }
}

// One workgroup per row: streaming top-k.  key = sortable(value) << 32 | ~column, so equal values are ordered by
// LOWER column first and the result is deterministic.  Values at or below the current k-th best are dropped as they
// stream by; survivors collect in the staging half and are merged (one bitonic sort) whenever it fills.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ vals, int64_t m, int64_t ld, const float* __restrict__ col_scale,
                                                        const float* __restrict__ col_bias, const int32_t* __restrict__ payload, int k, int largest,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
    __shared__ unsigned long long buf[TOPK_BUF];
    __shared__ int n_stage;
    const int tid = threadIdx.x;
    const float* row = vals + (int64_t)blockIdx.x * ld;
    for (int i = tid; i < TOPK_BUF; i += 256) buf[i] = 0ull;
    if (tid == 0) n_stage = 0;
    __syncthreads();
    unsigned long long kth = 0ull;   // key of the current k-th best (0: fewer than k candidates so far)
    for (int64_t c0 = 0; c0 < m; c0 += 256) {
        const int64_t c = c0 + tid;
        if (c < m) {
            float v = row[c];
            if (col_scale) v *= col_scale[c];
            if (col_bias) v += col_bias[c];
            if (!largest) v = -v;
            const unsigned long long key = ((unsigned long long)sortable(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
            if (key > kth) buf[TOPK_MAX + atomicAdd(&n_stage, 1)] = key;
        }
        __syncthreads();
        if (n_stage > TOPK_MAX - 256 || c0 + 256 >= m) {   // staging (nearly) full, or end of the row: merge
            sort_desc(buf, tid);
            if (tid == 0) n_stage = 0;
            kth = buf[k - 1];
            __syncthreads();
            for (int i = TOPK_MAX + tid; i < TOPK_BUF; i += 256) buf[i] = 0ull;
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const unsigned long long key = buf[i];
        const bool have = key != 0ull;
        float v = unsortable((uint32_t)(key >> 32));
        const int32_t col = (int32_t)(0xFFFFFFFFu - (uint32_t)key);
        out_idx[(int64_t)blockIdx.x * k + i] = have ? (payload ? payload[(int64_t)blockIdx.x * ld + col] : col) : -1;
        out_val[(int64_t)blockIdx.x * k + i] = have ? (largest ? v : -v) : (largest ? -INFINITY : INFINITY);
    }
}

// out[j] = mul * sum_c row(ids[j])[c]^2, or its reciprocal square root (rsq): one wave per row
__global__ __launch_bounds__(256) void row_sqnorms_kernel(const float* __restrict__ table, int K, const int32_t* __restrict__ ids, int64_t lo, int64_t n,
                                                          float mul, int rsq, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const float* row = table + (ids ? (int64_t)ids[lo + j] : lo + j) * K;
    float acc = 0.f;
    for (int c = lane; c < K; c += 64) acc = fmaf(row[c], row[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[j] = rsq ? 1.f / sqrtf(fmaxf(acc, 1e-30f)) : mul * acc;
}

// exact distances of explicit (query row i, table row pos[i][j]) pairs: one wave per pair.  The GEMM form used for the
// selection (|q|^2 + |e|^2 - 2 <q,e>) cancels catastrophically for near neighbours; the k survivors are re-measured here.
__global__ __launch_bounds__(256) void pair_dist_kernel(const float* __restrict__ q, const float* __restrict__ table, int K, const int32_t* __restrict__ ids,
                                                        int64_t lo, const int32_t* __restrict__ pos, int64_t n, int k, int cosine, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n * k) return;
    const int32_t p = pos[t];
    if (p < 0) { if (lane == 0) out[t] = INFINITY; return; }
    const float* qr = q + (t / k) * K;
    const float* er = table + (ids ? (int64_t)ids[lo + p] : lo + p) * K;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int u = lane; u < K; u += 64) {
        const float x = qr[u], y = er[u];
        if (cosine) { a = fmaf(x, y, a); b = fmaf(x, x, b); c = fmaf(y, y, c); }
        else { const float d = x - y; a = fmaf(d, d, a); }
    }
    a = wave_sum(a);
    if (cosine) { b = wave_sum(b); c = wave_sum(c); }
    if (lane == 0) out[t] = cosine ? 1.f - a / (sqrtf(fmaxf(b, 1e-30f)) * sqrtf(fmaxf(c, 1e-30f))) : sqrtf(a);
}

}  // namespace kge

using namespace kge;

extern "C" int amdkge_topk_rows(const float* d_vals, int64_t n, int64_t m, int64_t ld, const float* d_col_scale, const float* d_col_bias,
                                const int32_t* d_payload, int32_t k, int32_t largest, int32_t* d_out_idx, float* d_out_val, void* stream) {
    if (n < 0 || m < 0 || ld < m || k < 1 || k > TOPK_MAX) return set_error(AMDKGE_EINVAL, "topk_rows: bad sizes (1 <= k <= 1024, ld >= m)");
    if (n > 0x7FFFFFFFll || m > 0xFFFFFFFEll) return set_error(AMDKGE_EUNSUPPORTED, "topk_rows: too many rows / columns for one call");
    if (n == 0) return AMDKGE_OK;
    if (!d_out_idx || !d_out_val || (m > 0 && !d_vals)) return set_error(AMDKGE_EINVAL, "topk_rows: NULL pointer");
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, d_vals, m, ld, d_col_scale, d_col_bias, d_payload,
                       (int)k, (int)largest, d_out_idx, d_out_val);
    return check_launch("topk_rows");
}

extern "C" int amdkge_row_sqnorms(const float* d_table, int32_t row_floats, const int32_t* d_ids, int64_t lo, int64_t n, float mul, int32_t rsqrt,
                                  float* d_out, void* stream) {
    if (n < 0 || lo < 0 || row_floats < 1) return set_error(AMDKGE_EINVAL, "row_sqnorms: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_table || !d_out) return set_error(AMDKGE_EINVAL, "row_sqnorms: NULL pointer");
    hipLaunchKernelGGL(row_sqnorms_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_table, (int)row_floats, d_ids, lo, n, mul, (int)rsqrt, d_out);
    return check_launch("row_sqnorms");
}

extern "C" int amdkge_pair_distances(const float* d_q, int64_t n, const float* d_table, int32_t row_floats, const int32_t* d_ids, int64_t lo,
                                     const int32_t* d_pos, int32_t k, int32_t cosine, float* d_out, void* stream) {
    if (n < 0 || k < 1 || lo < 0 || row_floats < 1) return set_error(AMDKGE_EINVAL, "pair_distances: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_q || !d_table || !d_pos || !d_out) return set_error(AMDKGE_EINVAL, "pair_distances: NULL pointer");
    const int64_t pairs = n * k;
    hipLaunchKernelGGL(pair_dist_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_q, d_table, (int)row_floats, d_ids, lo,
                       d_pos, n, (int)k, (int)cosine, d_out);
    return check_launch("pair_distances");
}
