// evaluate(use_filter=...): the true-positive filter index, built ON THE DEVICE.
//
// Replaces the reference's per-batch pandas group-by + Python `sum(lists, [])`
// (/root/reference/ampligraph/datasets/graph_data_loader.py:287-350,382-439) and this repo's earlier host-side numpy
// restatement of it (datasets/filters.py, kept as the checker): for a test triple (s, p, o) the subject-side filter is the SET
// {s' : (s', p, o) in any filter dataset}, the object-side filter the SET {o' : (s, p, o') in any filter dataset}.
// As a CSR over sorted group keys:
//     subject side: group key (p * N + o), values s      object side: group key (s * R + p), values o
//   1. one 64-bit key per filter triple, key = group * N + value              (filter_keys_kernel)
//   2. device radix sort of the keys (rocPRIM, only the bits R * N^2 needs)
//   3. flags "new key" / "new group" per sorted position, one exclusive scan of the packed pair of counters (rocPRIM)
//   4. scatter: ids[unique position] = key % N; keys[group] = key / N, start[group] = unique position   (filter_emit_kernel)
// The per-triple lookup into this index is amdkge_filter_ranges (kge_rank.hip).  HBM-bound integer work: 310 k filter
// triples at C2 are 2.5 MB of keys -- microseconds; the point is that evaluate() no longer spends 15 ms in host sorts.
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "kge_host.h"

namespace kge {

__global__ void filter_keys_kernel(const int32_t* __restrict__ tri, int64_t m, int side, uint64_t N, uint64_t R, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t s = (uint64_t)tri[3 * i], p = (uint64_t)tri[3 * i + 1], o = (uint64_t)tri[3 * i + 2];
    keys[i] = (side == AMDKGE_SIDE_S) ? (p * N + o) * N + s : (s * R + p) * N + o;
}

// low 32 bits: 1 where a NEW KEY starts (duplicates of a triple across the filter datasets collapse), high 32 bits: 1 where a
// new GROUP starts
__global__ void filter_flags_kernel(const uint64_t* __restrict__ k, int64_t m, uint64_t N, uint64_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool nk = i == 0 || k[i] != k[i - 1];
    const bool ng = i == 0 || k[i] / N != k[i - 1] / N;
    flags[i] = (nk ? 1ull : 0ull) | (ng ? (1ull << 32) : 0ull);
}

__global__ void filter_emit_kernel(const uint64_t* __restrict__ k, const uint64_t* __restrict__ flags, const uint64_t* __restrict__ pos,
                                   int64_t m, uint64_t N, int64_t* __restrict__ keys_out, int64_t* __restrict__ start, int32_t* __restrict__ ids,
                                   int64_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t f = flags[i], at = pos[i];
    const int64_t u = (int64_t)(at & 0xFFFFFFFFull), g = (int64_t)(at >> 32);
    if (f & 1ull) ids[u] = (int32_t)(k[i] % N);
    if (f >> 32) { keys_out[g] = (int64_t)(k[i] / N); start[g] = u; }
    if (i == m - 1) {
        const int64_t n_unique = u + (int64_t)(f & 1ull), n_groups = g + (int64_t)(f >> 32);
        start[n_groups] = n_unique;
        counts[0] = n_groups;
        counts[1] = n_unique;
    }
}

struct FilterPlan {
    size_t off_a, off_b, off_tmp, tmp_bytes, total;
    int key_bits;
};

static int key_bits_of(int64_t n_ents, int64_t n_rels) {
    // keys < R * N^2 (subject side: (p N + o) N + s; object side: (s R + p) N + o)
    long double top = (long double)n_rels * (long double)n_ents * (long double)n_ents;
    int b = 1;
    while (b < 64 && ldexpl(1.0L, b) < top) ++b;
    return b;
}

static int make_filter_plan(int64_t m, int64_t n_ents, int64_t n_rels, FilterPlan& p) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    p.key_bits = key_bits_of(n_ents, n_rels);
    size_t t_sort = 0, t_scan = 0;
    uint64_t* nul = nullptr;
    if (rocprim::radix_sort_keys(nullptr, t_sort, nul, nul, (size_t)m, 0u, (unsigned)p.key_bits, (hipStream_t)0) != hipSuccess) return -1;
    if (rocprim::exclusive_scan(nullptr, t_scan, nul, nul, (uint64_t)0, (size_t)m, rocprim::plus<uint64_t>(), (hipStream_t)0) != hipSuccess) return -1;
    p.tmp_bytes = t_sort > t_scan ? t_sort : t_scan;
    size_t o = 0;
    p.off_a = o; o += up((size_t)m * 8);     // unsorted keys, then the flags
    p.off_b = o; o += up((size_t)m * 8);     // sorted keys
    p.off_tmp = o; o += up(p.tmp_bytes);     // rocPRIM scratch, then (behind the sort) the scanned positions share off_a's partner
    p.total = o + up((size_t)m * 8) + 256;   // + scanned positions
    return 0;
}

}  // namespace kge

using namespace kge;

extern "C" int64_t amdkge_filter_build_workspace_bytes(int64_t m, int64_t n_ents, int64_t n_rels) {
    if (m < 0 || n_ents <= 0 || n_rels <= 0) return -1;
    if (m == 0) return 256;
    FilterPlan p;
    if (make_filter_plan(m, n_ents, n_rels, p)) return -1;
    return (int64_t)p.total;
}

extern "C" int amdkge_filter_build(const int32_t* d_triples, int64_t m, int32_t side, int64_t n_ents, int64_t n_rels,
                                   int64_t* d_keys, int64_t* d_start, int32_t* d_ids, int64_t* d_counts, void* d_work, void* stream) {
    if (side != AMDKGE_SIDE_S && side != AMDKGE_SIDE_O) return set_error(AMDKGE_EINVAL, "filter_build: side must be AMDKGE_SIDE_S or AMDKGE_SIDE_O");
    if (m < 0 || n_ents <= 0 || n_rels <= 0 || m > 0xFFFFFFFFll) return set_error(AMDKGE_EINVAL, "filter_build: bad sizes (at most 2^32 - 1 filter triples)");
    if ((long double)n_rels * (long double)n_ents * (long double)n_ents >= 9.2e18L)
        return set_error(AMDKGE_EUNSUPPORTED, "filter_build: n_rels * n_ents^2 does not fit the packed 64-bit sort keys");
    if (!d_start || !d_counts) return set_error(AMDKGE_EINVAL, "filter_build: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        if (hipError_t e = hipMemsetAsync(d_start, 0, 8, st)) return set_error_hip(e, "hipMemsetAsync(filter start)");
        if (hipError_t e = hipMemsetAsync(d_counts, 0, 16, st)) return set_error_hip(e, "hipMemsetAsync(filter counts)");
        return AMDKGE_OK;
    }
    if (!d_triples || !d_keys || !d_ids || !d_work) return set_error(AMDKGE_EINVAL, "filter_build: NULL pointer");
    FilterPlan p;
    if (make_filter_plan(m, n_ents, n_rels, p)) return set_error(AMDKGE_EHIP, "filter_build: rocPRIM size query failed");
    char* w = (char*)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    uint64_t* a = (uint64_t*)(w + p.off_a);
    uint64_t* b = (uint64_t*)(w + p.off_b);
    void* tmp = w + p.off_tmp;
    uint64_t* pos = (uint64_t*)(w + p.off_tmp + ((p.tmp_bytes + 255) & ~(size_t)255));
    const unsigned grid = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(filter_keys_kernel, dim3(grid), dim3(256), 0, st, d_triples, m, (int)side, (uint64_t)n_ents, (uint64_t)n_rels, a);
    if (int rc = check_launch("filter_keys")) return rc;
    size_t tb = p.tmp_bytes;
    if (hipError_t e = rocprim::radix_sort_keys(tmp, tb, a, b, (size_t)m, 0u, (unsigned)p.key_bits, st)) return set_error_hip(e, "rocprim::radix_sort_keys");
    hipLaunchKernelGGL(filter_flags_kernel, dim3(grid), dim3(256), 0, st, b, m, (uint64_t)n_ents, a);
    if (int rc = check_launch("filter_flags")) return rc;
    tb = p.tmp_bytes;
    if (hipError_t e = rocprim::exclusive_scan(tmp, tb, a, pos, (uint64_t)0, (size_t)m, rocprim::plus<uint64_t>(), st)) return set_error_hip(e, "rocprim::exclusive_scan");
    hipLaunchKernelGGL(filter_emit_kernel, dim3(grid), dim3(256), 0, st, b, a, pos, m, (uint64_t)n_ents, d_keys, d_start, d_ids, d_counts);
    return check_launch("filter_emit");
}
