// Multi-GPU data-path kernels of libamdkge (one process per GPU; the collectives themselves are RCCL calls made by the host).
//
// Row-sharded entity table (the reference's partitioned training, /root/reference/ampligraph/latent_features/
// models/ScoringBasedEmbeddingModel.py:227,259-261,1431-1452, with owner(e) = e / ceil(N / G) from
// datasets/graph_partitioner.py:339-344): rank r owns rows [lo, hi) and keeps scratch rows for fetched copies of remote rows
// BEHIND its shard in the same allocation.  Per step, entirely on the device and without a host round trip:
//   shard_route   : every entity id of the batch (s, o of the positives; with global negatives also the corruptions) is
//                   classified by owner; remote ids are de-duplicated through an open-addressing hash table and appended to
//                   the request list of their owner (fixed capacity per peer, so the all_to_all that follows has equal,
//                   host-known splits); the batch is rewritten into the local index space (local row, or scratch row
//                   n_local + peer * cap + position).
//   gather_rows   : the owner side of the exchange: requested rows -> a contiguous send buffer (16-byte loads / stores).
//   scatter_add_rows : gradient rows of fetched copies, returned to their owner, are added to the owner's gradient rows
//                   (hardware fp32 atomics: different peers may return the same row).
// Replicated tables (data parallel): opt_step_merged sums the W partial gradient slices a reduce-scatter-by-all_to_all
// delivered and applies the optimizer in the same sweep (replaces a separate sum kernel + amdkge_opt_step).
#include "kge_opt.h"

namespace kge {

struct RouteArgs {
    const int32_t* triples;   // [b][3] global ids
    const int32_t* negs;      // [nneg][3] or NULL
    int32_t* out_triples;     // [b][3] local index space
    int32_t* out_negs;        // [nneg][3]
    int64_t b, nneg;
    int64_t lo, hi;           // owned id range
    int64_t rows_per;         // ceil(N / world): owner(e) = e / rows_per
    int world, rank;
    int64_t n_local;          // hi - lo
    int cap;                  // request slots per peer
    uint32_t hmask;           // hash table size - 1 (power of two)
    int32_t* hkey;            // [H] id + 1, 0 = empty
    int32_t* hval;            // [H] scratch slot of the id
    int32_t* send_ids;        // [world * cap] row index AT THE OWNER of each requested id, -1 = unused
    int32_t* counts;          // [world] requests per peer; counts[world] = overflow flag
};

__device__ __forceinline__ uint32_t hash_id(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int64_t route_id_at(const RouteArgs& a, int64_t t) {
    // entity slot t: [0, 2b) = s / o of the positives, [2b, 2b + 2 nneg) = s / o of the corruptions
    if (t < 2 * a.b) return a.triples[3 * (t >> 1) + ((t & 1) ? 2 : 0)];
    t -= 2 * a.b;
    return a.negs[3 * (t >> 1) + ((t & 1) ? 2 : 0)];
}

// pass 1: first writer of a remote id claims a request slot at its owner.  Slots are handed out per WAVE: with two ranks every
// remote id has the same owner, and one returning atomic per id on that owner's counter serialised at the L2 (measured 92 us
// for the 16 384 ids of a C4 step); a wave now counts its claims per owner with a ballot and its first lane takes them all.
__global__ __launch_bounds__(256) void route_insert_kernel(RouteArgs a) {
    const int64_t n = 2 * (a.b + a.nneg);
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < n; t0 += stride) {   // (wave-uniform trip count: ballots below)
        const int64_t t = t0 + threadIdx.x;
        int64_t id = -1;
        bool claimed = false;
        uint32_t h = 0;
        if (t < n) {
            id = route_id_at(a, t);
            if (id < a.lo || id >= a.hi) {
                h = hash_id((uint32_t)id) & a.hmask;
                const int32_t key = (int32_t)id + 1;
                for (;;) {
                    const int32_t prev = atomicCAS(a.hkey + h, 0, key);
                    if (prev == 0) { claimed = true; break; }
                    if (prev == key) break;   // somebody else owns the entry
                    h = (h + 1) & a.hmask;
                }
            }
        }
        const int owner = claimed ? (int)(id / a.rows_per) : -1;
        for (int q = 0; q < a.world; ++q) {
            const unsigned long long m = __ballot(owner == q);
            if (!m) continue;
            int base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(a.counts + q, __popcll(m));
            base = __shfl(base, __builtin_ctzll(m), 64);
            if (owner == q) {
                const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                int slot;
                if (pos < a.cap) {
                    slot = q * a.cap + pos;
                    a.send_ids[slot] = (int32_t)(id - (int64_t)q * a.rows_per);
                } else {       // this peer's list is full: flag it (the host raises), keep indices in range
                    atomicExch(a.counts + a.world, 1);
                    slot = q * a.cap;
                }
                a.hval[h] = slot;
            }
        }
    }
}

// pass 2 (separate launch: every hval is written): rewrite the batch into the local index space
__global__ __launch_bounds__(256) void route_rewrite_kernel(RouteArgs a) {
    const int64_t n = 2 * (a.b + a.nneg);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = route_id_at(a, t);
        int32_t loc;
        if (id >= a.lo && id < a.hi) {
            loc = (int32_t)(id - a.lo);
        } else {
            uint32_t h = hash_id((uint32_t)id) & a.hmask;
            const int32_t key = (int32_t)id + 1;
            while (a.hkey[h] != key) h = (h + 1) & a.hmask;
            loc = (int32_t)a.n_local + a.hval[h];
        }
        const bool is_neg = t >= 2 * a.b;
        const int64_t tt = is_neg ? t - 2 * a.b : t;
        int32_t* dst = (is_neg ? a.out_negs : a.out_triples) + 3 * (tt >> 1);
        const int32_t* src = (is_neg ? a.negs : a.triples) + 3 * (tt >> 1);
        dst[(tt & 1) ? 2 : 0] = loc;
        if (!(tt & 1)) dst[1] = src[1];   // the relation id travels unchanged
    }
}

// rows[idx[j]] -> out[j] (idx < 0: a zero row); one wave per row, 16-byte accesses
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int K, const int32_t* __restrict__ idx,
                                                          int64_t n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += (int64_t)gridDim.x * 4) {
        const int32_t r = idx[j];
        const float4* src = reinterpret_cast<const float4*>(table + (int64_t)(r < 0 ? 0 : r) * K);
        float4* dst = reinterpret_cast<float4*>(out + j * K);
        for (int q = lane; q < (K >> 2); q += 64) dst[q] = r < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : src[q];
    }
}

// table[idx[j]] += src[j] (idx < 0: skipped); lane-contiguous fp32 atomics (whole 128-byte lines per wave instruction)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ table, int K, const int32_t* __restrict__ idx, int64_t n,
                                                               const float* __restrict__ src) {
    const int lane = threadIdx.x & 63;
    for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += (int64_t)gridDim.x * 4) {
        const int32_t r = idx[j];
        if (r < 0) continue;
        float* dst = table + (int64_t)r * K;
        const float* s = src + j * K;
        for (int c = lane; c < K; c += 64) {
            const float v = s[c];
            if (v != 0.f) atomic_add_f32(dst + c, v);
        }
    }
}

// data-parallel merge: gradient of element i = sum over the W received partial slices, then the ordinary update
template <int KIND>
__global__ __launch_bounds__(256) void opt_merged_kernel(OptArgs a, const float* __restrict__ parts, int nparts, int64_t part_stride) {
    const int64_t n4 = a.n >> 2;
    float reg_acc = 0.f;
    float4* x4 = reinterpret_cast<float4*>(a.x);
    float4* m4 = reinterpret_cast<float4*>(a.s0);
    float4* v4 = reinterpret_cast<float4*>(a.s1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 g = reinterpret_cast<const float4*>(parts)[i];
        for (int q = 1; q < nparts; ++q) {   // rank order: the same sum on every run
            const float4 t = reinterpret_cast<const float4*>(parts + q * part_stride)[i];
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
        }
        float4 x = x4[i];
        float4 m = make_float4(0, 0, 0, 0), v = make_float4(0, 0, 0, 0);
        if constexpr (opt_nslots(KIND) >= 1) m = m4[i];
        if constexpr (opt_nslots(KIND) == 2) v = v4[i];
        opt_elem<KIND>(a, x.x, g.x, m.x, v.x, reg_acc);
        opt_elem<KIND>(a, x.y, g.y, m.y, v.y, reg_acc);
        opt_elem<KIND>(a, x.z, g.z, m.z, v.z, reg_acc);
        opt_elem<KIND>(a, x.w, g.w, m.w, v.w, reg_acc);
        x4[i] = x;
        if constexpr (opt_nslots(KIND) >= 1) m4[i] = m;
        if constexpr (opt_nslots(KIND) == 2) v4[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {   // n % 4 tail
        float g = parts[i];
        for (int q = 1; q < nparts; ++q) g += parts[q * part_stride + i];
        float x = a.x[i], m = 0.f, v = 0.f;
        if constexpr (opt_nslots(KIND) >= 1) m = a.s0[i];
        if constexpr (opt_nslots(KIND) == 2) v = a.s1[i];
        opt_elem<KIND>(a, x, g, m, v, reg_acc);
        a.x[i] = x;
        if constexpr (opt_nslots(KIND) >= 1) a.s0[i] = m;
        if constexpr (opt_nslots(KIND) == 2) a.s1[i] = v;
    }
    if (a.reg_loss && a.lam != 0.f) {
        __shared__ float red[4];
        const float w = wave_sum(reg_acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.reg_loss, (double)a.lam * ((double)red[0] + red[1] + red[2] + red[3]));
    }
}

// synthetic triple stream (synth-50M): triple `row` = Philox4x32-10(counter = row, key = seed)
__global__ void synth_triples_kernel(uint64_t seed, int64_t first_row, int64_t n, uint32_t n_ents, uint32_t n_rels, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t row = (uint64_t)(first_row + i);
    const u32x4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), 0x53594e54u /* "SYNT" */, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[3 * i + 0] = (int32_t)__umulhi(r.x, n_ents);
    out[3 * i + 1] = (int32_t)__umulhi(r.y, n_rels);
    out[3 * i + 2] = (int32_t)__umulhi(r.z, n_ents);
}

static inline uint32_t route_table_size(int64_t n_ids) {
    uint32_t h = 1024;
    while ((int64_t)h < 2 * n_ids && h < (1u << 30)) h <<= 1;
    return h;
}

}  // namespace kge

using namespace kge;

extern "C" int64_t amdkge_shard_route_workspace_bytes(int64_t b, int64_t nneg) {
    if (b < 0 || nneg < 0 || 2 * (b + nneg) >= (1ll << 29)) return -1;
    return (int64_t)route_table_size(2 * (b + nneg)) * 8 + 512;
}

extern "C" int amdkge_shard_route(int64_t n_ents, int32_t world, int32_t rank, const int32_t* d_triples, int64_t b,
                                  const int32_t* d_negs, int64_t nneg, int32_t cap, int32_t* d_out_triples, int32_t* d_out_negs,
                                  int32_t* d_send_ids, int32_t* d_counts, void* d_work, void* stream) {
    if (n_ents <= 0 || n_ents > 0x7FFFFFFFll || world < 1 || rank < 0 || rank >= world) return set_error(AMDKGE_EINVAL, "shard_route: bad shard geometry");
    if (b < 0 || nneg < 0 || cap < 1 || (int64_t)world * cap > 0x7FFFFFFFll) return set_error(AMDKGE_EINVAL, "shard_route: bad sizes");
    if (!d_send_ids || !d_counts || !d_work || (b > 0 && (!d_triples || !d_out_triples)) || (nneg > 0 && (!d_negs || !d_out_negs)))
        return set_error(AMDKGE_EINVAL, "shard_route: NULL pointer");
    const int64_t ws = amdkge_shard_route_workspace_bytes(b, nneg);
    if (ws < 0) return set_error(AMDKGE_EUNSUPPORTED, "shard_route: batch too large for one call");
    hipStream_t st = (hipStream_t)stream;
    RouteArgs a{};
    a.triples = d_triples; a.negs = d_negs; a.out_triples = d_out_triples; a.out_negs = d_out_negs; a.b = b; a.nneg = nneg;
    a.rows_per = (n_ents + world - 1) / world;
    a.lo = a.rows_per * rank < n_ents ? a.rows_per * rank : n_ents;
    a.hi = a.rows_per * (rank + 1) < n_ents ? a.rows_per * (rank + 1) : n_ents;
    a.n_local = a.hi - a.lo; a.world = world; a.rank = rank; a.cap = cap;
    const uint32_t H = route_table_size(2 * (b + nneg));
    a.hmask = H - 1;
    char* w = (char*)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    a.hkey = (int32_t*)w; a.hval = (int32_t*)(w + (size_t)H * 4);
    a.send_ids = d_send_ids; a.counts = d_counts;
    if (hipError_t e = hipMemsetAsync(a.hkey, 0, (size_t)H * 4, st)) return set_error_hip(e, "hipMemsetAsync(route table)");
    // (d_counts[world], the overflow flag, is sticky: only the caller clears it)
    if (hipError_t e = hipMemsetAsync(d_counts, 0, (size_t)world * 4, st)) return set_error_hip(e, "hipMemsetAsync(route counts)");
    if (hipError_t e = hipMemsetAsync(d_send_ids, 0xFF, (size_t)world * cap * 4, st)) return set_error_hip(e, "hipMemsetAsync(request lists)");
    const int64_t n = 2 * (b + nneg);
    if (n == 0) return AMDKGE_OK;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(route_insert_kernel, dim3(grid), dim3(256), 0, st, a);
    if (int rc = check_launch("shard_route(insert)")) return rc;
    hipLaunchKernelGGL(route_rewrite_kernel, dim3(grid), dim3(256), 0, st, a);
    return check_launch("shard_route(rewrite)");
}

extern "C" int amdkge_gather_rows(const float* d_table, int32_t row_floats, const int32_t* d_idx, int64_t n, float* d_out, void* stream) {
    if (n < 0 || row_floats < 4 || row_floats % 4 != 0) return set_error(AMDKGE_EINVAL, "gather_rows: row_floats must be a positive multiple of 4 (stored layout)");
    if (n == 0) return AMDKGE_OK;
    if (!d_table || !d_idx || !d_out) return set_error(AMDKGE_EINVAL, "gather_rows: NULL pointer");
    const unsigned grid = (unsigned)((n + 3) / 4 < 16384 ? (n + 3) / 4 : 16384);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_table, (int)row_floats, d_idx, n, d_out);
    return check_launch("gather_rows");
}

extern "C" int amdkge_scatter_add_rows(float* d_table, int32_t row_floats, const int32_t* d_idx, int64_t n, const float* d_src, void* stream) {
    if (n < 0 || row_floats < 1) return set_error(AMDKGE_EINVAL, "scatter_add_rows: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_table || !d_idx || !d_src) return set_error(AMDKGE_EINVAL, "scatter_add_rows: NULL pointer");
    const unsigned grid = (unsigned)((n + 3) / 4 < 16384 ? (n + 3) / 4 : 16384);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_table, (int)row_floats, d_idx, n, d_src);
    return check_launch("scatter_add_rows");
}

extern "C" int amdkge_opt_step_merged(const amdkge_opt* opt, float* d_x, const float* d_grad_parts, int32_t n_parts, int64_t part_stride,
                                      float* d_slot0, float* d_slot1, int64_t n_elems, double* d_reg_loss, void* stream) {
    if (int rc = validate_opt(opt)) return rc;
    if (opt->lazy) return set_error(AMDKGE_EUNSUPPORTED, "opt_step_merged: the touched-rows mode sweeps whole rows; use amdkge_opt_step on the all-reduced gradient");
    if (n_elems < 0 || n_parts < 1 || part_stride < n_elems) return set_error(AMDKGE_EINVAL, "opt_step_merged: bad sizes");
    if (n_elems == 0) return AMDKGE_OK;
    if (part_stride % 4 != 0) return set_error(AMDKGE_EINVAL, "opt_step_merged: part_stride must be a multiple of 4 floats");
    if (!d_x || !d_grad_parts) return set_error(AMDKGE_EINVAL, "opt_step_merged: NULL pointer");
    if (opt_nslots(opt->kind) >= 1 && !d_slot0) return set_error(AMDKGE_EINVAL, "opt_step_merged: optimizer slot 0 is NULL");
    if (opt_nslots(opt->kind) == 2 && !d_slot1) return set_error(AMDKGE_EINVAL, "opt_step_merged: optimizer slot 1 is NULL");
    if ((((uintptr_t)d_x | (uintptr_t)d_grad_parts | (uintptr_t)d_slot0 | (uintptr_t)d_slot1) & 15) != 0)
        return set_error(AMDKGE_EINVAL, "opt_step_merged: buffers must be 16-byte aligned");
    OptArgs a{};
    a.x = d_x; a.g = nullptr; a.s0 = d_slot0; a.s1 = d_slot1; a.n = n_elems; a.reg_loss = d_reg_loss;
    fill_opt_args(a, opt);
    unsigned grid = (unsigned)(((n_elems + 3) / 4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipStream_t st = (hipStream_t)stream;
#define KGE_OPT_MERGED(KIND) hipLaunchKernelGGL(opt_merged_kernel<KIND>, dim3(grid), dim3(256), 0, st, a, d_grad_parts, (int)n_parts, part_stride)
    KGE_OPT_DISPATCH(opt->kind, KGE_OPT_MERGED)
#undef KGE_OPT_MERGED
    return check_launch("opt_step_merged");
}

extern "C" int amdkge_synth_triples(uint64_t seed, int64_t first_row, int64_t n, int64_t n_ents, int64_t n_rels, int32_t* d_out, void* stream) {
    if (n < 0 || n_ents <= 0 || n_ents > 0x7FFFFFFFll || n_rels <= 0 || n_rels > 0x7FFFFFFFll) return set_error(AMDKGE_EINVAL, "synth_triples: bad sizes");
    if (n == 0) return AMDKGE_OK;
    if (!d_out) return set_error(AMDKGE_EINVAL, "synth_triples: NULL output");
    hipLaunchKernelGGL(synth_triples_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, first_row, n,
                       (uint32_t)n_ents, (uint32_t)n_rels, d_out);
    return check_launch("synth_triples");
}
