// Session GROUP: the session layer of the C ABI on several GPUs of one node, SINGLE PROCESS, multi device -- the shape SURVEY.md
// 8(b) specified for a host without torch ("the library owns ... the RCCL communicators"): one amdkge_session (replicated
// tables, optimizer state, scratch, stream) per device, the collectives issued by the library itself.
//
// The reference has no multi-device path (/root/reference: no tf.distribute / NCCL / Horovod call site); what a group step
// replaces is ScoringBasedEmbeddingModel.train_step (ScoringBasedEmbeddingModel.py:370-429) on ONE global batch:
//   1. replica d takes the contiguous share [lo_d, hi_d) of the batch and runs the fused kernels in their GRADIENT-ONLY form;
//      negatives are keyed by the GLOBAL corruption row (row_offset = lo_d, b_global = B), so n replicas at B / n draw exactly
//      the corruptions one GPU draws at B;
//   2. the dense gradients of both tables are summed over the replicas: ncclAllReduce on every replica's stream inside one
//      ncclGroupStart / ncclGroupEnd (RCCL over xGMI; librccl is bound at run time with dlopen -- a process that never creates
//      a multi-device group never loads it, and no second copy of the library is mapped beside torch's);
//   3. every replica applies the same dense optimizer sweep to its copy: the replicas stay bit-identical.
// Replicas on ONE device (devices = {d, d, ...}: RCCL refuses duplicate devices) sum their gradients with a plain kernel
// instead -- the whole group path except the three RCCL calls, which is what the one-GPU development box can test.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "kge_group_staging.h"
#include "kge_session_impl.h"

using namespace kge;

namespace {

#define KGE_HIP(call, what) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return set_error_hip(e_, what); } while (0)
#define KGE_RC(call) do { const int rc_ = (call); if (rc_ != AMDKGE_OK) return rc_; } while (0)

// the slice of rccl.h this file needs (ABI of RCCL 2.x: ncclResult_t / ncclDataType_t / ncclRedOp_t are plain enums)
typedef void* ncclComm_t;
enum { kNcclSuccess = 0, kNcclInt32 = 2, kNcclFloat = 7, kNcclSum = 0 };
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

int rccl_error(const Rccl& r, int code, const char* where) {
    static thread_local char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", where, r.GetErrorString ? r.GetErrorString(code) : "RCCL error");
    return set_error(AMDKGE_ERCCL, msg);
}

int load_rccl(Rccl& r) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.handle) return set_error(AMDKGE_ERCCL, "session_group: librccl.so not found (dlopen)");
#define KGE_SYM(field, sym) do { *(void**)(&r.field) = dlsym(r.handle, sym); if (!r.field) return set_error(AMDKGE_ERCCL, "session_group: librccl lacks " sym); } while (0)
    KGE_SYM(CommInitAll, "ncclCommInitAll"); KGE_SYM(CommDestroy, "ncclCommDestroy"); KGE_SYM(AllReduce, "ncclAllReduce");
    KGE_SYM(GroupStart, "ncclGroupStart"); KGE_SYM(GroupEnd, "ncclGroupEnd"); KGE_SYM(GetErrorString, "ncclGetErrorString");
    KGE_SYM(GetVersion, "ncclGetVersion"); KGE_SYM(Send, "ncclSend"); KGE_SYM(Recv, "ncclRecv");
#undef KGE_SYM
    return AMDKGE_OK;
}

// same-device replicas: dst[0] = sum of all, then every replica receives the sum (grid-stride, float4 body + tail)
struct SumArgs { float* p[16]; int n; int64_t len; };
__global__ void replica_sum_kernel(SumArgs a) {
    const int64_t n4 = a.len >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 t = reinterpret_cast<const float4*>(a.p[0])[i];
        for (int q = 1; q < a.n; ++q) { const float4 u = reinterpret_cast<const float4*>(a.p[q])[i]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        for (int q = 0; q < a.n; ++q) reinterpret_cast<float4*>(a.p[q])[i] = t;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.len; i += (int64_t)gridDim.x * blockDim.x) {
        float t = a.p[0][i];
        for (int q = 1; q < a.n; ++q) t += a.p[q][i];
        for (int q = 0; q < a.n; ++q) a.p[q][i] = t;
    }
}

}  // namespace

struct amdkge_session_group {
    std::vector<amdkge_session*> rep;
    std::vector<ncclComm_t> comm;   // empty: single replica, or all replicas on one device (local sum)
    std::vector<hipEvent_t> ev;     // per replica: its share of the step is enqueued / done (same-device sum)
    Rccl rccl;
    bool same_device = false;
    int32_t flags = 0;              // AMDKGE_GROUP_*
    // ---- AMDKGE_GROUP_ROWS: the entity table row-sharded over the replicas (see the ROWS section below) ----
    bool rows = false;
    int64_t N = 0, rows_per = 0, cap = 0, max_share = 0;   // global entity count, rows per shard, request slots per peer, positives per replica and step at most
    uint64_t step = 0;
    int64_t iteration = 0;
    struct Shard {
        int64_t lo = 0, n_local = 0;
        int32_t *tri = nullptr, *negs = nullptr, *xl = nullptr, *nl = nullptr, *send = nullptr, *recv = nullptr, *counts = nullptr;
        void* route_work = nullptr;
        float *rows_out = nullptr, *back = nullptr;
        double* acc3 = nullptr;     // [data loss, entity-shard regulariser, relation regulariser]
    };
    std::vector<Shard> sh;
    // ---- AMDKGE_GROUP_COLS: every table COLUMN-sharded over the replicas (see the COLS section below) ----
    bool cols = false;
    int k_full = 0;                 // units per half of the whole model; replica d holds units [d k_full / W, (d + 1) k_full / W)
    bool poisoned = false;          // a column-sharded step failed after some slices had applied their update: the slices of ONE model
                                    // have parted (see cols_train_step); every later step is refused until the whole entity table is written again (set_rows)
};

extern "C" void amdkge_session_group_destroy(amdkge_session_group* g) {
    if (!g) return;
    for (size_t d = 0; d < g->sh.size() && d < g->rep.size(); ++d) {
        (void)hipSetDevice(g->rep[d]->cfg.device);
        if (g->rep[d]->st) (void)hipStreamSynchronize(g->rep[d]->st);
        amdkge_session_group::Shard& h = g->sh[d];
        for (void* p : {(void*)h.tri, (void*)h.negs, (void*)h.xl, (void*)h.nl, (void*)h.send, (void*)h.recv, (void*)h.counts, h.route_work,
                        (void*)h.rows_out, (void*)h.back, (void*)h.acc3})
            if (p) (void)hipFree(p);
    }
    for (size_t d = 0; d < g->comm.size(); ++d)
        if (g->comm[d] && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(g->comm[d]);
    for (size_t d = 0; d < g->ev.size(); ++d)
        if (g->ev[d]) { (void)hipSetDevice(g->rep[d]->cfg.device); (void)hipEventDestroy(g->ev[d]); }
    for (amdkge_session* s : g->rep) amdkge_session_destroy(s);
    if (g->rccl.handle) (void)dlclose(g->rccl.handle);
    delete g;
}

extern "C" int amdkge_session_group_create(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, amdkge_session_group** out) try {
    return amdkge_session_group_create_ex(cfg, devices, n_gpus, 0, out);
} KGE_CATCH("session_group_create")

extern "C" int amdkge_session_group_info(const amdkge_session_group* g, int32_t* uses_rccl, int32_t* rccl_version) try {
    if (!g) return set_error(AMDKGE_EINVAL, "session_group_info: NULL group");
    if (uses_rccl) *uses_rccl = g->comm.empty() ? 0 : 1;
    if (rccl_version) {
        int v = 0;
        if (g->rccl.GetVersion) (void)g->rccl.GetVersion(&v);
        *rccl_version = v;
    }
    return AMDKGE_OK;
} KGE_CATCH("session_group_info")

// replicas (one session per entry of `devices`, each with `n_ents_alloc` entity rows), events, and the RCCL communicators when the
// replicas sit on distinct devices (or one replica is forced through RCCL)
static int group_create(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags, int64_t n_ents_alloc,
                        amdkge_session_group** out) {
    amdkge_session_group* g = new amdkge_session_group();
    g->flags = flags;
    auto fail = [&](int rc) { amdkge_session_group_destroy(g); return rc; };
    bool same = true, distinct = true;
    for (int d = 0; d < n_gpus; ++d) {
        amdkge_session_config c = *cfg;
        c.device = devices ? devices[d] : d;
        c.model.n_ents = n_ents_alloc;
        if (d > 0 && c.device != g->rep[0]->cfg.device) same = false;
        for (int q = 0; q < d; ++q) if (g->rep[q]->cfg.device == c.device) distinct = false;
        amdkge_session* s = nullptr;
        if (int rc = amdkge_session_create(&c, &s)) return fail(rc);
        g->rep.push_back(s);
    }
    if (n_gpus > 1 && !same && !distinct) return fail(set_error(AMDKGE_EINVAL, "session_group_create: devices must be all distinct (RCCL) or all the same (local sum)"));
    g->same_device = n_gpus > 1 && same;
    g->ev.assign((size_t)n_gpus, nullptr);
    for (int d = 0; d < n_gpus; ++d) {
        if (hipError_t e = hipSetDevice(g->rep[d]->cfg.device)) return fail(set_error_hip(e, "hipSetDevice"));
        if (hipError_t e = hipEventCreateWithFlags(&g->ev[d], hipEventDisableTiming)) return fail(set_error_hip(e, "hipEventCreate"));
    }
    if ((n_gpus > 1 && !g->same_device) || (n_gpus == 1 && (flags & AMDKGE_GROUP_FORCE_RCCL))) {
        if (int rc = load_rccl(g->rccl)) return fail(rc);
        std::vector<int> devs;
        for (amdkge_session* s : g->rep) devs.push_back(s->cfg.device);
        g->comm.assign((size_t)n_gpus, nullptr);
        if (int rc = g->rccl.CommInitAll(g->comm.data(), n_gpus, devs.data())) return fail(rccl_error(g->rccl, rc, "ncclCommInitAll"));
    }
    *out = g;
    return AMDKGE_OK;
}

extern "C" int amdkge_session_group_create_ex(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                              amdkge_session_group** out) try {
    if (!cfg || !out || n_gpus < 1 || n_gpus > 16) return set_error(AMDKGE_EINVAL, "session_group_create: bad arguments (1 <= n_gpus <= 16)");
    if (flags & ~(AMDKGE_GROUP_FORCE_RCCL | AMDKGE_GROUP_FORCE_THREADS)) return set_error(AMDKGE_EINVAL, "session_group_create_ex: unknown flag (row sharding: amdkge_session_group_create_rows)");
    *out = nullptr;
    return group_create(cfg, devices, n_gpus, flags, cfg->model.n_ents, out);
} KGE_CATCH("session_group_create_ex")

extern "C" int32_t amdkge_session_group_size(const amdkge_session_group* g) { return g ? (int32_t)g->rep.size() : 0; }

extern "C" int amdkge_session_group_replica(amdkge_session_group* g, int32_t i, amdkge_session** out) try {
    if (!g || !out || i < 0 || i >= (int32_t)g->rep.size()) return set_error(AMDKGE_EINVAL, "session_group_replica: no such replica");
    *out = g->rep[(size_t)i];
    return AMDKGE_OK;
} KGE_CATCH("session_group_replica")

static int rows_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host);
static int rows_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out);
static int cols_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host);
static int cols_get_rows(amdkge_session_group* g, int32_t table, const int32_t* ids, int64_t row0, int64_t nrows, float* host);
static int cols_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out);

extern "C" int amdkge_session_group_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host) try {
    if (!g) return set_error(AMDKGE_EINVAL, "session_group_set_rows: NULL group");
    if (g->rows) return rows_set_rows(g, table, row0, nrows, host);
    if (g->cols) return cols_set_rows(g, table, row0, nrows, host);
    for (amdkge_session* s : g->rep) KGE_RC(amdkge_session_set_rows(s, table, row0, nrows, host));
    return AMDKGE_OK;
} KGE_CATCH("session_group_set_rows")

// sum `len` floats at `ptr_of(replica)` over the replicas, result on every replica, stream-ordered on each replica's stream
static int group_sum(amdkge_session_group* g, float* (*ptr_of)(amdkge_session*), int64_t len) {
    const int n = (int)g->rep.size();
    if ((n == 1 && g->comm.empty()) || len == 0) return AMDKGE_OK;
    if (g->same_device) {
        // replica 0's stream waits for the others' work, sums in place for everybody, and the others wait for the sum
        amdkge_session* s0 = g->rep[0];
        KGE_HIP(hipSetDevice(s0->cfg.device), "hipSetDevice");
        for (int d = 1; d < n; ++d) {
            KGE_HIP(hipEventRecord(g->ev[d], g->rep[d]->st), "hipEventRecord");
            KGE_HIP(hipStreamWaitEvent(s0->st, g->ev[d], 0), "hipStreamWaitEvent");
        }
        SumArgs a{};
        a.n = n; a.len = len;
        for (int d = 0; d < n; ++d) a.p[d] = ptr_of(g->rep[d]);
        const int64_t n4 = (len + 3) / 4;
        const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(replica_sum_kernel, dim3(grid), dim3(256), 0, s0->st, a);
        KGE_RC(check_launch("replica_sum"));
        KGE_HIP(hipEventRecord(g->ev[0], s0->st), "hipEventRecord");
        for (int d = 1; d < n; ++d) KGE_HIP(hipStreamWaitEvent(g->rep[d]->st, g->ev[0], 0), "hipStreamWaitEvent");
        return AMDKGE_OK;
    }
    if (int rc = g->rccl.GroupStart()) return rccl_error(g->rccl, rc, "ncclGroupStart");
    for (int d = 0; d < n; ++d) {
        float* p = ptr_of(g->rep[d]);
        if (int rc = g->rccl.AllReduce(p, p, (size_t)len, kNcclFloat, kNcclSum, g->comm[(size_t)d], g->rep[d]->st)) {
            (void)g->rccl.GroupEnd();
            return rccl_error(g->rccl, rc, "ncclAllReduce");
        }
    }
    if (int rc = g->rccl.GroupEnd()) return rccl_error(g->rccl, rc, "ncclGroupEnd");
    return AMDKGE_OK;
}

extern "C" int amdkge_session_group_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out) try {
    if (!g || B < 0) return set_error(AMDKGE_EINVAL, "session_group_train_step: bad arguments");
    if (g->rows) return rows_train_step(g, triples, B, focus_w, loss_out);
    if (g->cols) return cols_train_step(g, triples, B, focus_w, loss_out);
    const int n = (int)g->rep.size();
    if (n == 1 && g->comm.empty()) return amdkge_session_train_step(g->rep[0], triples, B, focus_w, loss_out);   // the complete fused step
    if (loss_out) *loss_out = 0.0;
    if (B == 0) return AMDKGE_OK;
    if (!triples) return set_error(AMDKGE_EINVAL, "session_group_train_step: NULL triples");
    // ---- 1. every replica: its share of the batch, gradients only ----
    for (int d = 0; d < n; ++d) {
        const int64_t lo = B * d / n, hi = B * (d + 1) / n;
        if (const int rc = amdkge_session_grad_step(g->rep[d], triples + 3 * lo, hi - lo, focus_w ? focus_w + lo : nullptr, lo, B)) {
            // replicas 0 .. d-1 already hold this batch's gradients: clear them, or the next call would add to them (their loss
            // accumulators are reset by every grad_step)
            for (int q = 0; q < d; ++q) {
                amdkge_session* s = g->rep[q];
                (void)hipSetDevice(s->cfg.device);
                (void)hipMemsetAsync(s->g_ent, 0, (size_t)s->cfg.model.n_ents * s->Ks * sizeof(float), s->st);
                (void)hipMemsetAsync(s->g_rel, 0, (size_t)s->cfg.model.n_rels * s->Ks * sizeof(float), s->st);
            }
            return rc;
        }
    }
    // ---- 2. gradient sum over the replicas (RCCL all-reduce, or the local sum of same-device replicas) ----
    KGE_RC(group_sum(g, [](amdkge_session* s) { return s->g_ent; }, g->rep[0]->cfg.model.n_ents * (int64_t)g->rep[0]->Ks));
    KGE_RC(group_sum(g, [](amdkge_session* s) { return s->g_rel; }, g->rep[0]->cfg.model.n_rels * (int64_t)g->rep[0]->Ks));
    // ---- 3. every replica: the same dense sweep; data loss = sum of the shares, regulariser terms from replica 0 ----
    double data = 0.0, reg = 0.0;
    for (int d = 0; d < n; ++d) KGE_RC(amdkge_session_apply_step(g->rep[d]));
    for (int d = 0; d < n; ++d) {
        double h[2];
        KGE_RC(amdkge_session_finish_step(g->rep[d], h));
        data += h[0];
        if (d == 0) reg = h[1];
    }
    if (loss_out) *loss_out = data + reg;
    // deterministic mode: a tile beyond its sort buffer fell back to arrival order -- reported like amdkge_session_train_step does
    if (g->rep[0]->cfg.flags & AMDKGE_TILED_DETERMINISTIC) {
        bool fell_back = false;
        for (int d = 0; d < n; ++d) {
            amdkge_session* s = g->rep[d];
            const int64_t lo = B * d / n, hi = B * (d + 1) / n;
            if (hi == lo || !s->twork) continue;
            KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
            int32_t st_flag = 0;
            KGE_RC(amdkge_train_tiled_status(&s->cfg.model, hi - lo, s->cfg.eta, s->cfg.flags & (AMDKGE_TILED_POS_ATOMIC | AMDKGE_TILED_DETERMINISTIC),
                                             s->twork, &st_flag, s->st));
            fell_back = fell_back || st_flag != 0;
        }
        if (fell_back)
            return set_error(AMDKGE_EUNSUPPORTED, "session_group_train_step: deterministic mode -- a tile received more entries than its sort buffer holds (a very hot row); this step's sums were not all added in canonical order");
    }
    return AMDKGE_OK;
} KGE_CATCH("session_group_train_step")

// =====================================================================================================================================
// AMDKGE_GROUP_ROWS -- the entity table ROW-SHARDED over the replicas of a group (BASELINE.json configs[3], configs[4]; the C-ABI
// form of ampligraph_amd/sharded.py ShardedStepLoop, for a host without torch).  Replica d owns the contiguous id range
// [d * rows_per, min(N, (d + 1) * rows_per)), rows_per = ceil(N / W) -- the reference's own bucket rule, owner(e) = e // ceil(N / G)
// (datasets/graph_partitioner.py:339-344) -- with the optimizer state of those rows; the relation table is replicated.  What a
// step replaces is the reference's partition loop (ScoringBasedEmbeddingModel.py:227,259-261 for training), with the partitions
// living on different GPUs:
//   1. replica d takes the share [B d / W, B (d + 1) / W) of the global batch; amdkge_shard_route classifies every s / o id of the
//      share (and, with global negatives, of its corruptions) by owner ON THE DEVICE, de-duplicates remote ids and gives each a
//      request slot at its owner (cap slots per peer: equal, host-known splits -- nothing is copied to the host);
//   2. the request lists are exchanged (grouped ncclSend / ncclRecv, one slice per peer: every xGMI link carries one slice at once),
//      owners gather the rows (amdkge_gather_rows), the rows travel back and land BEHIND the shard in the same allocation, where
//      the re-indexed batch finds them: the fused train kernels run unchanged on "a table of n_local + W cap rows";
//   3. the kernels run in their gradient-only form; gradient rows of the fetched copies go home by the reverse route and are added
//      at the owner (amdkge_scatter_add_rows); the relation gradient is all-reduced (ncclAllReduce);
//   4. every replica sweeps ITS rows and the replicated relation table.
// Negatives: shard-local by default (replacement ids from the replica's own range -- what the reference's partitioned training
// does; only <= 2 remote rows per positive move); AMDKGE_GROUP_GLOBAL_NEGATIVES reproduces one GPU's corruptions exactly (the same
// Philox rows over all N ids), which is what the parity test compares with a single session.  Replicas on ONE device exchange by
// device-to-device copies ordered with events -- everything but the RCCL calls, for the one-GPU box.
namespace {

typedef amdkge_session_group::Shard Shard;

int64_t table_rows_global(const amdkge_session_group* g, int t) {
    return (t == AMDKGE_TABLE_ENT || t == AMDKGE_TABLE_ENT_SLOT0 || t == AMDKGE_TABLE_ENT_SLOT1) ? g->N : g->rep[0]->cfg.model.n_rels;
}
bool is_entity_table(int t) { return t == AMDKGE_TABLE_ENT || t == AMDKGE_TABLE_ENT_SLOT0 || t == AMDKGE_TABLE_ENT_SLOT1; }

// all_to_all with equal splits: slice q of `send(d)` (elems elements of `bytes_per` bytes) -> slice d of `recv(q)`, stream-ordered on
// every replica's stream.  The self slice is a device-to-device copy; the others go through RCCL or, for same-device replicas, copies
// on the RECEIVER's stream behind an event of the sender's stream.
template <class FS, class FR>
int exchange(amdkge_session_group* g, FS send, FR recv, int64_t elems, int bytes_per, int nccl_type) {
    const int W = (int)g->rep.size();
    const size_t slice = (size_t)elems * bytes_per;
    if (slice == 0) return AMDKGE_OK;
    for (int d = 0; d < W; ++d) {
        KGE_HIP(hipSetDevice(g->rep[d]->cfg.device), "hipSetDevice");
        KGE_HIP(hipMemcpyAsync((char*)recv(d) + (size_t)d * slice, (const char*)send(d) + (size_t)d * slice, slice, hipMemcpyDeviceToDevice, g->rep[d]->st),
                "hipMemcpyAsync(self slice)");
    }
    if (W == 1) return AMDKGE_OK;
    if (g->same_device) {
        for (int d = 0; d < W; ++d) KGE_HIP(hipEventRecord(g->ev[d], g->rep[d]->st), "hipEventRecord");
        for (int q = 0; q < W; ++q)
            for (int d = 0; d < W; ++d) {
                if (d == q) continue;
                KGE_HIP(hipStreamWaitEvent(g->rep[q]->st, g->ev[d], 0), "hipStreamWaitEvent");
                KGE_HIP(hipMemcpyAsync((char*)recv(q) + (size_t)d * slice, (const char*)send(d) + (size_t)q * slice, slice, hipMemcpyDeviceToDevice, g->rep[q]->st),
                        "hipMemcpyAsync(peer slice)");
            }
        // a sender must not reuse its buffer before the receivers have read it: the receivers' copies are joined back
        for (int q = 0; q < W; ++q) KGE_HIP(hipEventRecord(g->ev[q], g->rep[q]->st), "hipEventRecord");
        for (int d = 0; d < W; ++d)
            for (int q = 0; q < W; ++q)
                if (q != d) KGE_HIP(hipStreamWaitEvent(g->rep[d]->st, g->ev[q], 0), "hipStreamWaitEvent");
        return AMDKGE_OK;
    }
    if (int rc = g->rccl.GroupStart()) return rccl_error(g->rccl, rc, "ncclGroupStart");
    for (int d = 0; d < W; ++d)
        for (int q = 0; q < W; ++q) {
            if (q == d) continue;
            int rc = g->rccl.Send((const char*)send(d) + (size_t)q * slice, (size_t)elems, nccl_type, q, g->comm[(size_t)d], g->rep[d]->st);
            if (!rc) rc = g->rccl.Recv((char*)recv(d) + (size_t)q * slice, (size_t)elems, nccl_type, q, g->comm[(size_t)d], g->rep[d]->st);
            if (rc) { (void)g->rccl.GroupEnd(); return rccl_error(g->rccl, rc, "ncclSend/ncclRecv"); }
        }
    if (int rc = g->rccl.GroupEnd()) return rccl_error(g->rccl, rc, "ncclGroupEnd");
    return AMDKGE_OK;
}

}  // namespace

extern "C" int amdkge_session_group_create_rows(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                                int64_t max_batch, amdkge_session_group** out) try {
    if (!cfg || !out || n_gpus < 1 || n_gpus > 16) return set_error(AMDKGE_EINVAL, "session_group_create_rows: bad arguments (1 <= n_gpus <= 16)");
    if (flags & ~(AMDKGE_GROUP_FORCE_RCCL | AMDKGE_GROUP_ROWS | AMDKGE_GROUP_GLOBAL_NEGATIVES | AMDKGE_GROUP_FORCE_THREADS)) return set_error(AMDKGE_EINVAL, "session_group_create_rows: unknown flag");
    if (max_batch < 1) return set_error(AMDKGE_EINVAL, "session_group_create_rows: max_batch must be >= 1");
    if (cfg->model.n_ents < n_gpus) return set_error(AMDKGE_EINVAL, "session_group_create_rows: fewer entities than replicas");
    if (cfg->flags & AMDKGE_TILED_DETERMINISTIC) return set_error(AMDKGE_EUNSUPPORTED, "session_group_create_rows: deterministic mode is not offered for row-sharded groups");
    *out = nullptr;
    const int W = n_gpus;
    const int64_t N = cfg->model.n_ents, rows_per = (N + W - 1) / W;
    if ((int64_t)(W - 1) * rows_per >= N) return set_error(AMDKGE_EINVAL, "session_group_create_rows: the last replica would own no rows (use fewer replicas)");
    const bool global = (flags & AMDKGE_GROUP_GLOBAL_NEGATIVES) != 0;
    const int64_t share = (max_batch + W - 1) / W;
    // request slots per peer: the worst case -- every id of a replica's share remote, distinct and owned by ONE peer (ordinary for
    // first-seen ids in sequential batches), bounded by the rows a peer owns
    int64_t cap = share * (2 + (global ? cfg->eta : 0));
    if (cap > rows_per) cap = rows_per;
    if (cap < 1) cap = 1;
    if (cap > 0x7FFFFFFFll / (W > 1 ? W : 1)) return set_error(AMDKGE_EUNSUPPORTED, "session_group_create_rows: request lists too long (max_batch x eta)");
    amdkge_session_group* g = nullptr;
    if (int rc = group_create(cfg, devices, n_gpus, flags & AMDKGE_GROUP_FORCE_RCCL, rows_per + (int64_t)W * cap, &g)) return rc;
    g->flags = flags | AMDKGE_GROUP_ROWS;
    g->rows = true; g->N = N; g->rows_per = rows_per; g->cap = cap; g->max_share = share;
    g->sh.assign((size_t)W, Shard());
    auto fail = [&](int rc) { amdkge_session_group_destroy(g); return rc; };
    const int64_t nneg = global ? share * cfg->eta : 0;
    const int64_t rw = amdkge_shard_route_workspace_bytes(share, nneg);
    if (rw < 0) return fail(set_error(AMDKGE_EUNSUPPORTED, "session_group_create_rows: max_batch too large for one route call"));
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        Shard& h = g->sh[(size_t)d];
        h.lo = (int64_t)d * rows_per < N ? (int64_t)d * rows_per : N;
        const int64_t hi = (int64_t)(d + 1) * rows_per < N ? (int64_t)(d + 1) * rows_per : N;
        h.n_local = hi - h.lo;
        if (hipError_t e = hipSetDevice(s->cfg.device)) return fail(set_error_hip(e, "hipSetDevice"));
        auto alloc = [&](void** p, size_t bytes) -> int { const hipError_t e = hipMalloc(p, bytes ? bytes : 16); return e == hipSuccess ? AMDKGE_OK : set_error_hip(e, "hipMalloc(row-shard scratch)"); };
        int rc = alloc((void**)&h.tri, (size_t)share * 12);
        if (!rc) rc = alloc((void**)&h.negs, (size_t)nneg * 12);
        if (!rc) rc = alloc((void**)&h.xl, (size_t)share * 12);
        if (!rc) rc = alloc((void**)&h.nl, (size_t)nneg * 12);
        if (!rc) rc = alloc((void**)&h.send, (size_t)W * cap * 4);
        if (!rc) rc = alloc((void**)&h.recv, (size_t)W * cap * 4);
        if (!rc) rc = alloc((void**)&h.counts, (size_t)(W + 1) * 4);
        if (!rc) rc = alloc(&h.route_work, (size_t)rw);
        if (!rc) rc = alloc((void**)&h.rows_out, (size_t)W * cap * s->Ks * 4);
        if (!rc) rc = alloc((void**)&h.back, (size_t)W * cap * s->Ks * 4);
        if (!rc) rc = alloc((void**)&h.acc3, 3 * sizeof(double));
        if (rc) return fail(rc);
        if (hipError_t e = hipMemsetAsync(h.counts, 0, (size_t)(W + 1) * 4, s->st)) return fail(set_error_hip(e, "hipMemsetAsync"));
        if (hipError_t e = hipStreamSynchronize(s->st)) return fail(set_error_hip(e, "hipStreamSynchronize"));
    }
    *out = g;
    return AMDKGE_OK;
} KGE_CATCH("session_group_create_rows")

// rows [row0, row0 + nrows) of a table in GLOBAL numbering: entity tables go to their owners, relation tables to every replica
static int rows_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host) {
    if (table < 0 || table > 5) return set_error(AMDKGE_EINVAL, "session_group_set_rows: no such table");
    if (row0 < 0 || nrows < 0 || row0 + nrows > table_rows_global(g, table)) return set_error(AMDKGE_EINVAL, "session_group_set_rows: rows outside the table");
    if (nrows == 0) return AMDKGE_OK;
    if (!host) return set_error(AMDKGE_EINVAL, "session_group_set_rows: NULL host buffer");
    if (!is_entity_table(table)) {
        for (amdkge_session* s : g->rep) KGE_RC(amdkge_session_set_rows(s, table, row0, nrows, host));
        return AMDKGE_OK;
    }
    const int K = g->rep[0]->K;
    for (size_t d = 0; d < g->rep.size(); ++d) {
        const Shard& h = g->sh[d];
        const int64_t a = row0 > h.lo ? row0 : h.lo, b = (row0 + nrows) < (h.lo + h.n_local) ? (row0 + nrows) : (h.lo + h.n_local);
        if (b > a) KGE_RC(amdkge_session_set_rows(g->rep[d], table, a - h.lo, b - a, host + (a - row0) * (int64_t)K));
    }
    return AMDKGE_OK;
}

extern "C" int amdkge_session_group_get_rows(amdkge_session_group* g, int32_t table, const int32_t* ids, int64_t row0, int64_t nrows, float* host) try {
    if (!g || table < 0 || table > 5) return set_error(AMDKGE_EINVAL, "session_group_get_rows: bad arguments");
    if (nrows < 0) return set_error(AMDKGE_EINVAL, "session_group_get_rows: nrows must be >= 0");
    if (nrows == 0) return AMDKGE_OK;
    if (!host) return set_error(AMDKGE_EINVAL, "session_group_get_rows: NULL host buffer");
    if (g->cols) return cols_get_rows(g, table, ids, row0, nrows, host);
    if (!g->rows || !is_entity_table(table)) return amdkge_session_get_rows(g->rep[0], table, ids, row0, nrows, host);   // every replica holds it whole
    const int K = g->rep[0]->K;
    const int64_t rows = g->N;
    if (!ids && (row0 < 0 || row0 + nrows > rows)) return set_error(AMDKGE_EINVAL, "session_group_get_rows: rows outside the table");
    // group the wanted rows by owner, fetch each owner's list, scatter into the caller's order
    std::vector<std::vector<int32_t>> local(g->rep.size());
    std::vector<std::vector<int64_t>> where(g->rep.size());
    for (int64_t i = 0; i < nrows; ++i) {
        const int64_t id = ids ? (int64_t)ids[i] : row0 + i;
        if (id < 0 || id >= rows) return set_error(AMDKGE_EINVAL, "session_group_get_rows: row id outside the table");
        const size_t d = (size_t)(id / g->rows_per);
        local[d].push_back((int32_t)(id - g->sh[d].lo));
        where[d].push_back(i);
    }
    std::vector<float> tmp;
    for (size_t d = 0; d < g->rep.size(); ++d) {
        if (local[d].empty()) continue;
        tmp.resize(local[d].size() * (size_t)K);
        KGE_RC(amdkge_session_get_rows(g->rep[d], table, local[d].data(), 0, (int64_t)local[d].size(), tmp.data()));
        for (size_t j = 0; j < local[d].size(); ++j) memcpy(host + where[d][j] * (int64_t)K, tmp.data() + j * (size_t)K, (size_t)K * sizeof(float));
    }
    return AMDKGE_OK;
} KGE_CATCH("session_group_get_rows")

extern "C" int amdkge_session_group_route_overflow(amdkge_session_group* g, int32_t* overflowed) try {
    if (!g || !overflowed) return set_error(AMDKGE_EINVAL, "session_group_route_overflow: bad arguments");
    *overflowed = 0;
    if (!g->rows) return AMDKGE_OK;
    const int W = (int)g->rep.size();
    for (int d = 0; d < W; ++d) {
        int32_t f = 0;
        KGE_HIP(hipSetDevice(g->rep[d]->cfg.device), "hipSetDevice");
        KGE_HIP(hipMemcpyAsync(&f, g->sh[(size_t)d].counts + W, sizeof(f), hipMemcpyDeviceToHost, g->rep[d]->st), "hipMemcpyAsync(D2H)");
        KGE_HIP(hipStreamSynchronize(g->rep[d]->st), "hipStreamSynchronize");
        if (f) {
            *overflowed = 1;
            KGE_HIP(hipMemsetAsync(g->sh[(size_t)d].counts + W, 0, sizeof(int32_t), g->rep[d]->st), "hipMemsetAsync");   // reported once
        }
    }
    return AMDKGE_OK;
} KGE_CATCH("session_group_route_overflow")

static int rows_train_step_body(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out);

//  a step that fails part-way -- some replicas have accumulated g_ent / g_rel, an exchange is half enqueued -- must not
// leave gradients behind that a retried step would add to: on ANY error every replica's gradient tables (shard + scratch rows) are
// cleared and its workspace is dropped (bookkeeping may be dirty), as the replicated group step does.  The error message of the
// failed call is kept (the clean-up below reports nothing).
static int rows_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out) {
    const int rc = rows_train_step_body(g, triples, B, focus_w, loss_out);
    if (rc == AMDKGE_OK) return rc;
    const int64_t Wcap = (int64_t)g->rep.size() * g->cap;
    for (size_t d = 0; d < g->rep.size(); ++d) {
        amdkge_session* s = g->rep[d];
        (void)hipSetDevice(s->cfg.device);
        (void)hipStreamSynchronize(s->st);
        (void)hipMemsetAsync(s->g_ent, 0, (size_t)(g->sh[d].n_local + Wcap) * s->Ks * sizeof(float), s->st);
        (void)hipMemsetAsync(s->g_rel, 0, (size_t)s->cfg.model.n_rels * s->Ks * sizeof(float), s->st);
        if (s->twork) { (void)hipFree(s->twork); s->twork = nullptr; s->twork_bytes = 0; }
        (void)hipStreamSynchronize(s->st);
    }
    return rc;
}

static int rows_train_step_body(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out) {
    const int W = (int)g->rep.size();
    if (loss_out) *loss_out = 0.0;
    if (B == 0) return AMDKGE_OK;
    if (!triples) return set_error(AMDKGE_EINVAL, "session_group_train_step: NULL triples");
    if ((B + W - 1) / W > g->max_share) return set_error(AMDKGE_EINVAL, "session_group_train_step: batch larger than the max_batch the row-sharded group was created for");
    const amdkge_session_config& c0 = g->rep[0]->cfg;
    if (focus_w && !c0.loss.focus_nonlinearity) return set_error(AMDKGE_EINVAL, "session_group_train_step: FocusE weights given but the group's loss has focus_nonlinearity == 0");
    for (int64_t i = 0; i < B; ++i)
        if (triples[3 * i] < 0 || triples[3 * i] >= g->N || triples[3 * i + 2] < 0 || triples[3 * i + 2] >= g->N || triples[3 * i + 1] < 0 ||
            triples[3 * i + 1] >= c0.model.n_rels)
            return set_error(AMDKGE_EINVAL, "session_group_train_step: a triple has an entity / relation id outside the tables");
    const bool global = (g->flags & AMDKGE_GROUP_GLOBAL_NEGATIVES) != 0;
    const int eta = c0.eta;
    const int64_t cap = g->cap, Wcap = (int64_t)W * cap;
    // ---- 1. route: local index space + request lists, per replica ----
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        Shard& h = g->sh[(size_t)d];
        const int64_t lo = B * d / W, hi = B * (d + 1) / W, b = hi - lo;
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        KGE_HIP(hipMemsetAsync(h.acc3, 0, 3 * sizeof(double), s->st), "hipMemsetAsync");
        if (b > 0) KGE_HIP(hipMemcpyAsync(h.tri, triples + 3 * lo, (size_t)b * 12, hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
        const int64_t nneg = global ? b * eta : 0;
        if (global && b > 0)   // the very corruptions one GPU would draw: global Philox rows, ids over all N entities
            KGE_RC(amdkge_sample_corruptions(h.tri, b, eta, 0, g->N, c0.seed, g->step, lo, B, h.negs, s->st));
        KGE_RC(amdkge_shard_route(g->N, W, d, h.tri, b, global ? h.negs : nullptr, nneg, (int32_t)cap, h.xl, global ? h.nl : nullptr, h.send, h.counts,
                                  h.route_work, s->st));
    }
    // ---- 2. request ids to the owners, rows back behind the shards ----
    KGE_RC(exchange(g, [&](int d) { return (void*)g->sh[(size_t)d].send; }, [&](int d) { return (void*)g->sh[(size_t)d].recv; }, cap, 4, kNcclInt32));
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        KGE_RC(amdkge_gather_rows(s->tab[AMDKGE_TABLE_ENT], s->Ks, g->sh[(size_t)d].recv, Wcap, g->sh[(size_t)d].rows_out, s->st));
    }
    KGE_RC(exchange(g, [&](int d) { return (void*)g->sh[(size_t)d].rows_out; },
                    [&](int d) { return (void*)(g->rep[d]->tab[AMDKGE_TABLE_ENT] + g->sh[(size_t)d].n_local * (int64_t)g->rep[d]->Ks); },
                    cap * (int64_t)g->rep[0]->Ks, 4, kNcclFloat));
    // ---- 3. the fused kernels, gradient only, on the local index space ----
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        Shard& h = g->sh[(size_t)d];
        const int64_t lo = B * d / W, hi = B * (d + 1) / W, b = hi - lo;
        if (b == 0) continue;
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        amdkge_model m = s->cfg.model;
        m.n_ents = h.n_local + Wcap;
        amdkge_loss loss = s->cfg.loss;
        if (focus_w) {   // (FocusE weights of the share ride in the session's scratch slot 1)
            if (s->buf_bytes[1] < b * (int64_t)sizeof(float)) {
                if (s->buf[1]) KGE_HIP(hipFree(s->buf[1]), "hipFree(scratch)");
                s->buf[1] = nullptr; s->buf_bytes[1] = 0;
                KGE_HIP(hipMalloc(&s->buf[1], (size_t)b * sizeof(float)), "hipMalloc(scratch)");
                s->buf_bytes[1] = b * (int64_t)sizeof(float);
            }
            KGE_HIP(hipMemcpyAsync(s->buf[1], focus_w + lo, (size_t)b * sizeof(float), hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
            loss.d_focus_w = (const float*)s->buf[1];
        } else { loss.focus_nonlinearity = AMDKGE_FOCUS_OFF; loss.d_focus_w = nullptr; }
        amdkge_opt opt = s->cfg.opt;
        opt.iteration = g->iteration + 1;
        const int64_t srange = global ? m.n_ents : h.n_local;   // shard-local negatives: replacement rows are local rows [0, n_local)
        const int32_t* nov = global ? h.nl : nullptr;
        const int64_t need = amdkge_train_tiled_workspace_bytes(&m, b, eta);
        if (need > 0) {
            if (need > s->twork_bytes) {
                if (s->twork) KGE_HIP(hipFree(s->twork), "hipFree(twork)");
                s->twork = nullptr; s->twork_bytes = 0;
                KGE_HIP(hipMalloc(&s->twork, (size_t)need), "hipMalloc(twork)");
                KGE_HIP(hipMemsetAsync(s->twork, 0, (size_t)need, s->st), "hipMemsetAsync(twork)");
                s->twork_bytes = need;
            }
            const int rc = amdkge_train_step_tiled(&m, &loss, &opt, s->tab[0], s->tab[1], nullptr, nullptr, nullptr, nullptr, 0.f, h.xl, b, eta, 0, srange,
                                                   c0.seed, g->step, lo, B, nov, s->g_ent, s->g_rel, 0, s->cfg.flags & AMDKGE_TILED_POS_ATOMIC, h.acc3,
                                                   h.acc3 + 1, nullptr, nullptr, s->twork, s->st);
            if (rc != AMDKGE_OK) { (void)hipFree(s->twork); s->twork = nullptr; s->twork_bytes = 0; return rc; }
        } else {
            KGE_RC(amdkge_train_fwdbwd(&m, &loss, s->tab[0], s->tab[1], h.xl, b, eta, 0, srange, c0.seed, g->step, lo, B, nov, s->g_ent, s->g_rel, h.acc3,
                                       nullptr, nullptr, s->st));
        }
    }
    // ---- 4. gradients of the fetched copies go home; the relation gradient is summed over the replicas ----
    KGE_RC(exchange(g, [&](int d) { return (void*)(g->rep[d]->g_ent + g->sh[(size_t)d].n_local * (int64_t)g->rep[d]->Ks); },
                    [&](int d) { return (void*)g->sh[(size_t)d].back; }, cap * (int64_t)g->rep[0]->Ks, 4, kNcclFloat));
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        Shard& h = g->sh[(size_t)d];
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        KGE_RC(amdkge_scatter_add_rows(s->g_ent, s->Ks, h.recv, Wcap, h.back, s->st));
        KGE_HIP(hipMemsetAsync(s->g_ent + h.n_local * (int64_t)s->Ks, 0, (size_t)Wcap * s->Ks * sizeof(float), s->st), "hipMemsetAsync(scratch gradients)");
    }
    KGE_RC(group_sum(g, [](amdkge_session* s) { return s->g_rel; }, c0.model.n_rels * (int64_t)g->rep[0]->Ks));
    // ---- 5. every replica sweeps its rows and the replicated relation table ----
    double data = 0.0, reg = 0.0;
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        Shard& h = g->sh[(size_t)d];
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        amdkge_opt opt = s->cfg.opt;
        opt.iteration = g->iteration + 1;
        if (h.n_local > 0) KGE_RC(amdkge_opt_step(&opt, s->tab[0], s->g_ent, s->tab[2], s->tab[3], h.n_local * (int64_t)s->Ks, h.acc3 + 1, s->st));
        amdkge_opt orel = opt;
        orel.reg_lambda = s->cfg.rel_reg_lambda;
        if (opt.rel_reg_p > 0) orel.reg_p = opt.rel_reg_p;
        orel.reg2_p = opt.rel_reg2_p; orel.reg2_lambda = opt.rel_reg2_lambda;
        KGE_RC(amdkge_opt_step(&orel, s->tab[1], s->g_rel, s->tab[4], s->tab[5], c0.model.n_rels * (int64_t)s->Ks, h.acc3 + 2, s->st));
    }
    for (int d = 0; d < W; ++d) {
        amdkge_session* s = g->rep[d];
        double hst[3] = {0.0, 0.0, 0.0};
        KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
        KGE_HIP(hipMemcpyAsync(hst, g->sh[(size_t)d].acc3, sizeof(hst), hipMemcpyDeviceToHost, s->st), "hipMemcpyAsync(D2H)");
        KGE_HIP(hipStreamSynchronize(s->st), "hipStreamSynchronize");
        data += hst[0];
        reg += hst[1] + (d == 0 ? hst[2] : 0.0);   // entity shards: summed; the relation table's term is the same on every replica
    }
    g->step += 1;
    g->iteration += 1;
    if (loss_out) *loss_out = data + reg;
    return AMDKGE_OK;
}

// =====================================================================================================================================
// Evaluation through a group.  What it replaces is the reference's loop over entity partitions in evaluate()
// (ScoringBasedEmbeddingModel.py:1431-1452: every partition's rows are the candidates of one get_ranks call whose filter ids are
// restricted to the partition, AbstractScoringLayer.py:280-288; the per-partition counts are summed, +1 once, :1459-1463,1684) --
// with the partitions living on different GPUs:
//   row-sharded group: every replica counts ALL queries of a chunk against ITS rows.  The s / o rows a chunk's queries need are
//     gathered at their owners (amdkge_gather_rows, zero rows elsewhere), summed bit-wise over the replicas (ncclAllReduce of the
//     int32 patterns -- one non-zero contributor per row, so the sum IS the row, -0.0 included; same-device replicas: a kernel) into
//     the scratch rows behind every shard, and the chunk's triples are re-indexed into "shard + scratch" -- the index space the
//     train step uses, so the rank kernels run unchanged with candidates [0, n_local).  Counts and filter subtractions of the
//     replicas are summed the same way (ncclAllReduce / kernel) on every replica; replica 0 applies the tie strategy and the +1.
//   replicated group: every replica holds the whole model: the queries are split over the replicas (one host thread per device).
// Ranks are those of amdkge_session_rank on the whole table, bit for bit (integer counts of the same per-pair comparisons).
namespace {

struct ISumArgs { const int32_t* src[16]; int32_t* dst[16]; int n; int64_t len; };
__global__ void replica_isum_kernel(ISumArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.len; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t t = a.src[0][i];
        for (int q = 1; q < a.n; ++q) t += a.src[q][i];
        for (int q = 0; q < a.n; ++q) a.dst[q][i] = t;
    }
}

// dst(d)[i] = sum over the replicas of src(q)[i] (int32), on every replica, stream-ordered on each replica's stream
template <class FS, class FD>
int group_isum(amdkge_session_group* g, FS src, FD dst, int64_t len) {
    const int n = (int)g->rep.size();
    if (len == 0) return AMDKGE_OK;
    if (n == 1 && g->comm.empty()) {
        if ((const void*)src(0) != (const void*)dst(0)) {
            KGE_HIP(hipSetDevice(g->rep[0]->cfg.device), "hipSetDevice");
            KGE_HIP(hipMemcpyAsync(dst(0), src(0), (size_t)len * 4, hipMemcpyDeviceToDevice, g->rep[0]->st), "hipMemcpyAsync(D2D)");
        }
        return AMDKGE_OK;
    }
    if (g->same_device) {
        amdkge_session* s0 = g->rep[0];
        KGE_HIP(hipSetDevice(s0->cfg.device), "hipSetDevice");
        for (int d = 1; d < n; ++d) {
            KGE_HIP(hipEventRecord(g->ev[d], g->rep[d]->st), "hipEventRecord");
            KGE_HIP(hipStreamWaitEvent(s0->st, g->ev[d], 0), "hipStreamWaitEvent");
        }
        ISumArgs a{};
        a.n = n; a.len = len;
        for (int d = 0; d < n; ++d) { a.src[d] = src(d); a.dst[d] = dst(d); }
        const unsigned grid = (unsigned)((len + 255) / 256 < 4096 ? (len + 255) / 256 : 4096);
        hipLaunchKernelGGL(replica_isum_kernel, dim3(grid), dim3(256), 0, s0->st, a);
        KGE_RC(check_launch("replica_isum"));
        KGE_HIP(hipEventRecord(g->ev[0], s0->st), "hipEventRecord");
        for (int d = 1; d < n; ++d) KGE_HIP(hipStreamWaitEvent(g->rep[d]->st, g->ev[0], 0), "hipStreamWaitEvent");
        return AMDKGE_OK;
    }
    if (int rc = g->rccl.GroupStart()) return rccl_error(g->rccl, rc, "ncclGroupStart");
    for (int d = 0; d < n; ++d)
        if (int rc = g->rccl.AllReduce(src(d), dst(d), (size_t)len, kNcclInt32, kNcclSum, g->comm[(size_t)d], g->rep[d]->st)) {
            (void)g->rccl.GroupEnd();
            return rccl_error(g->rccl, rc, "ncclAllReduce(int32)");
        }
    if (int rc = g->rccl.GroupEnd()) return rccl_error(g->rccl, rc, "ncclGroupEnd");
    return AMDKGE_OK;
}

// fn(d) for every replica d: ONE HOST THREAD PER REPLICA when the replicas sit on distinct devices (or AMDKGE_GROUP_FORCE_THREADS asks
// for it on one device: what a one-GPU box can test) -- the calls fn makes (staging copies, launches, the 8-byte read-back of the
// distance models' early-exit probe, stream synchronisation) then overlap across the devices instead of queueing behind one host
// thread; sequential otherwise.  fn touches only replica d's session, stream and staging.  The first failing replica's code and
// message are reported (error messages are thread-local).
template <class F>
int for_each_replica(amdkge_session_group* g, F&& fn) {
    const int W = (int)g->rep.size();
    if (W == 1 || (g->same_device && !(g->flags & AMDKGE_GROUP_FORCE_THREADS))) {
        for (int d = 0; d < W; ++d) KGE_RC(fn(d));
        return AMDKGE_OK;
    }
    std::vector<int> rcs((size_t)W, AMDKGE_OK);
    std::vector<std::string> msgs((size_t)W);
    std::vector<std::thread> th;
    th.reserve((size_t)W);
    for (int d = 0; d < W; ++d)
        th.emplace_back([&, d]() {
            try {
                rcs[(size_t)d] = fn(d);
            } catch (const std::bad_alloc&) {
                rcs[(size_t)d] = set_error(AMDKGE_ENOMEM, "session_group: out of host memory in a replica's thread");
            } catch (...) {
                rcs[(size_t)d] = set_error(AMDKGE_EINVAL, "session_group: unexpected C++ exception in a replica's thread");
            }
            if (rcs[(size_t)d] != AMDKGE_OK) msgs[(size_t)d] = amdkge_last_error();   // (the message is thread-local)
        });
    for (std::thread& t : th) t.join();
    for (int d = 0; d < W; ++d)
        if (rcs[(size_t)d] != AMDKGE_OK) return set_error(rcs[(size_t)d], msgs[(size_t)d].c_str());
    return AMDKGE_OK;
}

struct DevBuf {   // a per-call device allocation, freed on every path out
    void* p = nullptr; int dev = 0;
    ~DevBuf() { if (p) { (void)hipSetDevice(dev); (void)hipFree(p); } }
};

int rows_rank(amdkge_session_group* g, const int32_t* triples, int64_t n, const int64_t* fs_off, const int32_t* fs_ids, const int64_t* fo_off,
              const int32_t* fo_ids, const int32_t* ent_subset, int64_t n_subset, int32_t corrupt_side, int32_t strategy, int32_t* ranks_out) {
    const int W = (int)g->rep.size();
    const amdkge_session_config& c0 = g->rep[0]->cfg;
    for (int64_t i = 0; i < n; ++i)
        if (triples[3 * i] < 0 || triples[3 * i] >= g->N || triples[3 * i + 2] < 0 || triples[3 * i + 2] >= g->N || triples[3 * i + 1] < 0 ||
            triples[3 * i + 1] >= c0.model.n_rels)
            return set_error(AMDKGE_EINVAL, "session_group_rank: a triple has an entity / relation id outside the tables");
    KGE_RC(amdkge_session_check_filter(fs_off, fs_ids, n, g->N, "session_group_rank"));
    KGE_RC(amdkge_session_check_filter(fo_off, fo_ids, n, g->N, "session_group_rank"));
    const int64_t S = (int64_t)W * g->cap;            // scratch rows behind every shard
    const int64_t chunk = S >= 2 ? S / 2 : 1;          // queries per pass: at most 2 distinct entity rows each
    if (S < 2) return set_error(AMDKGE_EUNSUPPORTED, "session_group_rank: the group was created with fewer than two scratch rows per replica (max_batch too small)");
    const int Ks = g->rep[0]->Ks;
    // entities_subset: per replica the owned candidates in the caller's order (duplicates counted as given) and the local
    // id -> position table of the filter pass (last wins, ScoringBasedEmbeddingModel.py:1639-1643; -1 elsewhere, scratch rows included)
    std::vector<DevBuf> sel((size_t)W);
    std::vector<int64_t> nsel((size_t)W, 0);
    if (n_subset > 0) {
        std::vector<std::vector<int32_t>> lst((size_t)W), pos((size_t)W);
        for (int d = 0; d < W; ++d) pos[(size_t)d].assign((size_t)(g->sh[(size_t)d].n_local + S), -1);
        for (int64_t i = 0; i < n_subset; ++i) {
            const int64_t id = ent_subset[i];
            if (id < 0 || id >= g->N) return set_error(AMDKGE_EINVAL, "session_group_rank: subset id outside the entity table");
            const size_t d = (size_t)(id / g->rows_per);
            pos[d][(size_t)(id - g->sh[d].lo)] = (int32_t)lst[d].size();
            lst[d].push_back((int32_t)(id - g->sh[d].lo));
        }
        for (int d = 0; d < W; ++d) {
            amdkge_session* s = g->rep[d];
            nsel[(size_t)d] = (int64_t)lst[(size_t)d].size();
            KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
            sel[(size_t)d].dev = s->cfg.device;
            const size_t nl = lst[(size_t)d].size(), np = pos[(size_t)d].size();
            KGE_HIP(hipMalloc(&sel[(size_t)d].p, (nl + np) * 4 + 16), "hipMalloc(subset lists)");
            if (nl) KGE_HIP(hipMemcpyAsync(sel[(size_t)d].p, lst[(size_t)d].data(), nl * 4, hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
            KGE_HIP(hipMemcpyAsync((int32_t*)sel[(size_t)d].p + nl, pos[(size_t)d].data(), np * 4, hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
            KGE_HIP(hipStreamSynchronize(s->st), "hipStreamSynchronize");   // the host lists leave scope
        }
    }
    const bool two_cols = corrupt_side == AMDKGE_CORRUPT_S_O;
    const int ncols = (corrupt_side == AMDKGE_CORRUPT_S || corrupt_side == AMDKGE_CORRUPT_O) ? 1 : 2;
    // Host staging: everything a replica's copies read is PER REPLICA (xl, idxl) or read-only during the parallel phases (U, the CSR
    // slices), and lives until the chunk's closing synchronisation -- no synchronisation inside a phase (round 5 shared one `idx` and
    // synchronised every replica's stream before moving to the next: on distinct devices the shards' passes ran one after the other).
    // Per chunk the replicas meet exactly at the two kinds of exchange: the bit-wise sum of the gathered query rows and, per side, the
    // int32 sum of the counts (group_isum: one grouped ncclAllReduce, or the local kernel); everything else of a replica -- staging,
    // gather, prep, probe read-back, count and filter passes -- runs in its own host thread (for_each_replica).
    std::vector<int32_t> U, hr;
    std::vector<std::vector<int32_t>> xl((size_t)W), idxl((size_t)W);
    std::vector<int64_t> offs, offo;
    for (int64_t q0 = 0; q0 < n; q0 += chunk) {
        const int64_t nq = (n - q0) < chunk ? (n - q0) : chunk;
        const int32_t* tq = triples + 3 * q0;
        // ---- the distinct entity rows of the chunk, their slots behind the shards ----
        stage_distinct_rows(tq, nq, U);
        const int64_t nu = (int64_t)U.size();
        std::vector<const int32_t*> d_tri((size_t)W, nullptr);
        KGE_RC(for_each_replica(g, [&](int d) -> int {
            amdkge_session* s = g->rep[d];
            Shard& h = g->sh[(size_t)d];
            KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
            std::vector<int32_t>& x = xl[(size_t)d];
            std::vector<int32_t>& idx = idxl[(size_t)d];
            stage_replica(tq, nq, U, h.lo, h.n_local, x, idx);   // (kge_group_staging.h: CPU-tested)
            void* dt = nullptr;
            KGE_RC(amdkge_session_scratch(s, 0, nq * 12, &dt));
            KGE_HIP(hipMemcpyAsync(dt, x.data(), (size_t)nq * 12, hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
            d_tri[(size_t)d] = (const int32_t*)dt;
            KGE_HIP(hipMemcpyAsync(h.recv, idx.data(), (size_t)nu * 4, hipMemcpyHostToDevice, s->st), "hipMemcpyAsync(H2D)");
            KGE_RC(amdkge_gather_rows(s->tab[AMDKGE_TABLE_ENT], Ks, h.recv, nu, h.rows_out, s->st));
            return AMDKGE_OK;
        }));
        KGE_RC(group_isum(g, [&](int d) { return (const int32_t*)g->sh[(size_t)d].rows_out; },
                          [&](int d) { return (int32_t*)(g->rep[d]->tab[AMDKGE_TABLE_ENT] + g->sh[(size_t)d].n_local * (int64_t)Ks); }, nu * (int64_t)Ks));
        // ---- every wanted side: per-shard counts, summed over the replicas, composed on replica 0 ----
        void* d_ranks = nullptr;
        KGE_HIP(hipSetDevice(g->rep[0]->cfg.device), "hipSetDevice");
        KGE_RC(amdkge_session_scratch(g->rep[0], 3, nq * 2 * (int64_t)sizeof(int32_t), &d_ranks));
        int col = 0;
        for (int side = AMDKGE_SIDE_S; side <= AMDKGE_SIDE_O; ++side) {
            const bool want = (side == AMDKGE_SIDE_S) ? (corrupt_side != AMDKGE_CORRUPT_O) : (corrupt_side != AMDKGE_CORRUPT_S);
            if (!want) continue;
            const int64_t* off = (side == AMDKGE_SIDE_S) ? fs_off : fo_off;
            const int32_t* ids = (side == AMDKGE_SIDE_S) ? fs_ids : fo_ids;
            std::vector<int64_t>& lo = (side == AMDKGE_SIDE_S) ? offs : offo;
            if (off) stage_csr_slice(off, q0, nq, lo);   // the chunk's slice of the CSR, zero-based
            std::vector<int32_t*> c3((size_t)W, nullptr);
            KGE_RC(for_each_replica(g, [&](int d) -> int {
                amdkge_session* s = g->rep[d];
                Shard& h = g->sh[(size_t)d];
                KGE_HIP(hipSetDevice(s->cfg.device), "hipSetDevice");
                amdkge_model m = s->cfg.model;
                m.n_ents = h.n_local + S;
                const int64_t ncand = n_subset > 0 ? nsel[(size_t)d] : h.n_local;
                if (ncand == 0) {   // (a shard without a candidate of the subset)
                    void* z = nullptr;
                    KGE_RC(amdkge_session_scratch(s, 2, nq * 3 * (int64_t)sizeof(int32_t), &z));
                    KGE_HIP(hipMemsetAsync(z, 0, (size_t)nq * 3 * sizeof(int32_t), s->st), "hipMemsetAsync");
                    c3[(size_t)d] = (int32_t*)z;
                    return AMDKGE_OK;
                }
                const int32_t* e_ids = n_subset > 0 ? (const int32_t*)sel[(size_t)d].p : nullptr;
                const int32_t* e_pos = n_subset > 0 ? (const int32_t*)sel[(size_t)d].p + nsel[(size_t)d] : nullptr;
                KGE_RC(amdkge_session_count_side(s, &m, d_tri[(size_t)d], nq, side, off ? lo.data() : nullptr, off ? ids + off[q0] : nullptr, h.lo, h.n_local,
                                                 e_ids, e_pos, 0, ncand, &c3[(size_t)d]));
                return AMDKGE_OK;
            }));
            KGE_HIP(hipSetDevice(g->rep[0]->cfg.device), "hipSetDevice");
            KGE_RC(group_isum(g, [&](int d) { return (const int32_t*)c3[(size_t)d]; }, [&](int d) { return c3[(size_t)d]; }, nq * 3));
            KGE_HIP(hipSetDevice(g->rep[0]->cfg.device), "hipSetDevice");
            KGE_RC(amdkge_rank_compose(c3[0], off ? c3[0] + 2 * nq : nullptr, nq, strategy, (int32_t*)d_ranks + (two_cols ? col : col * nq), two_cols ? 2 : 1, g->rep[0]->st));
            ++col;
        }
        // ---- the chunk's ranks to the host; every replica idle before its scratch (and the host staging above) is reused ----
        hr.resize((size_t)(nq * ncols));
        KGE_HIP(hipSetDevice(g->rep[0]->cfg.device), "hipSetDevice");
        KGE_HIP(hipMemcpyAsync(hr.data(), d_ranks, (size_t)nq * ncols * sizeof(int32_t), hipMemcpyDeviceToHost, g->rep[0]->st), "hipMemcpyAsync(D2H)");
        for (int d = 0; d < W; ++d) {
            KGE_HIP(hipSetDevice(g->rep[d]->cfg.device), "hipSetDevice");
            KGE_HIP(hipStreamSynchronize(g->rep[d]->st), "hipStreamSynchronize");
        }
        if (corrupt_side == AMDKGE_CORRUPT_S_PLUS_O) for (int64_t i = 0; i < nq; ++i) ranks_out[q0 + i] = hr[(size_t)i] + hr[(size_t)(nq + i)] - 1;   // (:1459-1463,1684)
        else memcpy(ranks_out + q0 * ncols, hr.data(), (size_t)nq * ncols * sizeof(int32_t));
    }
    return AMDKGE_OK;
}

}  // namespace

extern "C" int amdkge_session_group_rank(amdkge_session_group* g, const int32_t* triples, int64_t n, const int64_t* fs_off, const int32_t* fs_ids,
                                         const int64_t* fo_off, const int32_t* fo_ids, const int32_t* ent_subset, int64_t n_subset,
                                         int32_t corrupt_side, int32_t strategy, int32_t* ranks_out) try {
    if (!g || n < 0 || corrupt_side < AMDKGE_CORRUPT_S || corrupt_side > AMDKGE_CORRUPT_S_PLUS_O)
        return set_error(AMDKGE_EINVAL, "session_group_rank: bad arguments (corrupt_side must be AMDKGE_CORRUPT_*)");
    if (strategy < 0 || strategy > 2) return set_error(AMDKGE_EINVAL, "session_group_rank: unknown ranking strategy");
    if (n == 0) return AMDKGE_OK;
    if (!triples || !ranks_out) return set_error(AMDKGE_EINVAL, "session_group_rank: NULL buffer");
    if (n_subset < 0 || (n_subset > 0 && !ent_subset)) return set_error(AMDKGE_EINVAL, "session_group_rank: bad entities subset");
    if (g->cols) return set_error(AMDKGE_EUNSUPPORTED, "session_group_rank: a column-sharded group holds no whole rows -- gather them (amdkge_session_group_get_rows) into one amdkge_session to evaluate");
    if (g->rows) return rows_rank(g, triples, n, fs_off, fs_ids, fo_off, fo_ids, ent_subset, n_subset, corrupt_side, strategy, ranks_out);
    // replicated tables: replica d ranks the queries [n d / W, n (d + 1) / W) -- its slice of the offsets indexes the whole id arrays
    const int W = (int)g->rep.size();
    const int ncols = corrupt_side == AMDKGE_CORRUPT_S_O ? 2 : 1;
    auto share = [&](int d) {
        const int64_t lo = n * d / W, hi = n * (d + 1) / W;
        if (hi == lo) return (int)AMDKGE_OK;
        return amdkge_session_rank(g->rep[(size_t)d], triples + 3 * lo, hi - lo, fs_off ? fs_off + lo : nullptr, fs_ids, fo_off ? fo_off + lo : nullptr, fo_ids,
                                   ent_subset, n_subset, corrupt_side, strategy, ranks_out + lo * ncols);
    };
    return for_each_replica(g, share);   // (one host thread per device)
} KGE_CATCH("session_group_rank")


// =====================================================================================================================================
// AMDKGE_GROUP_COLS -- every table COLUMN-sharded over the replicas of a group (kge_train_cols.h).
// Replica d holds units [d k / W, (d + 1) k / W) of EVERY entity and relation row (the re and im slices of the same units for the
// complex models) with the optimizer state of those columns, and processes the WHOLE batch on its slice: the scores are sums over
// units, so the one exchange of a step is the all-reduce of B (1 + eta) partial sums -- 6.7 MB at B = 80 000, eta = 20, whatever
// the table size, against the two table-sized exchanges of the replicated and the row-sharded modes at BASELINE configs[1].  Loss,
// backward, gradient merge, regulariser and optimizer are element-wise in the columns: local.  What a step replaces is
// ScoringBasedEmbeddingModel.train_step (ScoringBasedEmbeddingModel.py:370-429) on one global batch; the reference has no
// multi-device path.  W replicas compute one GPU's step up to fp32 summation order (the same Philox corruptions: every replica
// draws them for the whole batch).  Replicas on ONE device sum the partial scores with the library's kernel (tests on a one-GPU box).
namespace {

// dense host rows [nrows, internal_k(k_full)] <-> replica d's slice [nrows, internal_k(k_full / W)]
void col_slice(const float* full, int64_t nrows, int model, int k_full, int W, int d, float* out) {
    const int kp = k_full / W, halves = internal_k_of(model, 1);   // 1 or 2 halves per row
    for (int64_t r = 0; r < nrows; ++r)
        for (int h = 0; h < halves; ++h)
            memcpy(out + (r * halves + h) * (int64_t)kp, full + (r * halves + h) * (int64_t)k_full + (int64_t)d * kp, (size_t)kp * sizeof(float));
}
void col_merge(const float* slice, int64_t nrows, int model, int k_full, int W, int d, float* full) {
    const int kp = k_full / W, halves = internal_k_of(model, 1);
    for (int64_t r = 0; r < nrows; ++r)
        for (int h = 0; h < halves; ++h)
            memcpy(full + (r * halves + h) * (int64_t)k_full + (int64_t)d * kp, slice + (r * halves + h) * (int64_t)kp, (size_t)kp * sizeof(float));
}

}  // namespace

extern "C" int amdkge_session_group_create_cols(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                                amdkge_session_group** out) try {
    if (!cfg || !out || n_gpus < 1 || n_gpus > 16) return set_error(AMDKGE_EINVAL, "session_group_create_cols: bad arguments (1 <= n_gpus <= 16)");
    if (flags & ~(AMDKGE_GROUP_FORCE_RCCL | AMDKGE_GROUP_COLS)) return set_error(AMDKGE_EINVAL, "session_group_create_cols: unknown flag");
    *out = nullptr;
    if (cfg->model.k % n_gpus != 0) return set_error(AMDKGE_EINVAL, "session_group_create_cols: k must be a multiple of the number of replicas");
    if (cfg->model.k_full != 0) return set_error(AMDKGE_EINVAL, "session_group_create_cols: cfg->model describes the WHOLE model (k_full = 0)");
    if (cfg->flags & (AMDKGE_TILED_DETERMINISTIC | AMDKGE_TILED_POS_ATOMIC)) return set_error(AMDKGE_EUNSUPPORTED, "session_group_create_cols: DETERMINISTIC / POS_ATOMIC are not offered for column-sharded groups");
    if (cfg->loss.focus_nonlinearity) return set_error(AMDKGE_EUNSUPPORTED, "session_group_create_cols: FocusE is not offered for column-sharded groups");
    if (amdkge_padded_k(cfg->model.k / n_gpus) > 256) return set_error(AMDKGE_EUNSUPPORTED, "session_group_create_cols: a replica's slice may hold up to 256 units per half (use more replicas)");
    amdkge_session_config c = *cfg;
    c.model.k = cfg->model.k / n_gpus;
    c.model.k_full = cfg->model.k;
    amdkge_session_group* g = nullptr;
    if (int rc = group_create(&c, devices, n_gpus, flags & AMDKGE_GROUP_FORCE_RCCL, cfg->model.n_ents, &g)) return rc;
    g->flags = flags | AMDKGE_GROUP_COLS;
    g->cols = true;
    g->k_full = cfg->model.k;
    g->N = cfg->model.n_ents;
    *out = g;
    return AMDKGE_OK;
} KGE_CATCH("session_group_create_cols")

static int cols_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host) {
    if (table < 0 || table > 5) return set_error(AMDKGE_EINVAL, "session_group_set_rows: no such table");
    if (nrows == 0) return AMDKGE_OK;
    if (!host || nrows < 0) return set_error(AMDKGE_EINVAL, "session_group_set_rows: bad host buffer / row count");
    const int W = (int)g->rep.size(), model = g->rep[0]->cfg.model.scoring_type;
    std::vector<float> tmp((size_t)nrows * g->rep[0]->K);
    for (int d = 0; d < W; ++d) {
        col_slice(host, nrows, model, g->k_full, W, d, tmp.data());
        KGE_RC(amdkge_session_set_rows(g->rep[d], table, row0, nrows, tmp.data()));   // (validates table / rows; synchronises)
    }
    // a poisoned group (cols_train_step) is usable again once the whole entity table has been written: the caller reloading a
    // checkpoint writes every table, and this is the one it cannot leave out
    if (table == 0 && row0 == 0 && nrows == g->N) g->poisoned = false;
    return AMDKGE_OK;
}

static int cols_get_rows(amdkge_session_group* g, int32_t table, const int32_t* ids, int64_t row0, int64_t nrows, float* host) {
    const int W = (int)g->rep.size(), model = g->rep[0]->cfg.model.scoring_type;
    std::vector<float> tmp((size_t)nrows * g->rep[0]->K);
    for (int d = 0; d < W; ++d) {
        KGE_RC(amdkge_session_get_rows(g->rep[d], table, ids, row0, nrows, tmp.data()));
        col_merge(tmp.data(), nrows, model, g->k_full, W, d, host);
    }
    return AMDKGE_OK;
}

static int cols_train_step_body(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out, int* applied);

//  A column-sharded step that fails part-way.  Before any slice has applied its update (phase A, the exchange) nothing of the
// model has moved: gradients are cleared, workspaces dropped (their bookkeeping may be dirty) and the step can be retried.  Once SOME
// slices have run their optimizer (phase B + C is one fused launch sequence per slice) the columns of one model are at different steps,
// and the step / iteration counters -- bumped by finish_step only -- would let a retried step update the applied slices twice with the same
// Philox draws: there is nothing to roll back to, so the group is POISONED: train_step refuses until the tables have been written
// again through amdkge_session_group_set_rows (a checkpoint: the whole entity table clears the state), or the group is rebuilt.  The failed call's message is kept.
static int cols_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out) {
    if (g->poisoned)
        return set_error(AMDKGE_EINVAL, "session_group_train_step: an earlier column-sharded step failed after some slices had been updated; "
                                        "reload the tables (amdkge_session_group_set_rows) or rebuild the group");
    int applied = 0;
    const int rc = cols_train_step_body(g, triples, B, focus_w, loss_out, &applied);
    if (rc == AMDKGE_OK) return rc;
    for (amdkge_session* s : g->rep) {
        (void)hipSetDevice(s->cfg.device);
        (void)hipStreamSynchronize(s->st);
        (void)hipMemsetAsync(s->g_ent, 0, (size_t)s->cfg.model.n_ents * s->Ks * sizeof(float), s->st);
        (void)hipMemsetAsync(s->g_rel, 0, (size_t)s->cfg.model.n_rels * s->Ks * sizeof(float), s->st);
        if (s->twork) { (void)hipFree(s->twork); s->twork = nullptr; s->twork_bytes = 0; }
        (void)hipStreamSynchronize(s->st);
    }
    if (applied > 0) g->poisoned = true;
    return rc;
}

static int cols_train_step_body(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out, int* applied) {
    const int W = (int)g->rep.size();
    if (loss_out) *loss_out = 0.0;
    if (B == 0) return AMDKGE_OK;
    if (!triples) return set_error(AMDKGE_EINVAL, "session_group_train_step: NULL triples");
    if (focus_w) return set_error(AMDKGE_EUNSUPPORTED, "session_group_train_step: FocusE is not offered for column-sharded groups");
    // ---- A: every replica scores the WHOLE batch on its columns ----
    std::vector<float*> sc((size_t)W, nullptr);
    for (int d = 0; d < W; ++d) KGE_RC(amdkge_session_cols_scores(g->rep[d], triples, B, &sc[(size_t)d]));
    // ---- the one exchange of the step: the partial sums meet (ncclAllReduce over xGMI / the local kernel) ----
    KGE_RC(group_sum(g, [](amdkge_session* s) { return (float*)s->buf[2]; }, B * (int64_t)(1 + g->rep[0]->cfg.eta)));
    // ---- B + C: loss on the complete scores (the same values on every replica), backward / merge / optimizer on the slice ----
    for (int d = 0; d < W; ++d) {
        *applied = d + 1;   // (a slice whose apply fails may have launched part of its sequence: it counts)
        KGE_RC(amdkge_session_cols_apply(g->rep[d], B));
    }
    double data = 0.0, reg = 0.0;
    for (int d = 0; d < W; ++d) {
        double h[2];
        KGE_RC(amdkge_session_finish_step(g->rep[d], h));
        if (d == 0) data = h[0];   // (every replica evaluates the same loss on the same complete scores)
        reg += h[1];               // (the regulariser is a sum over elements: the slices' terms add up)
    }
    if (loss_out) *loss_out = data + reg;
    return AMDKGE_OK;
}
