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

#include <vector>

#include "kge_session_impl.h"

using namespace kge;

namespace {

#define KGE_HIP(call, what) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return set_error_hip(e_, what); } while (0)
#define KGE_RC(call) do { const int rc_ = (call); if (rc_ != AMDKGE_OK) return rc_; } while (0)

// the slice of rccl.h this file needs (ABI of RCCL 2.x: ncclResult_t / ncclDataType_t / ncclRedOp_t are plain enums)
typedef void* ncclComm_t;
enum { kNcclSuccess = 0, kNcclFloat = 7, kNcclSum = 0 };
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
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
    KGE_SYM(GetVersion, "ncclGetVersion");
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
};

extern "C" void amdkge_session_group_destroy(amdkge_session_group* g) {
    if (!g) return;
    for (size_t d = 0; d < g->comm.size(); ++d)
        if (g->comm[d] && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(g->comm[d]);
    for (size_t d = 0; d < g->ev.size(); ++d)
        if (g->ev[d]) { (void)hipSetDevice(g->rep[d]->cfg.device); (void)hipEventDestroy(g->ev[d]); }
    for (amdkge_session* s : g->rep) amdkge_session_destroy(s);
    if (g->rccl.handle) (void)dlclose(g->rccl.handle);
    delete g;
}

extern "C" int amdkge_session_group_create(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, amdkge_session_group** out) {
    return amdkge_session_group_create_ex(cfg, devices, n_gpus, 0, out);
}

extern "C" int amdkge_session_group_info(const amdkge_session_group* g, int32_t* uses_rccl, int32_t* rccl_version) {
    if (!g) return set_error(AMDKGE_EINVAL, "session_group_info: NULL group");
    if (uses_rccl) *uses_rccl = g->comm.empty() ? 0 : 1;
    if (rccl_version) {
        int v = 0;
        if (g->rccl.GetVersion) (void)g->rccl.GetVersion(&v);
        *rccl_version = v;
    }
    return AMDKGE_OK;
}

extern "C" int amdkge_session_group_create_ex(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                              amdkge_session_group** out) {
    if (!cfg || !out || n_gpus < 1 || n_gpus > 16) return set_error(AMDKGE_EINVAL, "session_group_create: bad arguments (1 <= n_gpus <= 16)");
    if (flags & ~AMDKGE_GROUP_FORCE_RCCL) return set_error(AMDKGE_EINVAL, "session_group_create_ex: unknown flag");
    *out = nullptr;
    amdkge_session_group* g = new amdkge_session_group();
    g->flags = flags;
    auto fail = [&](int rc) { amdkge_session_group_destroy(g); return rc; };
    bool same = true, distinct = true;
    for (int d = 0; d < n_gpus; ++d) {
        amdkge_session_config c = *cfg;
        c.device = devices ? devices[d] : d;
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

extern "C" int32_t amdkge_session_group_size(const amdkge_session_group* g) { return g ? (int32_t)g->rep.size() : 0; }

extern "C" int amdkge_session_group_replica(amdkge_session_group* g, int32_t i, amdkge_session** out) {
    if (!g || !out || i < 0 || i >= (int32_t)g->rep.size()) return set_error(AMDKGE_EINVAL, "session_group_replica: no such replica");
    *out = g->rep[(size_t)i];
    return AMDKGE_OK;
}

extern "C" int amdkge_session_group_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host) {
    if (!g) return set_error(AMDKGE_EINVAL, "session_group_set_rows: NULL group");
    for (amdkge_session* s : g->rep) KGE_RC(amdkge_session_set_rows(s, table, row0, nrows, host));
    return AMDKGE_OK;
}

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

extern "C" int amdkge_session_group_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out) {
    if (!g || B < 0) return set_error(AMDKGE_EINVAL, "session_group_train_step: bad arguments");
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
}
