// The fused forward(+backward) training kernel template, shared by the atomic-scatter path
// (kge_train.hip) and the owner-computes path (kge_train_tiled.hip).  See kge_train.hip for the
// description of the slot mapping.
#pragma once
#include "kge_device.h"
#include "kge_host.h"

namespace kge {

// one row-gradient contribution, appended by the forward kernel to the bucket of the tile owning row `dest`
struct __attribute__((aligned(16))) StageEntry {
    uint32_t pos;    // positive (index into this launch's batch)
    uint32_t meta;   // role | local row of the tile << 2 [| corruption index << 16: TransE sign codes, see below];
                     // role 0/1 = corruption with object/subject replaced, 2/3 = the positive's own s/o row
    float g;         // dL/dscore * score_sign * score_scale (1 for roles 2, 3)
    uint32_t dest;   // global row id
};

constexpr uint32_t ENTRY_LOCAL_MASK = 0x1FFFu;   // local row (tiles hold at most 4096 rows)
__host__ __device__ __forceinline__ uint32_t entry_local(uint32_t meta) { return (meta >> 2) & ENTRY_LOCAL_MASK; }
// TransE sign codes (one wave per positive).  The gradient of -sum |d| w.r.t. the replaced row is -/+ g sign(d_j): all the
// tile pass needs of corruption j is the SIGN of every unit of d_j = s + p - o, which the forward kernel has in registers when
// it scores the row.  It stores them -- the top byte of each of the lane's four d values packed into one dword, [B][eta][nq]
// dwords -- and the tile pass reads 4 bytes per lane and entry instead of recomputing d_j from three K-float rows (staged
// side copy, relation row, its own row: 2.4 KB per entry at k = 200).  sign(0) = 0 stays exact the slow way: the byte also
// carries the top 7 exponent bits, and an entry with a unit of the model whose |d| is below 2^-125 (zero, or as good as) is
// recomputed by the tile pass in the three-row form.
constexpr int ENTRY_J_SHIFT = 16;                // corruption index (eta <= 65535 when codes are in use)

// Per-block loss partials.  Thousands of blocks adding an fp64 atomic to ONE address serialise at the L2 (measured: 19 of the
// 46 us of a C1 forward kernel, 5 us at C2); spread over LOSS_PARTS cache lines they do not, and a single thread folds the
// partials into the caller's accumulator afterwards (the tile kernel's last workgroup, or loss_fold_kernel on the atomic path).
constexpr int LOSS_PARTS = 64;
constexpr int LOSS_PART_STRIDE = 16;   // doubles

// Hot rows (skewed graphs).  An entity that is the s / o of thousands of positives of one batch would put thousands of entries
// on one row of one tile (one wave adds them one after the other) or, through POS_ATOMIC, thousands of atomic row-adds on the
// same addresses.  Up to HOT_MAX such rows, named by the host, are instead spread over HOT_REPL replica rows: positive i adds
// its s / o gradient row atomically into replica (block index mod HOT_REPL), the owning tile sums the replicas when it flushes
// the row.  Every other row keeps the atomic-free staged path.
// Block-interleaved row ownership of the tile pass: rows are dealt to the tiles in blocks of TILE_RB consecutive rows
// (block b -> tile b % n_tiles).  See tile_backward_kernel.
#ifndef KGE_TILE_RB
#define KGE_TILE_RB 8
#endif
constexpr uint32_t TILE_RB = KGE_TILE_RB;   // (a plan uses fewer when the LDS cannot hold TILE_RB rows: rb = min(TILE_RB, tile_rows))
__host__ __device__ __forceinline__ void tile_of_row(uint32_t row, uint32_t n_tiles, uint32_t rb, uint32_t& tile, uint32_t& local) {
    const uint32_t blk = row / rb;
    tile = blk % n_tiles;
    local = (blk / n_tiles) * rb + row % rb;
}
__host__ __device__ __forceinline__ int64_t row_of_tile(uint32_t tile, uint32_t local, uint32_t n_tiles, uint32_t rb) {
    return ((int64_t)(local / rb) * n_tiles + tile) * rb + local % rb;
}

constexpr int HOT_MAX = 64;
constexpr int HOT_REPL = 16;

struct TrainArgs {
    const float* ent;
    const float* rel;
    const float* rel_cs;     // RotatE, owner-computes path: [R][cos(phase) || sin(phase)] of this step's relation table (rel_phase_kernel);
                             // NULL: the kernel evaluates cos / sin itself (prep_rel)
    const int32_t* triples;
    const int32_t* neg_override;
    float* g_ent;
    float* g_rel;
    double* loss_sum;
    double* loss_parts;      // [LOSS_PARTS] partial sums, 128-byte stride: blocks add here, one thread folds them into loss_sum
    float* pos_scores;
    float* neg_scores;
    int64_t B;
    int eta;
    int k;       // units per half AS STORED (k_pad of the model descriptor, or k for dense rows)
    int K;       // floats per stored row
    int k_live;  // the model's k: units >= k_live of a half are zero padding (only RotatE's gradient needs to know)
    int nq;      // quads per row ( = units / VEC )
    SampleCfg sc;
    ModelConst mc;
    amdkge_loss loss;
    // owner-computes (STAGE) outputs: see kge_train_tiled.hip
    float* stage_rows;       // [B][4][K]: gradient rows of the positive's s and o (unless pos_atomic), then the side rows A, B
    int pos_atomic;          // the positives' own s / o rows go through atomics into g_ent (skewed graphs)
    int sign_off;            // TransE: byte offset of the sign stash in dynamic LDS (see SIGNSTASH in the kernel)
    int ns;                  // staged rows per positive: 4, or 5 in deterministic mode (the relation-row gradient is staged too)
    int det;                 // deterministic mode (AMDKGE_TILED_DETERMINISTIC): no atomics on any gradient
    const uint8_t* hot_map;  // AMDKGE_TILED_HOT_ROWS: byte per entity row, slot + 1 of a hot row, 0 otherwise (NULL: feature off)
    float* hot_buf;          // [HOT_MAX][HOT_REPL][K]: replicas the gradient rows of hot entities are spread over
    uint8_t* touched;        // pos_atomic + lazy optimizer: byte per entity row, set for rows that received an atomic row-add
    uint32_t* sign_codes;    // TransE, one wave per positive: [B][eta][nq] packed sign bytes of d_j (see ENTRY_J_SHIFT); NULL = off
    StageEntry* st_lists;    // [n_tiles][cap] buckets of row-gradient entries, by owning tile
    StageEntry* st_ovf;      // overflow of full buckets
    int* st_counters;        // [(n_tiles + 1) * 32] bucket fill counts (128-byte stride), last = overflow count
    int st_tile_rows, st_n_tiles, st_cap, st_ovf_cap, st_rb;
#ifdef KGE_ABLATE
    int dbg;     // development ablation build only (make EXTRA=-DKGE_ABLATE, env AMDKGE_DEBUG): 1 no neg-row atomics, 2 no s/p/o atomics, 4 no pass 2, 32 no bucket appends, 64 no staged-row stores
#endif
};

// ablation switches exist only in development builds; the release library cannot skip work
#ifdef KGE_ABLATE
#define KGE_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define KGE_DBG(a, bit) false
#endif

// fold the per-block partials into the caller's accumulator and leave them zero (one thread, fixed order)
// (atomic exchanges: the partials may have been written by other CUs of the SAME launch -- the tile kernel's regulariser
// terms -- and must be read from the L2, not from this CU's L1)
// Called by ONE whole wave: lane i takes partial i (LOSS_PARTS == 64), the wave adds them up in a fixed butterfly order.
__device__ __forceinline__ void fold_loss_parts(double* parts, double* loss_sum, int lane, int lane_in_slot = 0) {
    static_assert(LOSS_PARTS == KGE_WAVE, "one partial per lane");
    const unsigned long long old = atomicExch(reinterpret_cast<unsigned long long*>(parts + (size_t)lane * LOSS_PART_STRIDE + lane_in_slot), 0ull);
    double t = __longlong_as_double((long long)old);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0 && loss_sum && t != 0.0) atomicAdd(loss_sum, t);
}

__device__ __forceinline__ float log_sigmoid(float x) {
    // -softplus(-x), stable on both tails
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) {
    return 1.f / (1.f + expf(-x));
}

// clip_before_exp (loss_functions.py:60-66) = tf.clip_by_value = maximum(minimum(x, 75), -75) built from TensorFlow's NaN-PROPAGATING
// minimum / maximum: a NaN score stays NaN in the loss VALUE.  C's fminf / fmaxf return the other operand instead, which made a NaN
// positive cost a finite eta * log(1 + e^75) = 375 eta.  The GRADIENT through the clip is the exact zero of the
// minimum / maximum masks (less_equal / greater_equal are false for NaN) -- what the `in` range tests beside every clip give.
__device__ __forceinline__ float clip_exp(float x) {
    return (x != x) ? x : fminf(fmaxf(x, -75.f), 75.f);
}
// tf.maximum(h, 0) of the two margin losses (:302-308, :458-464): NaN in the value, zero gradient (greater_equal(NaN, 0) is false)
__device__ __forceinline__ float hinge_nan(float h) {
    return (h != h) ? h : fmaxf(h, 0.f);
}
// A coefficient dL/dscore that a clip or hinge MASK sets to zero.  "Zero coefficient -> no bucket entry for the replaced row" (below,
// STAGE) is valid only while the score's Jacobian is finite: TensorFlow multiplies the exact zero by it, so a positive whose own rows
// hold a NaN hands 0 * NaN = NaN to the replacement rows of its corruptions.  The masked coefficient of a NON-FINITE score (x: the score,
// or the hinge argument it enters) is therefore written as -0.0f: the entry test keeps it (entry_wanted), the tile pass adds (-0) * A --
// NaN exactly where the side row is NaN, nothing elsewhere.  No loss produces -0.0f as a live coefficient.
__device__ __forceinline__ float masked_zero(float x) {
    return (fabsf(x) < INFINITY) ? 0.f : -0.f;
}
// The self-adversarial loss has no mask: its coefficient is COMPUTED, and for a score of -inf (a distance model's corruption that keeps a
// row holding an inf) it is an exact zero -- softmax weight 0 times sigma(-inf) = 0 -- which TensorFlow again multiplies by the Jacobian.
__device__ __forceinline__ float computed_zero(float c, float x) {
    return (c == 0.f) ? masked_zero(x) : c;
}
// coeff: dL/dscore as the loss code left it; g = coeff * score_sign * score_scale.  No entry below fp32's smallest NORMAL number
// (see the forward kernel) unless the coefficient is masked_zero's marker; a NaN coefficient is an entry.
__device__ __forceinline__ bool entry_wanted(float coeff, float g) {
    return !(fabsf(g) < 1.17549435e-38f) || __float_as_uint(coeff) == 0x80000000u;
}

// FocusE (ScoringBasedEmbeddingModel.py:396-406,492-513): y = f(x) * wgt, dfac = f'(x) * wgt.  The reference's
// "softplus" is log(1 + 9999 e^x) with the custom gradient 1 - 1/(1 + 9999 e^x) (:499-510).
__device__ __forceinline__ void focus_apply(int nl, float x, float wgt, float& y, float& dfac) {
    float f, fp;
    switch (nl) {
        case AMDKGE_FOCUS_TANH: f = tanhf(x); fp = 1.f - f * f; break;
        case AMDKGE_FOCUS_SIGMOID: f = sigmoidf(x); fp = f * (1.f - f); break;
        case AMDKGE_FOCUS_SOFTPLUS: { const float e = 9999.f * expf(x); f = logf(1.f + e); fp = 1.f - 1.f / (1.f + e); } break;
        default: f = x; fp = 1.f; break;
    }
    y = f * wgt;
    dfac = fp * wgt;
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
                acc += hinge_nan(h);
                cnt += act ? 1.f : 0.f;
                sn[j] = act ? 1.f / red : masked_zero(h);
            }
            per = wave_sum(acc) / red;
            dP = -wave_sum(cnt) / red;
        } break;
        case AMDKGE_LOSS_NLL: {  // :376-382 (clip at :60-66)
            if (L.reduction_mean) red = 2.f * feta;
            const bool inP = (P >= -75.f) && (P <= 75.f);
            const float Pc = clip_exp(P);
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const bool in = (n >= -75.f) && (n <= 75.f);
                const float nc = clip_exp(n);
                acc += logf(1.f + expf(nc));
                sn[j] = in ? sigmoidf(nc) / red : masked_zero(n);
            }
            per = (feta * logf(1.f + expf(-Pc)) + wave_sum(acc)) / red;
            dP = inP ? -feta * sigmoidf(-Pc) / red : 0.f;
        } break;
        case AMDKGE_LOSS_ABSOLUTE_MARGIN: {  // :458-464
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float h = L.margin + sn[j];
                acc += hinge_nan(h);
                sn[j] = (h >= 0.f) ? 1.f / red : masked_zero(h);
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
                sn[j] = computed_zero((w * sigmoidf(n + L.margin) - L.alpha * w * (ell - lbar)) / red, n);
            }
            per = -log_sigmoid(L.margin + P) - lbar / red;
            dP = -sigmoidf(-(L.margin + P));
        } break;
        default: {  // AMDKGE_LOSS_MULTICLASS_NLL :647-654
            const bool inP = (P >= -75.f) && (P <= 75.f);
            const float eP = expf(clip_exp(P));
            float acc = 0.f;
            for (int j = lane; j < eta; j += KGE_WAVE) acc += expf(clip_exp(sn[j]));
            const float Z = wave_sum(acc) / red + eP;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const bool in = (n >= -75.f) && (n <= 75.f);
                sn[j] = in ? expf(clip_exp(n)) / Z / red : masked_zero(n);
            }
            per = -logf(eP / Z);
            dP = inP ? -1.f + eP / Z : 0.f;
        } break;
    }
}

// Single-pass backward for the trilinear models (DistMult / ComplEx / HolE).  There d(score)/d(s,p,o) is LINEAR in
// the replaced row e_j, so  sum_j g_j * d/d(.)(e_j) = d/d(.)(sum_j g_j e_j): the replacement rows need not be read
// a second time if sum_j g_j e_j can be formed while they stream by.  Every loss's dL/dneg_j factors as
//     g_j = kappa1 * c1_j + kappa2 * c2_j
// with c1_j, c2_j known when row j is scored (given the positive's score and running statistics) and
// kappa1, kappa2 known after the last row:
//   pairwise / nll / absolute_margin : c1_j = g_j * red, kappa1 = 1/red                     (loss_functions.py:302-308,376-382,458-464)
//   multiclass_nll                   : c1_j = exp(clip n_j) [in range], kappa1 = 1/(Z red)  (:647-654)
//   self_adversarial (softmax in-graph, :556-574), online softmax with running max m of alpha*n:
//        u_j = exp(alpha n_j - m), c1_j = u_j (sigma(n_j+gamma) - alpha l_j), c2_j = u_j, l_j = log sigma(-n_j-gamma)
//        kappa1 = 1/(S red), kappa2 = alpha (sum u l / S)/(S red), S = sum u   (accumulators rescaled when m grows)
struct OnePassState {
    float m;             // running maximum of alpha * n (wave-uniform)
    float S, Lw, Zs;     // PER-LANE partial sums (lane f accumulates the rows it evaluated):
                         //   self_adversarial: S = sum u, Lw = sum u*l ; nll: Lw = sum softplus(n) ; multiclass: Zs = sum exp(n)
};

// The single-pass path evaluates the loss terms once per GROUP of rows (4 lanes busy), so their instruction count
// matters: hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each) instead of the libm expansions.
__device__ __forceinline__ float fast_exp(float x) {
    // 2^(x*log2e) with the rounding error of the product folded back in: ~2 ulp over the clipped range |x| <= 75
    // (arguments below -110 give 0: e^-110 is under fp32's smallest denormal, and for -inf -- a distance model's score of a row that
    // holds an inf, the online softmax's x - max -- the residual would be inf - inf = NaN.  NaN stays NaN.  As a select on the RESULT:
    // clamping the argument instead cost the ComplEx forward kernel its 168th register, i.e. the third wave per SIMD.)
    const float t = x * 1.4426950216293335f;                                       // fl(log2 e)
    const float r = fmaf(x, 1.4426950216293335f, -t) + x * 1.9259629911266175e-8f;  // exact residual + low part of log2 e
    const float v = __builtin_amdgcn_exp2f(t) * fmaf(r, 0.6931471805599453f, 1.f);
    return (x < -110.f) ? 0.f : v;   // (x = -inf: r is inf - inf)
}
// sigma(y) and log sigma(-y) = -softplus(y) from one exponential
__device__ __forceinline__ void fast_sig_logsig(float y, float& sig, float& logsig_neg) {
    const float t = fast_exp(-fabsf(y));
    const float u = 1.f + t;
    const float r = __builtin_amdgcn_rcpf(u);
    sig = (y >= 0.f) ? r : t * r;
    // log1p(t) = log(u) + (t - (u - 1)) / u: the second term restores what the rounding of 1 + t dropped (for t < 6e-8 it is
    // all of t; TransE scores of -20 and below live there, and their dL/dscore is exactly these tails)
    logsig_neg = fminf(-y, 0.f) - (__builtin_amdgcn_logf(u) * 0.6931471805599453f + (t - (u - 1.f)) * r);
}

// DETERMINISTIC mode (AMDKGE_TILED_DETERMINISTIC): the loss terms from DECLARED transcendentals -- exp and log built from IEEE
// fp32 operations only (multiply, add, divide, round-to-even, exact scaling by a power of two; -ffp-contract=off keeps every
// operation where it is written), so that a CPU restates them bit for bit (oracle/train_ordered.py det_exp / det_log12: the same
// operations in numpy float32).  The hardware v_exp / v_log / v_rcp of the default mode are 1 ulp functions no CPU reproduces.
// About 1 ulp accurate themselves; ~30 more instructions per evaluation, which only this mode pays.
__device__ __forceinline__ float det_exp(float x) {
    // e^x for |x| <= 80 (clamped): n = rint(x log2 e); r = (x - n ln2_hi) - n ln2_lo (Cody-Waite: ln2_hi = 355/512 has 9 bits, so
    // n ln2_hi is exact for |n| < 2^15); e^r by its Taylor polynomial of degree 7 in Horner form; scaled by 2^n exactly
    x = (x != x) ? x : fminf(fmaxf(x, -80.f), 80.f);   // (NaN stays NaN, as np.clip keeps it in oracle/train_ordered.py)
    const float n = rintf(x * 1.4426950216293335f);
    const float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float p = 1.f / 5040.f;
    p = p * r + 1.f / 720.f;
    p = p * r + 1.f / 120.f;
    p = p * r + 1.f / 24.f;
    p = p * r + 1.f / 6.f;
    p = p * r + 0.5f;
    p = p * r + 1.f;
    p = p * r + 1.f;
    return ldexpf(p, (int)n);
}
__device__ __forceinline__ float det_log12(float u) {
    // log(u) for u in [1, 2]: 2 atanh(z), z = (u - 1) / (u + 1) in [0, 1/3]; the odd series up to z^15 (next term < 2e-9 relative)
    const float z = (u - 1.f) / (u + 1.f);
    const float w = z * z;
    float p = 1.f / 15.f;
    p = p * w + 1.f / 13.f;
    p = p * w + 1.f / 11.f;
    p = p * w + 1.f / 9.f;
    p = p * w + 1.f / 7.f;
    p = p * w + 1.f / 5.f;
    p = p * w + 1.f / 3.f;
    p = p * w + 1.f;
    return (2.f * z) * p;
}
// sigma(y) and log sigma(-y) = -softplus(y), the declared forms: t = e^-|y|, u = 1 + t, sigma = 1 / u or t / u,
// softplus = max(y, 0) + log(u) + (t - (u - 1)) / u
__device__ __forceinline__ void det_sig_logsig(float y, float& sig, float& logsig_neg) {
    const float t = det_exp(-fabsf(y));
    const float u = 1.f + t;
    sig = (y >= 0.f) ? 1.f / u : t / u;
    logsig_neg = fminf(-y, 0.f) - (det_log12(u) + (t - (u - 1.f)) / u);
}
// (the two families behind one call: `det` is wave-uniform)
__device__ __forceinline__ void sig_logsig(bool det, float y, float& sig, float& logsig_neg) {
    if (det) det_sig_logsig(y, sig, logsig_neg); else fast_sig_logsig(y, sig, logsig_neg);
}
__device__ __forceinline__ float exp_any(bool det, float x) { return det ? det_exp(x) : fast_exp(x); }

// Coefficients of up to 64 rows at once: lane f holds the score n of one row (valid == false: no row).  Returns the
// lane's c1, c2 and the wave-uniform factor by which everything accumulated so far must be rescaled (self-adversarial
// loss: the running softmax maximum grew), 1 otherwise.
__device__ __forceinline__ float onepass_coeff(const amdkge_loss& L, float P, float n, bool valid, OnePassState& st,
                                               float& c1, float& c2, bool det = false) {
    c1 = 0.f;
    c2 = 0.f;
    float rescale = 1.f;
    switch (L.kind) {
        case AMDKGE_LOSS_PAIRWISE: c1 = (valid && (L.margin - P + n >= 0.f)) ? 1.f : 0.f; break;
        case AMDKGE_LOSS_NLL: {
            const bool in = valid && (n >= -75.f) && (n <= 75.f);
            float sg, lsn;
            sig_logsig(det, clip_exp(n), sg, lsn);
            st.Lw += valid ? -lsn : 0.f;   // softplus(clip n) = log(1 + exp(clip n))
            c1 = in ? sg : 0.f;
        } break;
        case AMDKGE_LOSS_ABSOLUTE_MARGIN: c1 = (valid && (L.margin + n >= 0.f)) ? 1.f : 0.f; break;
        case AMDKGE_LOSS_SELF_ADVERSARIAL: {
            const float x = L.alpha * n;
            const float gm = wave_max(valid ? x : -INFINITY);
            if (gm > st.m) {
                rescale = (st.m == -INFINITY) ? 0.f : exp_any(det, st.m - gm);   // nothing accumulated before the first group
                st.S *= rescale; st.Lw *= rescale; st.m = gm;
            }
            const float u = valid ? exp_any(det, x - st.m) : 0.f;
            float sg, ell;
            sig_logsig(det, n + L.margin, sg, ell);   // sigma(n + gamma), l = log sigma(-n - gamma)
            st.S += u; st.Lw += valid ? u * ell : 0.f;
            c1 = valid ? u * (sg - L.alpha * ell) : 0.f;
            c2 = u;
        } break;
        default: {
            const bool in = valid && (n >= -75.f) && (n <= 75.f);
            const float ex = valid ? exp_any(det, clip_exp(n)) : 0.f;
            st.Zs += ex;
            c1 = in ? ex : 0.f;
        } break;
    }
    return rescale;
}

// st holds the wave totals here (wave_sum of the per-lane partials)
__device__ __forceinline__ void onepass_kappa(const amdkge_loss& L, float P, int eta, const OnePassState& st, float& k1, float& k2, bool det = false) {
    const float feta = (float)eta;
    float red = L.reduction_mean ? feta : 1.f;
    k2 = 0.f;
    switch (L.kind) {
        case AMDKGE_LOSS_NLL: if (L.reduction_mean) red = 2.f * feta; k1 = 1.f / red; break;
        case AMDKGE_LOSS_SELF_ADVERSARIAL: k1 = 1.f / (st.S * red); k2 = L.alpha * (st.Lw / st.S) / (st.S * red); break;
        case AMDKGE_LOSS_MULTICLASS_NLL: {
            const float eP = exp_any(det, clip_exp(P));
            k1 = 1.f / ((st.Zs / red + eP) * red);
        } break;
        default: k1 = 1.f / red; break;
    }
}

// Loss.__call__ for one positive on the single-pass path: same outputs as loss_and_dscore (per-sample loss, dL/dpos,
// sn[j] <- dL/dneg_j) from the statistics the row loop already gathered (st = wave totals).
__device__ __forceinline__ void onepass_finish(const amdkge_loss& L, float P, float* sn, int eta, int lane,
                                               const OnePassState& st, float& per, float& dP, bool det = false) {
    const float feta = (float)eta;
    float red = L.reduction_mean ? feta : 1.f;
    switch (L.kind) {
        case AMDKGE_LOSS_NLL: {   // loss_functions.py:376-382
            if (L.reduction_mean) red = 2.f * feta;
            const bool inP = (P >= -75.f) && (P <= 75.f);
            float sgP, lsP;
            sig_logsig(det, -clip_exp(P), sgP, lsP);   // sigma(-Pc), log sigma(Pc)
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                float sg, ls;
                sig_logsig(det, clip_exp(n), sg, ls);
                sn[j] = ((n >= -75.f) && (n <= 75.f)) ? sg / red : masked_zero(n);
            }
            per = (feta * -lsP + st.Lw) / red;   // log(1+exp(-Pc)) = -log sigma(Pc)
            dP = inP ? -feta * sgP / red : 0.f;
        } break;
        case AMDKGE_LOSS_SELF_ADVERSARIAL: {   // :556-574
            const float lbar = st.Lw / st.S;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                const float w = exp_any(det, L.alpha * n - st.m) / st.S;
                float sg, ell;
                sig_logsig(det, n + L.margin, sg, ell);
                sn[j] = computed_zero((w * sg - L.alpha * w * (ell - lbar)) / red, n);
            }
            float sgP, lsP;
            sig_logsig(det, -(L.margin + P), sgP, lsP);   // sigma(-(gamma+P)), log sigma(gamma+P)
            per = -lsP - lbar / red;
            dP = -sgP;
        } break;
        default: {   // AMDKGE_LOSS_MULTICLASS_NLL :647-654
            const bool inP = (P >= -75.f) && (P <= 75.f);
            const float Pc = clip_exp(P);
            const float eP = exp_any(det, Pc);
            const float Z = st.Zs / red + eP;
            for (int j = lane; j < eta; j += KGE_WAVE) {
                const float n = sn[j];
                sn[j] = ((n >= -75.f) && (n <= 75.f)) ? exp_any(det, clip_exp(n)) / Z / red : masked_zero(n);
            }
            per = (det ? logf(Z) : __builtin_amdgcn_logf(Z) * 0.6931471805599453f) - Pc;   // -log(eP / Z)  (the loss VALUE only: nothing feeds back)
            dP = inP ? -1.f + eP / Z : 0.f;
        } break;
    }
}

// TransE keeps, per corruption and unit, only sign(s + p - o) for its backward pass (the gradient of |x|): 2 bits per unit,
// one byte per lane and quad, stashed in LDS by the scoring pass so that the replacement rows are read from memory ONCE
// (measured at the C2 shape, k = 200: forward kernel 66.7 -> 54 us).
__host__ __device__ inline size_t sign_stash_bytes(int model, int eta, int CH) {
    return model == AMDKGE_TRANSE ? (size_t)256 * eta * CH : 0;
}

__host__ __device__ inline size_t slot_lds_bytes(int eta, int W) {
    // neg[eta+1], repl[eta+1], keep[eta+1], then (8-byte aligned) part[W][eta+1] (W>1) or perm[eta+1] pairs of (corruption,
    // replacement row) (W==1), dfac[eta+1] (FocusE), rounded to 8 bytes
    const size_t b = (size_t)(eta + 1) * (3 + (W > 1 ? W : 2) + 1) * 4 + 8 + (W > 1 ? 64 * 4 : 0);   // (+ the single-pass cross-wave sums)
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
// (Forcing 4 waves/SIMD with __launch_bounds__(256, 4) was measured: no gain over the natural 3 -- the row gather is
// bound by fabric bandwidth, not by loads in flight -- and it costs spills.)
// DET: the declared (CPU-reproducible) transcendentals of the deterministic mode are a compile-time variant -- as a
// run-time branch they cost every instantiation 7 VGPRs, which took the C2 kernel from 168 to 175 registers, i.e. from
// 3 to 2 waves per SIMD (F 74 -> 84 us, measured in profiles/r04g_*).
template <int MODEL, int VEC, int W, int CH, bool STAGE = false, bool DET = false>
#ifdef KGE_F_WAVES   // development builds: force an occupancy
__attribute__((amdgpu_waves_per_eu(KGE_F_WAVES, KGE_F_WAVES)))
#else
// The deterministic ComplEx / HolE forward kernel sits 3 registers above the 168 that three waves per SIMD allow (171: two waves,
// F 89 us against the default mode's 74.5 at C2): ask for three -- the allocator finds them without scratch (checked by
// tests/test_kernel_resources.py).  Every other instantiation keeps the compiler's own choice (1 = no constraint).
__attribute__((amdgpu_waves_per_eu((DET && STAGE && MODEL == AMDKGE_COMPLEX && W == 1 && CH == 1) ? 3 : 1)))
#endif
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
    float* sh_part = reinterpret_cast<float*>(base + (((size_t)e1 * 12 + 7) & ~(size_t)7));
    float* sh_dfac = sh_part + (size_t)e1 * (W > 1 ? W : 2);
    // FocusE weights of this positive (uniform per slot)
    const int focus_nl = a.loss.focus_nonlinearity;
    float focus_wp = 1.f, focus_wn = 1.f;
    if (focus_nl) {
        const float wi = a.loss.d_focus_w[i];
        focus_wp = a.loss.focus_beta + (1.f - a.loss.focus_beta) * (1.f - wi);
        focus_wn = a.loss.focus_beta + (1.f - a.loss.focus_beta) * wi;
    }
    float dPfac = 1.f;
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
    // RotatE with the per-step phase table: the relation "row" is (cos, sin) as rel_phase_kernel computed them with the same
    // prep_rel -- one sincos per relation unit and step instead of one per positive and unit (libm range reduction: ~20 % of
    // the forward kernel's VALU work at k = 200)
    const bool rel_is_cs = (MODEL == AMDKGE_ROTATE) && a.rel_cs != nullptr;
    const float* rp = (rel_is_cs ? a.rel_cs : a.rel) + (int64_t)pp * a.K;
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
        if (!rel_is_cs) {
#pragma unroll
            for (int u = 0; u < VEC; ++u) prep_rel<MODEL>(a.mc, p[c][u]);
        }
    }
    // RotatE on a padded row layout: 1 for the zero-padding units behind k_live (see grad_unit); folds away elsewhere
    float pad1[CH][VEC];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int u = 0; u < VEC; ++u) pad1[c][u] = (MODEL == AMDKGE_ROTATE && qoff[c] + u >= a.k_live) ? 1.f : 0.f;
    if constexpr (STAGE) {
        // side rows for the owner kernel.  Trilinear models: d(score)/d(replaced row) does not depend on the
        // replaced row, so the owner only needs g * A (A = d/do (s,p)) or g * B (B = d/ds (p,o)).
        // TransE: copies of s and o (the owner recomputes grad_unit with its own row); RotatE: s and o rotated onto the replaced row.
        static_assert(!STAGE || VEC == 4, "staging uses the 16-byte layout");
        constexpr bool TRILINEAR = (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX);
        if (active && !KGE_DBG(a, 64)) {
            float* qa = a.stage_rows + ((int64_t)i * a.ns + 2) * a.K;
            float* qb = a.stage_rows + ((int64_t)i * a.ns + 3) * a.K;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (!qok[c]) continue;
                float va[NC][4], vb[NC][4];
#pragma unroll
                for (int u = 0; u < VEC; ++u) {
                    if constexpr (TRILINEAR) {
                        float ds[NC], dp[NC], dd[NC];
                        grad_unit<MODEL>(s[c][u], p[c][u], o[c][u], 1.f, ds, dp, dd);
#pragma unroll
                        for (int h = 0; h < NC; ++h) { va[h][u] = dd[h]; vb[h][u] = ds[h]; }
                    } else if constexpr (MODEL == AMDKGE_ROTATE) {
                        // A = s o r (the reference's own first step of s o r - e, RotatE.py:100-101: the object-side entries of
                        // the tile pass are bit-identical to grad_unit), B = o o conj(r): |e o r - o| = |e - B| as |r| = 1, and
                        // d|e o r - o| / de = (e - B) / |e - B| -- one side row and the tile's own row per entry, no relation row
                        const float cs = p[c][u][0], sn = p[c][u][1];
                        va[0][u] = s[c][u][0] * cs - s[c][u][1] * sn; va[1][u] = s[c][u][0] * sn + s[c][u][1] * cs;
                        vb[0][u] = o[c][u][0] * cs + o[c][u][1] * sn; vb[1][u] = o[c][u][1] * cs - o[c][u][0] * sn;
                    } else {
#pragma unroll
                        for (int h = 0; h < NC; ++h) { va[h][u] = s[c][u][h]; vb[h][u] = o[c][u][h]; }
                    }
                }
#pragma unroll
                for (int h = 0; h < NC; ++h) {
                    *reinterpret_cast<float4*>(qa + qoff[c] + h * a.k) = make_float4(va[h][0], va[h][1], va[h][2], va[h][3]);
                    *reinterpret_cast<float4*>(qb + qoff[c] + h * a.k) = make_float4(vb[h][0], vb[h][1], vb[h][2], vb[h][3]);
                }
            }
        }
    }
    slot_sync<W>();

    const float sgn_scale = a.mc.score_sign * a.mc.score_scale;

    // Rows in flight per slot: the replacement rows of PF corruptions are requested together (the row loads are
    // dependent on nothing but LDS-resident ids), then reduced one by one.  One row at a time left the kernel
    // latency-bound (21 serial L2/MALL round trips per pass per positive).
#ifndef KGE_PF
#define KGE_PF 6
#endif
    // (single-pass RotatE keeps the rows as unit vectors next to two complex accumulators per side: with 3 rows in flight it
    // fits 3 waves per SIMD -- measured 85.8 us against 95.1 with 6 and 92.4 with 2)
    constexpr int PF = (CH * NC * VEC <= 8) ? (MODEL == AMDKGE_ROTATE && STAGE ? 3 : KGE_PF) : 2;
    auto load_row = [&](const float* re, float (&e)[CH][VEC][NC]) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<VEC> ve = ldg<VEC>(re + qoff[c] + h * a.k);
#pragma unroll
                for (int u = 0; u < VEC; ++u) e[c][u][h] = ve.v[u];
            }
    };

    // ---- pass 1: scores (positive as j == -1) -------------------------------------------------
    // (TransE: one quad per lane only -- with two, the single pass measured 145 us against the stash form's 81 at k = 352:
    // register pressure leaves it 2 waves per SIMD)
    // All but TransE also when the four waves of a workgroup share the positive (k > 512: the C5 row width): per group of rows the
    // waves' partial sums meet in LDS, every wave then evaluates the same coefficients -- one barrier per group, and the rows
    // (8 KB each at k = 1000, eta = 64 of them per positive) are read once instead of twice
    // MULTI: the PF row sums of a group in ONE transposing wave reduction and the group's (corruption, row) pairs in one LDS read
    // (round 5).  A quarter fewer VALU instructions in the row loop bought DistMult 2 - 4 % and TransE / ComplEx nothing
    // (profiles/r05j_*): the loop waits on its row gathers.  ComplEx and RotatE sit at the 168-register edge of three waves per
    // SIMD: there the form needs parked dwords (scratch), and with them C4 measured 0.408 ms against 0.384 (profiles/r05l_*) --
    // those instantiations keep one wave_sum per row.  The deterministic ComplEx kernel (already parked, 2.8 % faster) takes it.
    constexpr bool MULTI = W == 1 && (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_TRANSE || (DET && MODEL == AMDKGE_COMPLEX));
    constexpr bool ONEPASS = STAGE && (MODEL == AMDKGE_DISTMULT || MODEL == AMDKGE_COMPLEX || MODEL == AMDKGE_ROTATE ||
                                       (MODEL == AMDKGE_TRANSE && W == 1 && CH == 1));
    // TransE outside the single-pass geometry (two quads per lane, rows shared by four waves, atomic path): signs stashed by the
    // scoring pass
    constexpr bool SIGNSTASH = (MODEL == AMDKGE_TRANSE) && !ONEPASS && (STAGE || VEC == 4);   // (the scalar-load geometries keep the two-pass form)
    unsigned char* sh_sign = reinterpret_cast<unsigned char*>(smem) + a.sign_off;
    float av1[2][CH][VEC][NC], av2[2][CH][VEC][NC];   // [0]: sum c_j e_j over object-replaced rows, [1]: subject-replaced
    OnePassState ops{-INFINITY, 0.f, 0.f, 0.f};
    if constexpr (ONEPASS) {
        // Single pass (see OnePassState): scores as dot products with the side rows A = d/do(s,p), B = d/ds(p,o)
        // (score(s,p,e) = <A, e>, score(e,p,o) = <B, e>: the query-vector form the reference itself uses for
        // corruption scores, ComplEx.py:93-107,138-150), and sum_j c_j e_j accumulated while the rows stream by.
        // s, p, o are dead inside the loop (reloaded afterwards): the loop lives on A, B, PF rows and 4 accumulators.
        //
        // TransE (score = -sum |d|, d = s + p - o): d(score)/d(s, p, o) = -/+ sign(d_j), so what must be accumulated per
        // side is sum_j c_j sign(d_j) -- the coefficient with the sign bit of d_j xor-ed in (one v_bitop3 per unit), no
        // second pass and no stash.  sign(0) = 0 is kept exact: a group of rows with an exact zero in a live unit (a
        // wave-uniform test on compare masks) takes the select form instead.
        //
        // RotatE (score = -sum |z|, z = s o r - o): what the gradients of s, theta and o need from corruption j is the UNIT VECTOR
        // z_j / |z_j| per unit, known when the row streams by.  sum_j c_j z_j/|z_j| is accumulated per side and turned into the
        // row gradients once per positive (they are linear in it: d/ds = conj(r) o Z_obj, d/do = -Z_subj,
        // d/dtheta = Im(conj(A) Z_obj) + Im(conj(o) Z_subj) with A = s o r) -- one sqrt and one rcp per unit and row instead of
        // grad_unit's two evaluations, and no second read of the rows.  Object-side z_j = A - e_j and subject-side
        // z_j = e_j o r - o are the reference's own operations (RotatE.py:96-104).
        float qa[CH][VEC][NC], qb[CH][VEC][NC];
        float zmark[CH][VEC];   // TransE: what an exact zero of d compares equal to -- 0 in a unit of the model, NaN (never equal) in row
                                // padding and idle lanes.  (Per-unit lane masks did the same from 8 SGPRs and spilled them.)
        // this positive's block of the sign codes ([eta][nq] dwords; eta * nq < 2^23, so row offsets are 32-bit)
        uint32_t* const code_base = (MODEL == AMDKGE_TRANSE && a.sign_codes != nullptr)
                                        ? a.sign_codes + (int64_t)__builtin_amdgcn_readfirstlane((int)i) * eta * a.nq : nullptr;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                float ds[NC], dp[NC], dd[NC];
                if constexpr (MODEL == AMDKGE_TRANSE) {
                    ds[0] = 0.f; dd[0] = s[c][u][0] + p[c][u][0];   // the reference's first rounding of (s + p) - e, TransE.py:51-53
                    zmark[c][u] = (qok[c] && qoff[c] + u < a.k_live) ? 0.f : __builtin_nanf("");
                } else if constexpr (MODEL == AMDKGE_ROTATE) {
                    ds[0] = 0.f; ds[1] = 0.f;   // A = s o r (p holds cos, sin)
                    dd[0] = s[c][u][0] * p[c][u][0] - s[c][u][1] * p[c][u][1];
                    dd[1] = s[c][u][0] * p[c][u][1] + s[c][u][1] * p[c][u][0];
                } else {
                    grad_unit<MODEL>(s[c][u], p[c][u], o[c][u], 1.f, ds, dp, dd);
                }
#pragma unroll
                for (int h = 0; h < NC; ++h) {
                    qa[c][u][h] = dd[h]; qb[c][u][h] = ds[h];
                    av1[0][c][u][h] = 0.f; av1[1][c][u][h] = 0.f; av2[0][c][u][h] = 0.f; av2[1][c][u][h] = 0.f;
                }
                acc += score_unit<MODEL, DET>(s[c][u], p[c][u], o[c][u]);   // the positive keeps the reference's op order
            }
            part += qok[c] ? acc : 0.f;
        }
        // cross-wave sums (W > 1): [2][W][PF] partials of the row groups, double-buffered by group parity, then [W] for the positive
        float* sh_red = sh_dfac + e1;
        int grp_no = 0;
        float P1;
        if constexpr (W == 1) {
            P1 = sgn_scale * wave_sum(part);
        } else {
            const float w0 = wave_sum(part);
            if (lane == 0) sh_red[2 * W * PF + wv] = w0;
            __syncthreads();
            float t0 = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) t0 += sh_red[2 * W * PF + w];
            P1 = sgn_scale * t0;
        }
        if (focus_nl) focus_apply(focus_nl, P1, focus_wp, P1, dPfac);
        if (lane == 0) sh_neg[eta] = P1;
        // corruptions ordered by side (object-replaced first): each of the two row loops below then has a
        // compile-time side, i.e. fixed accumulators and no per-row selects
        int* sh_perm = reinterpret_cast<int*>(sh_part);
        int nkeep = 0;
        for (int jb = 0; jb < eta; jb += KGE_WAVE) {
            const int j = jb + lane;
            nkeep += __popcll(__ballot(j < eta && sh_keep[j] != 0));
        }
        {
            int offk = 0, offn = nkeep;
            for (int jb = 0; jb < eta; jb += KGE_WAVE) {
                const int j = jb + lane;
                const bool valid = j < eta;
                const bool k = valid && sh_keep[j] != 0;
                const unsigned long long mk = __ballot(k), mn = __ballot(valid && !k);
                const unsigned long long mine = k ? mk : mn;
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0));
                if (valid) {
                    const int pos = (k ? offk : offn) + before;
                    if constexpr (MULTI) *reinterpret_cast<int2*>(sh_perm + 2 * pos) = make_int2(j, sh_repl[j]);
                    else sh_perm[pos] = j;
                }
                offk += __popcll(mk); offn += __popcll(mn);
            }
        }
        slot_sync<W>();
        // MULTI: which of a group's PF row sums this lane ends up with (first eight lanes only: the others would count rows twice)
        [[maybe_unused]] const WaveMultiSel msel = wave_multi_sel(lane);
        [[maybe_unused]] const int mslot = lane < 8 ? wave_multi_slot(lane) : 8;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int p_begin = d == 0 ? 0 : nkeep, p_end = d == 0 ? nkeep : eta;
            for (int p0 = p_begin; p0 < p_end; p0 += PF) {
                float e[PF][CH][VEC][NC];
                int jv[PF];
                float nv = 0.f;   // the lane of row p0 + f (MULTI: wave_multi_lane(f), else lane f): the row's score
                int jl = 0;       // ... and its corruption index
                if constexpr (MULTI) {
                    // one LDS read for the group: every lane takes the (corruption, replacement row) pair of ITS row -- the one whose
                    // sum the transposing reduction below leaves in this lane --, the row addresses go through SGPRs
                    const int2 pe = *reinterpret_cast<const int2*>(sh_perm + 2 * min(p0 + min(mslot, PF - 1), p_end - 1));   // past the end: the last row again
                    jl = pe.x;
#pragma unroll
                    for (int f = 0; f < PF; ++f) {
                        jv[f] = __builtin_amdgcn_readlane(pe.x, wave_multi_lane(f));
                        load_row(a.ent + (int64_t)(KGE_DBG(a, 8) ? ps : __builtin_amdgcn_readlane(pe.y, wave_multi_lane(f))) * a.K, e[f]);   // (ablation 8: cache-hot row)
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < PF; ++f) {
                        jv[f] = __builtin_amdgcn_readfirstlane(sh_perm[min(p0 + f, p_end - 1)]);   // past the end: reload the last row
                        load_row(a.ent + (int64_t)(KGE_DBG(a, 8) ? ps : __builtin_amdgcn_readfirstlane(sh_repl[jv[f]])) * a.K, e[f]);
                    }
                }
                [[maybe_unused]] float accv[PF];
                unsigned long long zero_m = 0ull;   // TransE: lanes holding an exact zero of d in a live unit of this group
#pragma unroll
                for (int f = 0; f < PF; ++f) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        float t = 0.f;
                        if constexpr (MODEL == AMDKGE_TRANSE) {
#pragma unroll
                            for (int u = 0; u < VEC; ++u) {
                                // d of the corrupted triple in the reference's operation order; kept in place of the row
                                const float dj = (d == 0) ? (qa[c][u][0] - e[f][c][u][0]) : ((e[f][c][u][0] + p[c][u][0]) - o[c][u][0]);
                                e[f][c][u][0] = dj;
                                t += fabsf(dj);
                                zero_m |= __ballot(dj == zmark[c][u]);
                            }
                            if (a.sign_codes && active && p0 + f < p_end && qok[c]) {
                                // top bytes (sign + 7 exponent bits) of the four d values of this lane, one dword: 3 v_perm_b32
                                const unsigned t01 = __builtin_amdgcn_perm(__float_as_uint(e[f][c][1][0]), __float_as_uint(e[f][c][0][0]), 0x0c0c0703u);
                                const unsigned t23 = __builtin_amdgcn_perm(__float_as_uint(e[f][c][3][0]), __float_as_uint(e[f][c][2][0]), 0x07030c0cu);
                                // (wave-uniform row pointer + the lane's quad index: no per-lane address arithmetic)
                                uint32_t* crow = code_base + (uint32_t)(jv[f] * a.nq);
                                crow[qoff[c] >> 2] = t01 | t23;
                            }
                        } else if constexpr (MODEL == AMDKGE_ROTATE) {
#pragma unroll
                            for (int u = 0; u < VEC; ++u) {
                                float zr, zi;
                                if (d == 0) {
                                    zr = qa[c][u][0] - e[f][c][u][0]; zi = qa[c][u][1] - e[f][c][u][1];
                                } else {
                                    // (the reference's operations; e - o o conj(r) would save the product: measured 2 %)
                                    zr = e[f][c][u][0] * p[c][u][0] - e[f][c][u][1] * p[c][u][1] - o[c][u][0];
                                    zi = e[f][c][u][0] * p[c][u][1] + e[f][c][u][1] * p[c][u][0] - o[c][u][1];
                                }
                                const float m = kge_sqrt_t<DET>(zr * zr + zi * zi);
                                t += m;
                                const float inv = kge_div_t<DET>(1.f, m + pad1[c][u]);   // m == 0 in a unit of the model: NaN, like the reference
                                e[f][c][u][0] = zr * inv; e[f][c][u][1] = zi * inv;   // the unit vector, kept in place of the row
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < VEC; ++u)
#pragma unroll
                                for (int h = 0; h < NC; ++h) t = fmaf(d == 0 ? qa[c][u][h] : qb[c][u][h], e[f][c][u][h], t);
                        }
                        // (t is never -0: it starts at +0 and only fma / |.| / sqrt results are added -- 0 + t == t bit for bit)
                        if constexpr (MULTI && CH == 1) acc = qok[c] ? t : 0.f; else acc += qok[c] ? t : 0.f;
                    }
                    if constexpr (MULTI) {
                        accv[f] = acc;
                    } else if constexpr (W == 1) {
                        const float n = sgn_scale * wave_sum(acc);
                        nv = (lane == f) ? n : nv;
                        jl = (lane == f) ? jv[f] : jl;
                    } else {
                        const float w1 = wave_sum(acc);
                        if (lane == 0) sh_red[((grp_no & 1) * W + wv) * PF + f] = w1;
                        jl = (lane == f) ? jv[f] : jl;
                    }
                }
                if constexpr (MULTI) {
                    // the PF row sums in one transposing reduction (same additions as wave_sum: kge_device.h)
                    nv = sgn_scale * wave_sum_multi<PF>(accv, msel);
                }
                // TransE: a NaN unit of d makes its row's score NaN (a sum of |d|), and c * sign(NaN) must be NaN as tf.sign gives it:
                // such a group takes the select form too, which hands a NaN d through
                if constexpr (MODEL == AMDKGE_TRANSE) zero_m |= __ballot(nv != nv);
                if constexpr (W > 1) {
                    __syncthreads();   // (the other buffer is free again: every wave passed the previous group's barrier after reading it)
                    if (lane < PF) {
                        float t1 = 0.f;
#pragma unroll
                        for (int w = 0; w < W; ++w) t1 += sh_red[((grp_no & 1) * W + w) * PF + lane];
                        nv = sgn_scale * t1;
                    }
                    ++grp_no;
                }
                const bool lane_valid = (MULTI ? mslot : lane) < min(PF, p_end - p0);
                float dfl = 1.f;
                if (focus_nl) focus_apply(focus_nl, nv, focus_wn, nv, dfl);
                if (lane_valid) {
                    sh_neg[jl] = nv;
                    if (focus_nl) sh_dfac[jl] = dfl;
                }
                float c1l, c2l;
                const float rs = onepass_coeff(a.loss, P1, nv, lane_valid, ops, c1l, c2l, DET);
                c1l *= dfl; c2l *= dfl;   // d(neg')/d(neg) folded into the accumulation weights
                if (rs != 1.f) {   // the running softmax maximum grew: rescale what has been accumulated
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int u = 0; u < VEC; ++u)
#pragma unroll
                                for (int h = 0; h < NC; ++h) { av1[dd][c][u][h] *= rs; av2[dd][c][u][h] *= rs; }
                }
                const bool two = a.loss.kind == AMDKGE_LOSS_SELF_ADVERSARIAL;   // only this loss has a second coefficient
#pragma unroll
                for (int f = 0; f < PF; ++f) {
                    // rows past the end had invalid lanes: their coefficients are 0 and add nothing
                    const int fl = MULTI ? wave_multi_lane(f) : f;
                    const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c1l), fl));
                    const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c2l), fl));
                    if constexpr (MODEL == AMDKGE_TRANSE) {
                        if (zero_m == 0ull) {   // c * sign(d) for d != 0: the coefficient with d's sign bit xor-ed in
#pragma unroll
                            for (int c = 0; c < CH; ++c)
#pragma unroll
                                for (int u = 0; u < VEC; ++u) {
                                    // (d & sign bit) ^ c in one v_bitop3_b32 (truth table 0x6a = (a & b) ^ c)
                                    const unsigned db = __float_as_uint(e[f][c][u][0]);
                                    av1[d][c][u][0] += __uint_as_float(__builtin_amdgcn_bitop3_b32(db, 0x80000000u, __float_as_uint(c1), 0x6a));
                                    if (two) av2[d][c][u][0] += __uint_as_float(__builtin_amdgcn_bitop3_b32(db, 0x80000000u, __float_as_uint(c2), 0x6a));
                                }
                        } else {
#pragma unroll
                            for (int c = 0; c < CH; ++c)
#pragma unroll
                                for (int u = 0; u < VEC; ++u) {
                                    const float dj = e[f][c][u][0];
                                    const float sg = (dj > 0.f) ? 1.f : ((dj < 0.f) ? -1.f : dj);   // (+-0 or NaN)
                                    av1[d][c][u][0] += c1 * sg;
                                    if (two) av2[d][c][u][0] += c2 * sg;
                                }
                        }
                        continue;
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int u = 0; u < VEC; ++u)
#pragma unroll
                            for (int h = 0; h < NC; ++h) av1[d][c][u][h] = fmaf(c1, e[f][c][u][h], av1[d][c][u][h]);
                    if (two) {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int u = 0; u < VEC; ++u)
#pragma unroll
                                for (int h = 0; h < NC; ++h) av2[d][c][u][h] = fmaf(c2, e[f][c][u][h], av2[d][c][u][h]);
                    }
                }
            }
        }
        ops.S = wave_sum(ops.S); ops.Lw = wave_sum(ops.Lw); ops.Zs = wave_sum(ops.Zs);
        // s, p, o again for the gradient transform below (L2-hot)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                const fvec<VEC> vs = ldg<VEC>(rs + qoff[c] + h * a.k);
                const fvec<VEC> vp = ldg<VEC>(rp + qoff[c] + h * a.k);
                const fvec<VEC> vo = ldg<VEC>(ro + qoff[c] + h * a.k);
#pragma unroll
                for (int u = 0; u < VEC; ++u) { s[c][u][h] = vs.v[u]; p[c][u][h] = vp.v[u]; o[c][u][h] = vo.v[u]; }
            }
            if (!rel_is_cs) {
#pragma unroll
                for (int u = 0; u < VEC; ++u) prep_rel<MODEL>(a.mc, p[c][u]);
            }
        }
    }
    for (int j0 = -1; j0 < (ONEPASS ? -1 : eta); j0 += PF) {
        float e[PF][CH][VEC][NC];
        int keepv[PF];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int j = min(j0 + f, eta - 1);   // past the end: reload the last row, result unused
            keepv[f] = (j < 0) ? 1 : __builtin_amdgcn_readfirstlane(sh_keep[j]);
            const int64_t er = (j < 0) ? (int64_t)po : (int64_t)__builtin_amdgcn_readfirstlane(sh_repl[j]);
            load_row(a.ent + er * a.K, e[f]);
        }
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int j = j0 + f;
            if (j >= eta) break;
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float acc = 0.f;
                if constexpr (SIGNSTASH) {
                    // TransE.py:51-53 with the signs of s + p - o kept for the backward pass (same operations as score_unit)
                    unsigned code = 0;
                    float dv[VEC];
#pragma unroll
                    for (int u = 0; u < VEC; ++u) {
                        const float d = keepv[f] ? (s[c][u][0] + p[c][u][0] - e[f][c][u][0]) : (e[f][c][u][0] + p[c][u][0] - o[c][u][0]);
                        dv[u] = d;
                        acc += fabsf(d);
                        code |= ((d > 0.f) ? 1u : ((d < 0.f) ? 2u : ((d == d) ? 0u : 3u))) << (2 * u);   // 3: NaN
                    }
                    if (j >= 0) sh_sign[((size_t)j * CH + c) * 256 + tid] = (unsigned char)code;
                    if constexpr (STAGE && VEC == 4) {   // the tile pass's copy of the signs (see ENTRY_J_SHIFT)
                        if (a.sign_codes && active && j >= 0 && qok[c])
                            a.sign_codes[((int64_t)i * eta + j) * a.nq + (qoff[c] >> 2)] =
                                __builtin_amdgcn_perm(__float_as_uint(dv[1]), __float_as_uint(dv[0]), 0x0c0c0703u) |
                                __builtin_amdgcn_perm(__float_as_uint(dv[3]), __float_as_uint(dv[2]), 0x07030c0cu);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < VEC; ++u)
                        acc += keepv[f] ? score_unit<MODEL, DET>(s[c][u], p[c][u], e[f][c][u]) : score_unit<MODEL, DET>(e[f][c][u], p[c][u], o[c][u]);
                }
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
    }
    if constexpr (W > 1 && !ONEPASS) {
        __syncthreads();
        for (int j = ts; j < e1; j += TS) {
            float t2 = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) t2 += sh_part[w * e1 + j];
            sh_neg[j] = sgn_scale * t2;
        }
    }
    slot_sync<W>();
    if constexpr (!ONEPASS) {
        if (focus_nl) {   // FocusE on the two-pass path: transform the scores in place, keep d(score')/d(score)
            for (int j = ts; j < eta; j += TS) {
                float y, dfc;
                focus_apply(focus_nl, sh_neg[j], focus_wn, y, dfc);
                sh_neg[j] = y; sh_dfac[j] = dfc;
            }
            float y;
            focus_apply(focus_nl, sh_neg[eta], focus_wp, y, dPfac);
            slot_sync<W>();
            if (ts == 0) sh_neg[eta] = y;
            slot_sync<W>();
        }
    }
    const float P = sh_neg[eta];

    if (active && a.neg_scores)
        for (int j = ts; j < eta; j += TS) a.neg_scores[(int64_t)j * a.B + i] = sh_neg[j];
    if (active && a.pos_scores && ts == 0) a.pos_scores[i] = P;
    if constexpr (W > 1) __syncthreads();  // sh_neg is rewritten below by wave 0

    // ---- loss + dL/dscore (a12-a16): wave 0 of the slot, results through LDS -------------------
    float per = 0.f, dP = 0.f;
    if constexpr (ONEPASS) {
        // pairwise / absolute_margin have no transcendental and keep the generic evaluation
        if (a.loss.kind == AMDKGE_LOSS_PAIRWISE || a.loss.kind == AMDKGE_LOSS_ABSOLUTE_MARGIN) { if (W == 1 || wv == 0) loss_and_dscore(a.loss, P, sh_neg, eta, lane, per, dP); }
        else if (W == 1 || wv == 0) onepass_finish(a.loss, P, sh_neg, eta, lane, ops, per, dP, DET);   // (rewrites sh_neg in place: one wave)
    } else if (W == 1 || wv == 0) loss_and_dscore(a.loss, P, sh_neg, eta, lane, per, dP);
    if constexpr (W > 1) {
        if (wv == 0 && lane == 0) sh_part[0] = dP;
        __syncthreads();
        dP = sh_part[0];
    } else {
        slot_sync<W>();
    }
    if (focus_nl) {   // chain rule through the FocusE transform
        dP *= dPfac;
        for (int j = ts; j < eta; j += TS) sh_neg[j] *= sh_dfac[j];
        slot_sync<W>();
    }
    if (ts == 0) sh_loss[slot] = active ? (double)per : 0.0;
    if constexpr (STAGE) {
        // one entry per row gradient that lands in the entity table, into the bucket of the owning tile
        if (active && !KGE_DBG(a, 32))
            for (int j = ts; j < eta + (a.pos_atomic ? 0 : 2); j += TS) {
                uint32_t dest, role;
                float g, coeff = 1.f;
                if (j < eta) {
                    dest = (uint32_t)sh_repl[j]; role = sh_keep[j] ? 0u : 1u; coeff = sh_neg[j]; g = coeff * sgn_scale;
                    if (a.sign_codes) role |= (uint32_t)j << ENTRY_J_SHIFT;   // (bits above the local row)
                }
                else { dest = (uint32_t)(j == eta ? ps : po); role = (j == eta) ? 2u : 3u; g = 1.f; }
                // inactive margin / clipped corruption / a coefficient that underflows fp32: no entry.  The threshold is the smallest
                // NORMAL number, not zero: what the hardware transcendentals leave in the denormal range depends on the instruction
                // sequence, and touched-rows mode (amdkge_opt.lazy) defines "touched" through this line (oracle touched_rows).
                // Kept: NaN coefficients and the masked zeros of non-finite scores (masked_zero above).
                if (!entry_wanted(coeff, g)) continue;
                if (j >= eta && a.hot_map && a.hot_map[dest]) continue;   // hot row: went to its replicas (below), no entry
                uint32_t tile, local;
                tile_of_row(dest, (uint32_t)a.st_n_tiles, (uint32_t)a.st_rb, tile, local);   // block-interleaved ownership, see tile_backward_kernel
                StageEntry en{(uint32_t)i, role | (local << 2), g, dest};   // local < 4096: below the corruption index
                const int slotpos = atomicAdd(a.st_counters + (size_t)tile * 32, 1);
                if (slotpos < a.st_cap) {
                    a.st_lists[(size_t)tile * a.st_cap + slotpos] = en;
                } else {
                    const int op = atomicAdd(a.st_counters + (size_t)a.st_n_tiles * 32, 1);
                    if (op < a.st_ovf_cap) a.st_ovf[op] = en;
                }
            }
    }

    // ---- pass 2: gradients -------------------------------------------------------------------
    float gs[CH][VEC][NC], gp[CH][VEC][NC], go[CH][VEC][NC];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            float ds[NC], dp[NC], dd[NC];
            grad_unit<MODEL, DET>(s[c][u], p[c][u], o[c][u], dP * sgn_scale, ds, dp, dd, pad1[c][u]);
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
        } else if constexpr (W == 1) {
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
        } else {
            // one positive per workgroup (W == 4): the row is transposed through ONE LDS row shared by the four waves.
            // Called by every thread of the workgroup (SLOTS == 1, so `active` is workgroup-uniform).
            static_assert(VEC == 1 || W == 1 || W == 4, "multi-wave LDS-transposed emit needs one slot per workgroup");
            float* srow = reinterpret_cast<float*>(smem + (size_t)SLOTS * per_slot + SLOTS * sizeof(double));
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < NC; ++h)
                    if (qok[c])
                        *reinterpret_cast<float4*>(srow + qoff[c] + h * a.k) =
                            make_float4(gr[c][0][h] * mul, gr[c][1][h] * mul, gr[c][2][h] * mul, gr[c][3][h] * mul);
            __syncthreads();
            for (int idx = ts; idx < nfloats; idx += TS) atomic_add_f32(grow + idx, srow[idx]);
            __syncthreads();
        }
    };

    if constexpr (ONEPASS) {
        // E_side = sum_j g_j e_j (g_j incl. score scale) from the two coefficient sums; the gradients of s, p, o are
        // linear in it
        float k1, k2;
        onepass_kappa(a.loss, P, eta, ops, k1, k2, DET);
        k1 *= sgn_scale; k2 *= sgn_scale;
        // A NaN factor (multiclass_nll: Z is NaN as soon as one score of the positive is; self_adversarial: the softmax sum) times a
        // side's sum of coefficients.  The reference ADDS, per corruption, coefficient x Jacobian: a side whose corruptions were all
        // masked out (clipped / NaN scores: exact zeros) or that has no corruption at all contributes an exact zero, not NaN x 0.
        // Rare and wave-uniform: the side's coefficients are read back from LDS (sh_neg holds dL/dneg) instead of being tracked in the row loop.
        float k1s[2] = {k1, k1}, k2s[2] = {k2, k2};
        if (k1 != k1 || k2 != k2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                bool any = false;
                for (int j = lane; j < eta; j += KGE_WAVE) {
                    const float cj = sh_neg[j];
                    any |= ((sh_keep[j] != 0) == (d == 0)) && !(cj == 0.f);   // (+-0: masked; NaN counts)
                }
                if (__ballot(any) == 0ull) { k1s[d] = 0.f; k2s[d] = 0.f; }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                float eo[NC], es[NC], ds[NC], dp[NC], dd[NC];
#pragma unroll
                for (int h = 0; h < NC; ++h) {
                    eo[h] = k1s[0] * av1[0][c][u][h] + k2s[0] * av2[0][c][u][h];
                    es[h] = k1s[1] * av1[1][c][u][h] + k2s[1] * av2[1][c][u][h];
                }
                if constexpr (MODEL == AMDKGE_TRANSE) {
                    // eo / es = sum_j g'_j sign(d_j) per side (g' includes the score sign); padding units stay zero
                    const bool lv = qoff[c] + u < a.k_live;
                    const float Go = lv ? eo[0] : 0.f, Gs = lv ? es[0] : 0.f;
                    gs[c][u][0] += Go; gp[c][u][0] += Go + Gs; go[c][u][0] -= Gs;
                    continue;
                }
                if constexpr (MODEL == AMDKGE_ROTATE) {
                    // eo = Z_obj, es = Z_subj (sums of g'_j z_j / |z_j|, g' includes the score sign); p = (cos, sin)
                    const float cs = p[c][u][0], sn = p[c][u][1];
                    const float Ar = s[c][u][0] * cs - s[c][u][1] * sn, Ai = s[c][u][0] * sn + s[c][u][1] * cs;
                    gs[c][u][0] += eo[0] * cs + eo[1] * sn;          // conj(r) o Z_obj
                    gs[c][u][1] += eo[1] * cs - eo[0] * sn;
                    go[c][u][0] -= es[0]; go[c][u][1] -= es[1];
                    gp[c][u][0] += (eo[1] * Ar - eo[0] * Ai) + (es[1] * o[c][u][0] - es[0] * o[c][u][1]);   // d/dphase
                    continue;
                }
                grad_unit<MODEL>(s[c][u], p[c][u], eo, 1.f, ds, dp, dd);   // corruptions (s, p, e_j)
#pragma unroll
                for (int h = 0; h < NC; ++h) { gs[c][u][h] += ds[h]; gp[c][u][h] += dp[h]; }
                grad_unit<MODEL>(es, p[c][u], o[c][u], 1.f, ds, dp, dd);   // corruptions (e_j, p, o)
#pragma unroll
                for (int h = 0; h < NC; ++h) { go[c][u][h] += dd[h]; gp[c][u][h] += dp[h]; }
            }
    }
    const bool do_neg_atomics = active && !KGE_DBG(a, 1);
    if constexpr (SIGNSTASH) {
        // TransE backward from the stashed signs: no second read of the replacement rows.  Same additions, in the same
        // order (j ascending), as the generic loop below performs through grad_unit.
        for (int j = 0; j < eta; ++j) {
            const int keep = __builtin_amdgcn_readfirstlane(sh_keep[j]);
            const float g = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh_neg[j]))) * sgn_scale;
            float gr[CH][VEC][NC];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const unsigned code = sh_sign[((size_t)j * CH + c) * 256 + tid];
#pragma unroll
                for (int u = 0; u < VEC; ++u) {
                    const unsigned b = (code >> (2 * u)) & 3u;
                    const float sg = (b == 1u) ? g : ((b == 2u) ? -g : ((b == 3u) ? __builtin_nanf("") : 0.f));   // g * sign(d); sign(NaN) = NaN
                    if (keep) { gs[c][u][0] += sg; gp[c][u][0] += sg; gr[c][u][0] = -sg; }
                    else { go[c][u][0] += -sg; gp[c][u][0] += sg; gr[c][u][0] = sg; }
                }
            }
            if constexpr (!STAGE) {
                if (do_neg_atomics) emit_row(a.g_ent + (int64_t)__builtin_amdgcn_readfirstlane(sh_repl[j]) * a.K, gr, a.K, 1.f);
            }
        }
    }
    for (int j0 = 0; j0 < ((ONEPASS || SIGNSTASH || KGE_DBG(a, 4)) ? 0 : eta); j0 += PF) {
        float e[PF][CH][VEC][NC];
        int keepv[PF];
        int64_t erv[PF];
        float gv[PF];
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const int j = min(j0 + f, eta - 1);
            keepv[f] = __builtin_amdgcn_readfirstlane(sh_keep[j]);
            erv[f] = (int64_t)__builtin_amdgcn_readfirstlane(sh_repl[j]);
            gv[f] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh_neg[j]))) * sgn_scale;
            load_row(a.ent + erv[f] * a.K, e[f]);
        }
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            if (j0 + f >= eta) break;
            const float g = gv[f];
            float gr[CH][VEC][NC];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int u = 0; u < VEC; ++u) {
                    float ds[NC], dp[NC], dd[NC];
                    if (keepv[f]) {   // (s, p, e): object replaced
                        grad_unit<MODEL, DET>(s[c][u], p[c][u], e[f][c][u], g, ds, dp, dd, pad1[c][u]);
#pragma unroll
                        for (int h = 0; h < NC; ++h) { gs[c][u][h] += ds[h]; gp[c][u][h] += dp[h]; gr[c][u][h] = dd[h]; }
                    } else {          // (e, p, o): subject replaced
                        grad_unit<MODEL, DET>(e[f][c][u], p[c][u], o[c][u], g, ds, dp, dd, pad1[c][u]);
#pragma unroll
                        for (int h = 0; h < NC; ++h) { go[c][u][h] += dd[h]; gp[c][u][h] += dp[h]; gr[c][u][h] = ds[h]; }
                    }
                }
            }
            if constexpr (!STAGE) { if (do_neg_atomics) emit_row(a.g_ent + erv[f] * a.K, gr, a.K, 1.f); }
        }
    }

    // ---- per-block loss: one fp64 atomic -------------------------------------------------------
    __syncthreads();
    if (tid == 0 && !KGE_DBG(a, 128)) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) t += sh_loss[q];
        atomicAdd(a.loss_parts + (size_t)(blockIdx.x & (LOSS_PARTS - 1)) * LOSS_PART_STRIDE, t);
    }

    // ---- resident rows: one atomic row-add each, or (STAGE) plain 16-byte stores for the owner kernel ----
    if constexpr (STAGE) {
        // The positive's own s and o gradient rows.  Default: staged like the side rows and handed to the owning tiles
        // as bucket entries (roles 2, 3).  On SKEWED graphs a hot entity is the s or o of thousands of positives of one
        // batch, which piles thousands of entries onto one row of one tile and onto one bucket counter (measured 4x
        // slower steps on a zipf graph); the host then sets pos_atomic and these 2 rows per positive take the atomic
        // row-add into the dense gradient buffer instead (+15 us at C2), which the owner folds in when it flushes.
        if (active && !KGE_DBG(a, 64)) {
            if (a.pos_atomic) {
                emit_row(a.g_ent + (int64_t)ps * a.K, gs, a.K, 1.f);
                emit_row(a.g_ent + (int64_t)po * a.K, go, a.K, 1.f);
                if (a.touched && ts == 0) { a.touched[ps] = 1; a.touched[po] = 1; }   // same value from every writer
            } else {
                if (a.hot_map) {   // (wave-uniform branches: ps, po are per slot)
                    const int hs = a.hot_map[ps], ho = a.hot_map[po];
                    if (hs) emit_row(a.hot_buf + ((int64_t)(hs - 1) * HOT_REPL + (blockIdx.x & (HOT_REPL - 1))) * a.K, gs, a.K, 1.f);
                    if (ho) emit_row(a.hot_buf + ((int64_t)(ho - 1) * HOT_REPL + (blockIdx.x & (HOT_REPL - 1))) * a.K, go, a.K, 1.f);
                }
                float* ps_ = a.stage_rows + ((int64_t)i * a.ns + 0) * a.K;
                float* po_ = a.stage_rows + ((int64_t)i * a.ns + 1) * a.K;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (!qok[c]) continue;
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        *reinterpret_cast<float4*>(ps_ + qoff[c] + h * a.k) = make_float4(gs[c][0][h], gs[c][1][h], gs[c][2][h], gs[c][3][h]);
                        *reinterpret_cast<float4*>(po_ + qoff[c] + h * a.k) = make_float4(go[c][0][h], go[c][1][h], go[c][2][h], go[c][3][h]);
                    }
                }
            }
        }
        if (active && a.det) {
            // deterministic mode: the relation-row gradient of this positive is staged as a fifth row; rel_backward_det_kernel
            // adds the rows of one relation in batch order (fp32 atomics would add them in arrival order)
            const float mul = (MODEL == AMDKGE_ROTATE) ? 1.f / a.mc.phase_div : 1.f;   // RotatE: d/dtheta = d/dphi / phase_div, second half 0
            float* pr_ = a.stage_rows + ((int64_t)i * a.ns + 4) * a.K;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (!qok[c]) continue;
#pragma unroll
                for (int h = 0; h < NC; ++h)
                    *reinterpret_cast<float4*>(pr_ + qoff[c] + h * a.k) = make_float4(gp[c][0][h] * mul, gp[c][1][h] * mul, gp[c][2][h] * mul, gp[c][3][h] * mul);
            }
        } else if (active && !KGE_DBG(a, 2)) {
            // relation rows: few and hot -> atomic row-add into the dense relation gradient (swept by kge_opt.hip)
            if constexpr (MODEL == AMDKGE_ROTATE) emit_row(a.g_rel + (int64_t)pp * a.K, gp, a.k, 1.f / a.mc.phase_div);
            else emit_row(a.g_rel + (int64_t)pp * a.K, gp, a.K, 1.f);
        }
    } else if (active && !KGE_DBG(a, 2)) {
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

}  // namespace kge
