/*
 * amdkge.h -- C ABI of libamdkge.so, the MI355X (gfx950) engine behind AmpliGraph's
 * ScoringBasedEmbeddingModel.fit()/predict()/evaluate() hot path.
 *
 * The reference (Accenture/AmpliGraph, 100 % Python on TensorFlow) has no FFI: its "operator
 * interface" for this path is the set of Keras layers / objects listed next to each entry point
 * below (paths relative to /root/reference/).  Every entry point replaces one of them; the Python
 * host in ampligraph_amd/ binds these symbols with ctypes (see INTEGRATION.md for the binding a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE pointer (HBM), every other pointer
 *     is a host pointer.  Tables are fp32 row-major [rows, K]; K = k for TransE/DistMult and 2k
 *     ([re || im] halves) for ComplEx/HolE/RotatE.  Triples are int32 [n,3] row-major (s,p,o).
 *   - STORED row layout: a model descriptor may declare a padded half width k_pad >= k (amdkge_model.k_pad);
 *     every device table / gradient / optimizer-slot pointer of that model then has rows of
 *     amdkge_row_floats(m) = NC * k_pad floats, [re(k_pad) || im(k_pad)], with the k_pad - k trailing units of
 *     each half ZERO (they stay zero under every optimizer rule and regulariser; RotatE masks them in its
 *     gradient, whose modulus has no epsilon, RotatE.py:102-104).  k_pad % 4 == 0 is what gives every k the
 *     16-byte kernels (owner-computes train pair, pipelined MFMA rank kernel); amdkge_padded_k(k) returns the
 *     smallest such value, amdkge_pack_rows / amdkge_unpack_rows convert to and from the dense [rows, NC*k] form
 *     the reference's get_embeddings / checkpoints use.  k_pad == 0 means dense rows (k_pad = k).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All compute entry
 *     points are asynchronous on that stream; the caller owns synchronisation.
 *   - return value: 0 = ok, negative = error class below; amdkge_last_error() gives a message
 *     (thread local).  Nothing throws across the ABI: the entry points that allocate on the host (the session layers' staging
 *     buffers, threads and registries) catch C++ exceptions and report AMDKGE_ENOMEM (std::bad_alloc) or AMDKGE_EINVAL.
 */
#ifndef AMDKGE_H
#define AMDKGE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDKGE_ABI_VERSION 5

/* error classes */
#define AMDKGE_OK 0
#define AMDKGE_EINVAL (-1)       /* invalid argument */
#define AMDKGE_EHIP (-2)         /* HIP runtime error */
#define AMDKGE_ERCCL (-3)        /* RCCL error (session groups: librccl not found, communicator or collective failed) */
#define AMDKGE_ENOMEM (-4)       /* device (or, in the session layers, host) allocation failed */
#define AMDKGE_EUNSUPPORTED (-5) /* shape outside the compiled kernels' range */

/* scoring_type -- SCORING_LAYER_REGISTRY, latent_features/layers/scoring/AbstractScoringLayer.py:15-18 */
enum { AMDKGE_TRANSE = 0, AMDKGE_DISTMULT = 1, AMDKGE_COMPLEX = 2, AMDKGE_HOLE = 3, AMDKGE_ROTATE = 4 };
/* loss -- LOSS_REGISTRY, latent_features/loss_functions.py:17,229,312,386,468,578 */
enum { AMDKGE_LOSS_PAIRWISE = 0, AMDKGE_LOSS_NLL = 1, AMDKGE_LOSS_ABSOLUTE_MARGIN = 2,
       AMDKGE_LOSS_SELF_ADVERSARIAL = 3, AMDKGE_LOSS_MULTICLASS_NLL = 4 };
/* optimizer -- latent_features/optimizers.py:255-291 (Keras *legacy* update rules) */
/* Keras legacy update rules (tensorflow==2.15, keras/optimizers/legacy/<name>.py; reached from optimizers.py:57-67,279-287, which
 * accepts any legacy optimizer by name).  Slots used: SGD 0; ADAGRAD, MOMENTUM, RMSPROP 1; ADAM, RMSPROP_MOM, ADADELTA, ADAMAX 2. */
enum { AMDKGE_OPT_SGD = 0, AMDKGE_OPT_ADAGRAD = 1, AMDKGE_OPT_ADAM = 2,
       AMDKGE_OPT_MOMENTUM = 3,      /* SGD(momentum=beta1, nesterov = beta2 != 0): a = a*mom - lr*g; x += nesterov ? a*mom - lr*g : a */
       AMDKGE_OPT_RMSPROP = 4,       /* rho = beta1: r += (g^2 - r)(1-rho); x -= lr*g / (sqrt(r) + eps) */
       AMDKGE_OPT_RMSPROP_MOM = 5,   /* rho = beta1, momentum = beta2: r as above; mom = mom*momentum + lr*g / sqrt(r + eps); x -= mom */
       AMDKGE_OPT_ADADELTA = 6,      /* rho = beta1: a = a*rho + g^2(1-rho); u = sqrt(d + eps) / sqrt(a + eps) * g; x -= lr*u; d = d*rho + u^2(1-rho) */
       AMDKGE_OPT_ADAMAX = 7 };      /* m += (g-m)(1-beta1); u = max(beta2*u, |g|); x -= lr/(1-beta1^t) * m / (u + eps) */
/* corrupt side bit mask -- ScoringBasedEmbeddingModel.evaluate(corrupt_side=...) :1516 */
enum { AMDKGE_SIDE_S = 1, AMDKGE_SIDE_O = 2 };
/* ranking_strategy -- AbstractScoringLayer.get_ranks(comparison_type=...) :165 */
enum { AMDKGE_RANK_WORST = 0, AMDKGE_RANK_BEST = 1, AMDKGE_RANK_MIDDLE = 2 };

/* Model geometry: ScoringBasedEmbeddingModel.__init__(eta,k,scoring_type,max_ent_size,max_rel_size)
 * ScoringBasedEmbeddingModel.py:100-108 */
typedef struct amdkge_model {
    int32_t scoring_type;   /* AMDKGE_TRANSE .. AMDKGE_ROTATE */
    int32_t k;              /* user-facing embedding size (rows hold internal_k floats) */
    int64_t n_ents;         /* rows of the entity table */
    int64_t n_rels;         /* rows of the relation table */
    int32_t max_rel_size;   /* RotatE phase normaliser (RotatE.py:95); <=0 means "None" -> 1 */
    int32_t k_pad;          /* stored units per half (>= k), 0 = dense rows; see "STORED row layout" above */
    /* ABI 5, column-sharded tables (amdkge_cols_*): 0 = this descriptor is the whole model; else the descriptor describes a COLUMN
     * SLICE -- k of the k_full units of every row (for the complex models the re and im slices of the same units) -- of a model
     * with k_full units: HolE's 2 / k and RotatE's phase normaliser are those of the whole model, so the slice's sums are partial
     * sums of the whole model's scores. */
    int32_t k_full;
    int32_t reserved_;      /* keeps the struct a multiple of 8 bytes; set to 0 */
} amdkge_model;

/* Loss hyper-parameters: loss_functions.py:76-117 (`hyperparam_dict`), defaults :23-35 */
typedef struct amdkge_loss {
    int32_t kind;           /* AMDKGE_LOSS_* */
    int32_t reduction_mean; /* 0 = "sum" (default), 1 = "mean" over the corruptions */
    float margin;           /* pairwise / absolute_margin / self_adversarial */
    float alpha;            /* self_adversarial sampling temperature */
    /* FocusE numeric-edge weighting (ScoringBasedEmbeddingModel.py:342-368,396-406,468-542): the scores entering the
     * loss become  pos' = f(pos) * (beta + (1-beta)(1-w_i)),  neg'_ji = f(neg_ji) * (beta + (1-beta) w_i)  with
     * w_i = d_focus_w[i] (mean of the numeric columns of positive i).  0 = off. */
    int32_t focus_nonlinearity; /* AMDKGE_FOCUS_* */
    float focus_beta;           /* structural weight beta in [0,1] */
    const float* d_focus_w;     /* device fp32 [B] (this launch's positives); ignored when focus_nonlinearity == 0 */
} amdkge_loss;

enum { AMDKGE_FOCUS_OFF = 0, AMDKGE_FOCUS_LINEAR = 1, AMDKGE_FOCUS_TANH = 2, AMDKGE_FOCUS_SIGMOID = 3, AMDKGE_FOCUS_SOFTPLUS = 4 };

/* Optimizer + regulariser for one table sweep: optimizers.py:136-168, regularizers.py:14-37 */
typedef struct amdkge_opt {
    int32_t kind;           /* AMDKGE_OPT_* */
    int32_t reg_p;          /* LP regulariser power p (>=1); ignored when reg_lambda == 0 */
    float lr;
    float beta1, beta2;     /* Adam / Adamax betas; other kinds: see the AMDKGE_OPT_* enum */
    float epsilon;          /* Adam / Adagrad (Keras legacy default 1e-7) */
    float reg_lambda;       /* 0 = no regulariser */
    int64_t iteration;      /* t = optimizer.iterations + 1 of this step (1-based) */
    /* Touched-rows ("lazy") mode -- an explicit OPT-IN that DEVIATES from the reference, whose optimizer is dense
     * (optimizers.py:136-168 hands Keras a dense gradient: every row's slots decay and every row moves every step).
     * lazy != 0: only the table rows this step TOUCHES are updated (x, slots) and regularised; all other rows keep their
     * bits -- the semantics of TensorFlow-Addons' LazyAdam, generalised to every rule above.  Touched, precisely: in the
     * owner-computes step (amdkge_train_step_tiled) an entity row that is the s or o of a positive, or the replacement row of
     * a corruption whose loss coefficient dL/dscore is non-zero in fp32 (an inactive margin, a clipped score, or a coefficient
     * that underflows fp32 makes no entry); in the row-wise sweep (amdkge_opt_step, which also serves the relation table) a
     * row whose accumulated fp32 gradient row is not entirely zero.  oracle/kge_oracle.py touched_rows restates it.
     * This is what makes a 50 M-row table trainable at HBM speed: the dense sweep moves 7 * 4K bytes per row per step
     * whether or not the row was used.  row_floats = floats per stored table row (amdkge_row_floats); needed by
     * amdkge_opt_step to find row boundaries, ignored when lazy == 0. */
    int32_t lazy;
    int32_t row_floats;
    /* ABI 3: the regulariser forms the reference accepts beyond one shared LP term (EmbeddingLookupLayer.py:131-155 takes a
     * [entity, relation] pair of independent Keras regularisers; tf.keras 'l1_l2' is l1*sum|x| + l2*sum x^2).
     *   reg2_p / reg2_lambda : a SECOND LP term of the swept table, added to the first (0 = none);
     *   rel_reg_p, rel_reg2_p / rel_reg2_lambda : the relation table's terms where one call sweeps both tables
     *       (amdkge_train_step_tiled with relation slots; its rel_reg_lambda argument is the first term's weight);
     *       rel_reg_p == 0 means "the entity table's p" (what ABI 2 callers got). */
    int32_t reg2_p;
    float reg2_lambda;
    int32_t rel_reg_p;
    int32_t rel_reg2_p;
    float rel_reg2_lambda;
} amdkge_opt;

/* ---- library / device helpers (so that a host without torch can drive the engine) ---- */
int amdkge_abi_version(void);
const char* amdkge_last_error(void);
int amdkge_device_count(int* count);
/* Library-internal scratch that is keyed by a caller's pointers -- the 8 KB of per-block loss partials amdkge_train_fwdbwd keeps
 * per (device, d_loss_sum), and the per-(device, workspace) memory of the owner-computes plan guard -- is bounded (1 024 loss
 * accumulators, beyond which amdkge_train_fwdbwd returns AMDKGE_ENOMEM rather than free a buffer another host thread may be about
 * to launch on; 4 096 remembered workspaces, host state only, then forgotten wholesale) and is dropped explicitly here, e.g. after
 * freeing the accumulators and workspaces of a finished job.  Synchronises the devices that hold such scratch.  Call it when none
 * of your own calls is in flight. */
int amdkge_release_scratch(void);
int amdkge_set_device(int device);
int amdkge_dev_alloc(void** d_ptr, uint64_t bytes);
int amdkge_dev_free(void* d_ptr);
int amdkge_h2d(void* d_dst, const void* src, uint64_t bytes, void* stream);
int amdkge_d2h(void* dst, const void* d_src, uint64_t bytes, void* stream);
int amdkge_dev_memset(void* d_ptr, int value, uint64_t bytes, void* stream);
int amdkge_stream_sync(void* stream);

/* internal_k of a model (2k for ComplEx/HolE/RotatE) -- ComplEx.py:37, RotatE.py:57 */
int amdkge_internal_k(int scoring_type, int k);
/* smallest k_pad >= k with 16-byte-aligned halves (multiple of 4) */
int amdkge_padded_k(int k);
/* floats per STORED table row of a model: internal_k(scoring_type, k_pad ? k_pad : k); negative on a bad descriptor */
int amdkge_row_floats(const amdkge_model* m);
/* dense rows [n, internal_k(k)] (what EmbeddingLookupLayer holds, EmbeddingLookupLayer.py:307-342) <-> stored rows
 * [n, amdkge_row_floats(m)].  pack zero-fills the padding.  d_src and d_dst must not overlap. */
int amdkge_pack_rows(const amdkge_model* m, const float* d_dense, int64_t n, float* d_stored, void* stream);
int amdkge_unpack_rows(const amdkge_model* m, const float* d_stored, int64_t n, float* d_dense, void* stream);

/* predict(): EmbeddingLookupLayer.call + <Model>._compute_scores
 * (layers/encoding/EmbeddingLookupLayer.py:307-342; TransE.py:37, DistMult.py:34, ComplEx.py:39,
 *  HolE.py:31, RotatE.py:62).  d_scores[n] fp32. */
int amdkge_score(const amdkge_model* m, const float* d_ent, const float* d_rel,
                 const int32_t* d_triples, int64_t n, float* d_scores, void* stream);

/* CorruptionGenerationLayerTrain.call (layers/corruption_generation/
 * CorruptionGenerationLayerTrain.py:35-94): d_out[(B*eta),3], row j*B+i = j-th corruption of
 * positive i.  Draws: Philox4x32-10, counter = global row j*b_global + row_offset + i, step;
 * key = seed; replacement = sample_base + mulhi32(x1, sample_range). */
int amdkge_sample_corruptions(const int32_t* d_triples, int64_t B, int32_t eta,
                              int64_t sample_base, int64_t sample_range, uint64_t seed, uint64_t step,
                              int64_t row_offset, int64_t b_global, int32_t* d_out, void* stream);

/* One fused forward+backward of train_step (ScoringBasedEmbeddingModel.py:370-429): lookup,
 * negative sampling, scoring, Loss.__call__ (loss_functions.py:185-225) and the gradient of the
 * loss w.r.t. both tables, accumulated (+=) into the dense fp32 buffers d_grad_ent / d_grad_rel
 * (same shape as the tables; the caller zeroes them, amdkge_opt_step re-zeroes them).
 *   d_neg_override : NULL, or int32 [(B*eta),3] corruptions to use instead of sampling
 *   d_loss_sum     : double, += sum_i per-sample loss (no regulariser term)
 *   d_pos_scores   : NULL or fp32 [B];  d_neg_scores : NULL or fp32 [B*eta] (layout j*B+i)
 */
int amdkge_train_fwdbwd(const amdkge_model* m, const amdkge_loss* loss,
                        const float* d_ent, const float* d_rel,
                        const int32_t* d_triples, int64_t B, int32_t eta,
                        int64_t sample_base, int64_t sample_range, uint64_t seed, uint64_t step,
                        int64_t row_offset, int64_t b_global, const int32_t* d_neg_override,
                        float* d_grad_ent, float* d_grad_rel, double* d_loss_sum,
                        float* d_pos_scores, float* d_neg_scores, void* stream);

/* OptimizerWrapper.minimize -> Keras legacy apply_gradients (optimizers.py:166-168) fused with the
 * whole-table LP regulariser (regularizers.py:35-37): one dense sweep over `n_elems` floats of a
 * table.  grad_total = d_grad + lambda*p*|x|^(p-1)*sign(x); updates d_x and the slots in place,
 * zeroes d_grad, and adds lambda*sum|x|^p (pre-update x) to *d_reg_loss (double, may be NULL).
 * With opt->lazy the sweep is row-wise: d_x must start at a row boundary, n_elems must be a multiple of opt->row_floats,
 * rows whose gradient row is entirely zero are skipped (not read beyond the gradient, not written, not regularised).
 *   Adam:    d_slot0 = m, d_slot1 = v;  Adagrad: d_slot0 = accumulator;  SGD: no slots;  the other kinds: slot order as
 *   written in the AMDKGE_OPT_* enum (first state variable = d_slot0). */
int amdkge_opt_step(const amdkge_opt* opt, float* d_x, float* d_grad, float* d_slot0, float* d_slot1,
                    int64_t n_elems, double* d_reg_loss, void* stream);

/* Owner-computes variant of one WHOLE train step: a fused forward + staging kernel, then one workgroup per tile
 * of entity rows accumulates the staged row gradients in LDS and applies the optimizer + regulariser to its rows in
 * place.  Same reference code as amdkge_train_fwdbwd + amdkge_opt_step (ScoringBasedEmbeddingModel.py:370-429,
 * optimizers.py:136-168, regularizers.py:35-37) without global atomics or a dense gradient buffer for the entity
 * table (see `flags` for the skewed-graph variant).  Supported for all five models when the STORED half width (k_pad, or k when k_pad == 0) is a multiple of 4 and <= 2048 (amdkge_train_tiled_workspace_bytes returns 0
 * otherwise and the call returns AMDKGE_EUNSUPPORTED).
 *   apply_update : 1 -> the entity table and its slots are updated in place (single GPU);
 *                0 -> d_grad_ent receives the complete entity gradient (every row written), the relation gradient is ADDED
 *                     to d_grad_rel and nothing is updated (data-parallel: the caller all-reduces both and calls
 *                     amdkge_opt_step)
 *   flags      : AMDKGE_TILED_POS_ATOMIC -> the gradient rows of the positives' own s / o entities bypass the tile buckets
 *                and are added to d_grad_ent with 2 atomic row-adds per positive; the tiles fold d_grad_ent in when they
 *                flush (and reset it when apply_update).  For SKEWED graphs: a hot entity that is the s / o of thousands
 *                of positives of one batch would otherwise serialise one tile (measured 4x on a zipf graph); on uniform
 *                graphs the staged default is ~15 us faster at C2.  Results are identical up to fp32 summation order.
 *   d_grad_ent : dense entity gradient buffer; required unless apply_update && !POS_ATOMIC; zero on entry with POS_ATOMIC
 *   d_grad_rel : dense relation gradient buffer (+=), zero on entry
 *   d_rel_slot0/1, rel_reg_lambda : with apply_update and the slots the optimizer needs given, the relation
 *                table is swept as well (d_grad_rel is consumed and left zero): the call is the complete step.
 *                With NULL relation slots (and an optimizer that needs them) the caller sweeps the relation table.
 *   opt->reg_lambda : regulariser weight of the ENTITY table; d_reg_loss (double, may be NULL) += lambda*sum|x|^p of
 *                every table this call updates
 *   d_work     : scratch of amdkge_train_tiled_workspace_bytes(m, B, eta) bytes.  It must be zero-filled before
 *                its FIRST use; the library leaves its bookkeeping region zeroed after every successful call, so
 *                one buffer (sized for the largest B) serves every later step of the same model. */
#define AMDKGE_TILED_POS_ATOMIC 1
/* AMDKGE_TILED_DETERMINISTIC: bitwise reproducible tables and optimizer state from run to run.  Without it the order of the
 * fp32 additions into a gradient row is the arrival order of its contributions (bucket slots are handed out by a returning
 * atomic, the relation-row gradient uses fp32 atomics), as in the reference's GPU kernels.  With it every tile sorts its entries
 * into a canonical order (their full 128-bit content) in LDS before adding them, and the relation-row gradient is staged per
 * positive and added per relation in batch order by a second kernel: no atomics touch a gradient.  Costs a smaller tile (the LDS
 * is shared with the sort buffer), the sort, and a fifth staged row per positive.  Excludes POS_ATOMIC.  A tile whose entries
 * exceed the sort buffer (an extremely hot tile) is processed unsorted and reported by amdkge_train_tiled_status.
 * (The fp64 loss accumulators still use atomics: they agree to ~1e-15 relative and feed nothing back into the tables.)
 * The mode is also CPU-REPRODUCIBLE: its kernels evaluate the loss terms with declared transcendentals (IEEE add / mul / div
 * only: Cody-Waite exp, atanh-series log) and RotatE's moduli with IEEE sqrtf / division instead of the hardware
 * approximations, so that a numpy restatement of the declared order (oracle/train_ordered.py, test infrastructure) yields the
 * same bits for every model -- tests/test_gpu_learning.py, test_gpu_deterministic.py, test_gpu_fullsize.py. */
#define AMDKGE_TILED_DETERMINISTIC 2
/* AMDKGE_TILED_HOT_ROWS: skewed graphs.  Up to 64 "hot" entity rows, declared beforehand with amdkge_train_tiled_set_hot_rows,
 * receive the gradient rows of the positives whose s / o they are through atomic row-adds spread over 16 replica rows each
 * (summed by the owning tile); every other row keeps the atomic-free staged path.  Replaces POS_ATOMIC where a few entities
 * dominate (zipf: the top entity is the s or o of ~10 % of a batch).  Ignored with DETERMINISTIC / POS_ATOMIC.  Hot rows count
 * as touched in the lazy optimizer mode.  Results are identical up to fp32 summation order. */
#define AMDKGE_TILED_HOT_ROWS 4
/* AMDKGE_TILED_GIVEN_COEFFS (ABI 5): phase C of the COLUMN-SHARDED step (amdkge_cols_* below).  d_pos_scores / d_neg_scores are
 * INPUTS: dL/dscore of the positives [B] and of the corruptions [eta][B] in ONE buffer (d_neg_scores == d_pos_scores + B), as
 * amdkge_cols_loss left them; the forward kernel is replaced by the stage kernel of kge_train_cols.h (gradient rows of the slice
 * from the given coefficients, same staging protocol), the tile pass and the optimizer run unchanged on the slice.  d_loss_sum
 * receives nothing from this call (the data loss is amdkge_cols_loss's), d_reg_loss the slice's regulariser terms.  Excludes
 * DETERMINISTIC / POS_ATOMIC / HOT_ROWS and FocusE; stored slices of up to 256 units per half. */
#define AMDKGE_TILED_GIVEN_COEFFS 8
/* AMDKGE_TILED_DET_WIDE_SORT (ABI 5, with DETERMINISTIC): skewed graphs.  The deterministic tile pass orders a tile's entries inside
 * LDS; by default the buffer holds one bucket + slack (the default mode's tile geometry: the mode's price is 1.38x at C2), and a
 * tile that receives more -- a hub's tile -- sets the sticky status amdkge_train_tiled_status reports.  This flag sizes the buffer at
 * twice a bucket rounded up to a power of two (smaller tiles, possibly a second round of them: slower, as in rounds 2 - 4). */
#define AMDKGE_TILED_DET_WIDE_SORT 16
/* hipGraph capture: a workspace remembers (on the HOST, per device and address) the tile geometry of the last step enqueued on it
 * and re-zeroes its counters when the geometry changes; a captured-and-replayed step bypasses that memory, so a graph may only be
 * replayed on a workspace no step of another geometry (other B / eta / flags) has used since the capture. */
int64_t amdkge_train_tiled_workspace_bytes(const amdkge_model* m, int64_t B, int32_t eta);
/* Long rows (stored half width > 512 units, i.e. rows beyond 2 KB: the C5 row width) take the ROW-DIRECT form of the tile pass by
 * default (kge_tile_direct.h: one wave group per tile, its bucket sorted in LDS, every row folded in registers and updated in
 * one go -- x read once, several tiles resident per CU).  0 keeps them on the LDS-accumulator kernel (A/B measurements, tests);
 * process-wide, both forms compute the same step up to fp32 summation order.  A tile whose entries outgrow the direct form's LDS
 * list rescans the spill in memory (slower, complete): no status is raised for it.  (Any non-zero value means 1: round 5's value 2,
 * a one- / two-wave form for 32 .. 128-quad rows, measured slower than the LDS tiles at BASELINE configs[3] and left the library.) */
int amdkge_set_tile_direct(int on);
/* status != 0 after a DETERMINISTIC step: some tile fell back to unsorted accumulation since the last query (flag is cleared).
 * Synchronises the stream. */
/* Declares the hot rows of a workspace (d_hot_ids: device int32 [n_hot], n_hot <= 64; n_hot = 0 clears them).  The map and the
 * replicas live at offsets of d_work that do not depend on B / eta / flags, so one call serves every later step on that buffer. */
int amdkge_train_tiled_set_hot_rows(const amdkge_model* m, void* d_work, const int32_t* d_hot_ids, int32_t n_hot, void* stream);
int amdkge_train_tiled_status(const amdkge_model* m, int64_t B, int32_t eta, int32_t flags, void* d_work, int32_t* status, void* stream);
int amdkge_train_step_tiled(const amdkge_model* m, const amdkge_loss* loss, const amdkge_opt* opt,
                            float* d_ent, float* d_rel, float* d_ent_slot0, float* d_ent_slot1,
                            float* d_rel_slot0, float* d_rel_slot1, float rel_reg_lambda,
                            const int32_t* d_triples, int64_t B, int32_t eta,
                            int64_t sample_base, int64_t sample_range, uint64_t seed, uint64_t step,
                            int64_t row_offset, int64_t b_global, const int32_t* d_neg_override,
                            float* d_grad_ent, float* d_grad_rel, int32_t apply_update, int32_t flags,
                            double* d_loss_sum, double* d_reg_loss,
                            float* d_pos_scores, float* d_neg_scores, void* d_work, void* stream);

/* COLUMN-SHARDED train step (ABI 5; kge_train_cols.h): every GPU holds k / W units of every row (m->k = k / W,
 * m->k_full = k) and processes ALL positives of the global batch on its slice -- what replaces
 * ScoringBasedEmbeddingModel.train_step (ScoringBasedEmbeddingModel.py:370-429) when the tables are sharded by COLUMNS.  All five
 * scores are sums over units, so one exchange completes them:
 *   1. amdkge_cols_partial_scores: d_scores [B (1 + eta)] <- the slice's partial sums, positives [B] then corruptions [eta][B]
 *      (layout j * B + i, CorruptionGenerationLayerTrain.py:52), un-negated and un-scaled; corruptions drawn in-kernel exactly as
 *      amdkge_train_step_tiled draws them (same arguments), or taken from d_neg_override;
 *   2. the CALLER sums d_scores over the ranks (ncclAllReduce: B (1 + eta) floats, whatever the table size);
 *   3. amdkge_cols_loss: on the complete sums -- negate / scale (TransE, RotatE / HolE), Loss.__call__ (loss_functions.py:185-225):
 *      *d_loss_sum += the batch's data loss, d_scores <- dL/dscore in place (every rank computes the same values);
 *   4. amdkge_train_step_tiled(..., flags | AMDKGE_TILED_GIVEN_COEFFS, ..., d_pos_scores = d_scores, d_neg_scores = d_scores + B):
 *      backward, gradient merge, regulariser and optimizer on the slice -- all element-wise in the columns, nothing remote. */
int amdkge_cols_partial_scores(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples, int64_t B, int32_t eta,
                               int64_t sample_base, int64_t sample_range, uint64_t seed, uint64_t step, int64_t row_offset, int64_t b_global,
                               const int32_t* d_neg_override, float* d_scores, void* stream);
int amdkge_cols_loss(const amdkge_model* m, const amdkge_loss* loss, float* d_scores, int64_t B, int32_t eta, double* d_loss_sum, void* stream);

/* calibrate(): Platt-scaling objective + gradient for one batch of scores -- CalibrationLayer.call(training=1)
 * (layers/calibration/calibrate.py:78-129) and the gradient of ScoringBasedEmbeddingModel.calibrate (:2108-2121).
 *   logit_i = -(w*s_i + b); loss = mean_i weight_i * sigmoid_cross_entropy_with_logits(label_i, logit_i) over the
 *   n_pos + n_neg scores (labels / weights by side).  d_out3 (3 doubles, device) += {loss, dloss/dw, dloss/db}. */
int amdkge_platt_step(const float* d_scores_pos, int64_t n_pos, const float* d_scores_neg, int64_t n_neg,
                      float w, float b, float label_pos, float label_neg, float weight_pos, float weight_neg,
                      double* d_out3, void* stream);

/* evaluate(): AbstractScoringLayer.get_ranks steps (1)+(2) (AbstractScoringLayer.py:156-258,
 * 309-366) for ONE side: quantised positive score vs the quantised score of every corruption.
 *   d_ent_ids : NULL = corruptions are table rows [ent_lo, ent_hi); else int32 [m] row ids
 *               (entities_subset, ScoringBasedEmbeddingModel.py:1349-1354) and ent_lo/hi index it
 *   d_counts  : int32 [n,2], += (#corr with q(pos) < q(corr), #corr with q(pos) == q(corr))
 *   d_work    : scratch of amdkge_rank_workspace_bytes(m, n) bytes */
int64_t amdkge_rank_workspace_bytes(const amdkge_model* m, int64_t n);
/* Testing aid (process-wide): which tile kernel amdkge_rank_counts launches.  0 = automatic (the default: pipelined MFMA
 * kernel for DistMult / ComplEx / HolE -- behind the int8 screening pass when amdkge_rank_counts_screened is given its
 * workspace --, VALU tile kernel for TransE / RotatE), 1 = always the VALU tile kernel, 2 = the first (un-pipelined) MFMA
 * kernel, 3 = the pipelined MFMA kernel with the screening pass off.  All produce the same bits; the parity tests compare them. */
int amdkge_set_rank_kernel(int which);
/* RotatE's per-unit modulus in the rank / filter / corruption-score kernels (process-wide, set before the calls it should apply to).
 *   0 (default) = EXACT: cos / sin of the phase correctly rounded to fp32 (fp64 evaluation, one rounding) and the modulus a
 *       correctly rounded fp32 square root, so the chain  acc = fl(acc + sqrt(fl(fl(re re) + fl(im im))))  over the live units in
 *       table order is a function of the tables alone: filtered ranks are bit-identical to the CPU restatement of that chain
 *       (oracle/csrc/rank_ordered.c).  Needs whole-float4 stored halves (k_pad = amdkge_padded_k(k), what the Python host
 *       always uses); AMDKGE_EUNSUPPORTED otherwise.
 *   1 = FAST: the hardware's 1-ulp v_sqrt_f32 (about 1.3x faster; ranks may differ from the exact mode only where a
 *       quantised comparison at the int32(score * 1000) boundary is decided by the last bit of a modulus). */
int amdkge_set_rank_rotate_fast(int fast);
/* TransE / RotatE: the EXACT EARLY EXIT of the count pass behind amdkge_rank_counts_screened (kge_rank_early.h).  Their scores are
 * -sum of non-negative terms in unit order, so the fp32 partial sums only grow: a pair whose partial sum already quantises below
 * the positive's score is decided for good, with no error bound.  A tile whose undecided pairs have become few hands them to a
 * list, ends, and the list's pairs are recomputed by the full chain -- counts identical to amdkge_rank_counts, bit for bit (rows
 * with non-finite / huge values are exempt; a full list falls back to the plain kernel on the device).  Process-wide testing /
 * tuning aid: on = 0 switches it off (amdkge_rank_screen_workspace_bytes then returns 0 for these models); check_l1 / check_rot =
 * stages of 16 units between two checks (defaults 4 / 1: EarlyCfg in kge_rank_early.h), cost = how many tile-kernel pair chains a
 * re-checked pair is priced at when deciding whether a tile ends (default 16); arguments <= 0 keep the current value.  probe: 1 (default) = a sample of 4 096
 * pairs decides whether the call is worth the early-exit kernel (are at least half of them decided at half their units?) or runs
 * the plain kernel (tables whose positives do not stand out: an untrained model) -- the 8-byte answer is read back on the host,
 * so such a call synchronises its stream once, right after its prep kernels; 0 = always the early-exit kernel (tests); < 0 keeps
 * the current value. */
int amdkge_set_rank_early(int on, int check_l1, int check_rot, int cost, int probe);
int amdkge_rank_counts(const amdkge_model* m, const float* d_ent, const float* d_rel,
                       const int32_t* d_triples, int64_t n, int32_t side,
                       const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                       int32_t* d_counts, void* d_work, void* stream);
/* The same counts, bit for bit, through the INT8 SCREENING PASS for the contraction models (DistMult / ComplEx / HolE;
 * kge_rank_screen.h): rows become 24-bit fixed point (three int8 limbs, a power-of-two scale per row), v_mfma_i32_32x32x32_i8
 * forms the integer dot products exactly, a rigorous per-pair error bound (the fp32 chain's own rounding + the fixed-point
 * rounding + the dropped limb products) decides every comparison whose outcome it cannot change, and the rest -- a fraction of a
 * per cent -- are recomputed with the exact fp32 chain.  d_screen: amdkge_rank_screen_workspace_bytes(m, n, ent_hi - ent_lo)
 * bytes (0 = nothing to gain for this model / size); NULL / too small or tiny problems: the call is amdkge_rank_counts.  With the
 * same workspace TransE / RotatE take their exact early exit (amdkge_set_rank_early above).
 * After the stream is synchronised the first int32 of d_screen (aligned up to 256 bytes) holds the number of rechecked pairs,
 * the second is non-zero if the recheck list overflowed and the call fell back to the exact kernel (distance models: the third
 * counts the tiles that ended early). */
int64_t amdkge_rank_screen_workspace_bytes(const amdkge_model* m, int64_t n, int64_t n_cand);
int amdkge_rank_counts_screened(const amdkge_model* m, const float* d_ent, const float* d_rel,
                                const int32_t* d_triples, int64_t n, int32_t side,
                                const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi,
                                int32_t* d_counts, void* d_work, void* d_screen, int64_t screen_bytes, void* stream);

/* get_ranks step (3): filter correction (AbstractScoringLayer.py:260-307,368-417).  For triple i
 * the true-positive ids are d_flt_ids[d_flt_lo[i] .. d_flt_hi[i]) (a CSR when lo=off[i],
 * hi=off[i+1]).  Ids are table row ids; they are kept if ent_lo <= id < ent_hi (partition rule
 * :280-288) and, when d_subset_pos != NULL (int32 [n_ents], -1 = not in entities_subset,
 * :266-275), if d_subset_pos[id] >= 0.  d_sub[n] int32 += #{f : q(pos) <= q(corr_f)}. */
/* The filter index itself, built on the device (kge_filter.hip): the CSR that amdkge_filter_ranges searches, from the
 * concatenated, id-mapped filter datasets -- replaces the reference's per-batch pandas group-by + `sum(lists, [])`
 * (datasets/graph_data_loader.py:287-350,382-439).  side S: groups keyed (p * n_ents + o) with the SET of subjects seen with
 * them; side O: groups keyed (s * n_rels + p) with the SET of objects.
 *   d_triples : int32 [m, 3] (duplicates allowed: the datasets may overlap)
 *   d_keys    : int64 [m]      out: the sorted distinct group keys (first n_groups valid)
 *   d_start   : int64 [m + 1]  out: CSR offsets into d_ids (first n_groups + 1 valid)
 *   d_ids     : int32 [m]      out: the values, ascending within a group (first n_unique valid)
 *   d_counts  : int64 [2]      out: n_groups, n_unique (read them back after synchronising the stream)
 *   d_work    : amdkge_filter_build_workspace_bytes(m, n_ents, n_rels) bytes.  n_rels * n_ents^2 must be < 2^63. */
int64_t amdkge_filter_build_workspace_bytes(int64_t m, int64_t n_ents, int64_t n_rels);
int amdkge_filter_build(const int32_t* d_triples, int64_t m, int32_t side, int64_t n_ents, int64_t n_rels,
                        int64_t* d_keys, int64_t* d_start, int32_t* d_ids, int64_t* d_counts, void* d_work, void* stream);
int amdkge_rank_filter(const amdkge_model* m, const float* d_ent, const float* d_rel,
                       const int32_t* d_triples, int64_t n, int32_t side,
                       const int64_t* d_flt_lo, const int64_t* d_flt_hi, const int32_t* d_flt_ids,
                       const int32_t* d_subset_pos, int64_t ent_lo, int64_t ent_hi,
                       int32_t* d_sub, void* d_work, void* stream);

/* Filter lookup for a batch of test triples: what the reference's data handler does per batch with pandas
 * (graph_data_loader.py:287-350 get_participating_entities, :382-439) -- here a binary search per triple in a
 * sorted key array built once on the host (datasets/filters.py: key = p * n_ents + o for the subject side,
 * s * n_rels + p for the object side; d_start[n_keys + 1] = CSR offsets of each key's ids).
 * d_lo[i], d_hi[i] = the range of triple i's true-positive ids (0, 0 when the key is absent): the d_flt_lo / d_flt_hi
 * arguments of amdkge_rank_filter. */
int amdkge_filter_ranges(const int64_t* d_keys, const int64_t* d_start, int64_t n_keys,
                         const int32_t* d_triples, int64_t n, int32_t side, int64_t n_ents, int64_t n_rels,
                         int64_t* d_lo, int64_t* d_hi, void* stream);

/* Tie strategy + "+1" (AbstractScoringLayer.py:217-258, ScoringBasedEmbeddingModel.py:1684):
 * d_ranks[i] = strategy(gt,eq) - sub + 1;  d_sub may be NULL (unfiltered). */
int amdkge_rank_compose(const int32_t* d_counts, const int32_t* d_sub, int64_t n, int32_t strategy,
                        int32_t* d_ranks, int64_t rank_stride, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Discovery helpers on the device: query_topn (discovery/discovery.py:985-1168) and find_nearest_neighbours (:1171-1244).
 *
 * amdkge_corruption_scores: the un-quantised scores of EVERY corruption of one side, d_scores[i * ld + j] = score of triple i
 *   with its subject (side S) / object (side O) replaced by candidate j (table row ent_lo + j, or d_ent_ids[ent_lo + j]) --
 *   <Model>._get_subject/object_corruption_scores (TransE.py:56-114, DistMult.py:51-99, ComplEx.py:65-151, HolE.py:47-89,
 *   RotatE.py:107-217) through the same prep + tile kernels as amdkge_rank_counts (same accumulation chain).  Callers pass
 *   a bounded chunk of queries and select with amdkge_topk_rows: the (n, m) matrix never exists beyond a chunk.
 *   d_work: amdkge_rank_workspace_bytes(m, n) bytes.
 * amdkge_row_dots   : d_out[i * ld + j] = <d_q[i], row j> (rows of row_floats floats) -- the GEMM form of euclidean / cosine
 *   nearest neighbours (|q - e|^2 = |q|^2 + |e|^2 - 2 <q, e>)
 * amdkge_row_sqnorms: d_out[j] = mul * |row j|^2, or 1 / |row j| when rsqrt != 0, for rows lo + j (or d_ids[lo + j])
 * amdkge_topk_rows  : per row i of d_vals [n, m] (leading dimension ld) the k (<= 1024) largest (largest != 0) or smallest
 *   entries of v[j] = d_vals[i*ld + j] * d_col_scale[j] + d_col_bias[j] (either array may be NULL): d_out_idx [n, k] column
 *   indices (-1 when m < k) -- or, with d_payload int32 [n, ld], the payload entries of those columns (merging per-shard
 *   candidate lists: payload = their global ids) --, d_out_val [n, k] the values v, best first; equal values in order of
 *   increasing column. */
int amdkge_corruption_scores(const amdkge_model* m, const float* d_ent, const float* d_rel, const int32_t* d_triples, int64_t n,
                             int32_t side, const int32_t* d_ent_ids, int64_t ent_lo, int64_t ent_hi, float* d_scores, int64_t ld,
                             void* d_work, void* stream);
int amdkge_row_dots(const float* d_q, int64_t n, const float* d_table, int32_t row_floats, const int32_t* d_ent_ids,
                    int64_t ent_lo, int64_t ent_hi, float* d_out, int64_t ld, void* stream);
int amdkge_row_sqnorms(const float* d_table, int32_t row_floats, const int32_t* d_ids, int64_t lo, int64_t n, float mul, int32_t rsqrt,
                       float* d_out, void* stream);
int amdkge_topk_rows(const float* d_vals, int64_t n, int64_t m, int64_t ld, const float* d_col_scale, const float* d_col_bias,
                     const int32_t* d_payload, int32_t k, int32_t largest, int32_t* d_out_idx, float* d_out_val, void* stream);
/* exact distances of explicit pairs: d_out[i*k + j] = |d_q[i] - row(d_pos[i*k + j])| (cosine != 0: 1 - cos), rows addressed like
 * amdkge_row_sqnorms (lo + pos, or d_ids[lo + pos]); pos < 0 -> +inf.  Re-measures the neighbours the GEMM-form selection kept. */
int amdkge_pair_distances(const float* d_q, int64_t n, const float* d_table, int32_t row_floats, const int32_t* d_ids, int64_t lo,
                          const int32_t* d_pos, int32_t k, int32_t cosine, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU data path (one process per GPU; the host issues the RCCL collectives between these calls -- see
 * ampligraph_amd/sharded.py and trainer.py).  The reference has no multi-device path; what these replace is its
 * partitioned training loop, ScoringBasedEmbeddingModel.py:227,259-261 (corruptions from the partition's entities) and
 * :1431-1452 (evaluation bucket by bucket), with owner(e) = e / ceil(N / G) as in datasets/graph_partitioner.py:339-344.
 *
 * amdkge_shard_route: rank `rank` of `world` owns rows [rank * rows_per, min(N, (rank + 1) * rows_per)), rows_per =
 *   ceil(n_ents / world).  Every s / o id of d_triples [b,3] (and d_negs [nneg,3], may be NULL / 0) is rewritten into the
 *   rank's LOCAL index space: an owned id e -> e - lo; a remote id -> n_local + peer * cap + position, the scratch row
 *   behind the shard that will hold its fetched copy (equal ids get equal rows).  d_send_ids [world * cap] receives, per
 *   peer, the row indices AT THAT PEER of the distinct ids requested from it (-1 = unused slot): exchanged with one
 *   equal-split all_to_all, they are the gather list of amdkge_gather_rows there.  d_counts [world + 1]: requests per peer,
 *   d_counts[world] is set to 1 if a peer's list overflowed `cap` (results are then invalid: grow cap); it is sticky -- the
 *   library never clears it, so the host can check it once per epoch.  No host synchronisation.
 *   d_work: amdkge_shard_route_workspace_bytes(b, nneg) bytes.
 * amdkge_gather_rows      : d_out[j] = d_table[d_idx[j]] (rows of row_floats floats; idx < 0 -> zero row)
 * amdkge_scatter_add_rows : d_table[d_idx[j]] += d_src[j] (idx < 0 skipped; indices may repeat) -- gradient rows of fetched
 *                           copies returning to their owner
 * amdkge_opt_step_merged  : amdkge_opt_step whose gradient is the sum of n_parts slices d_grad_parts + q * part_stride
 *                           (the partial sums a reduce-scatter by all_to_all delivered), summed in part order; dense mode only
 * amdkge_synth_triples    : d_out[i] = triple number first_row + i of the counter-based synthetic stream (Philox4x32-10,
 *                           key = seed): s, o uniform over n_ents, p uniform over n_rels (SURVEY.md 8d "synth-50M") */
int64_t amdkge_shard_route_workspace_bytes(int64_t b, int64_t nneg);
int amdkge_shard_route(int64_t n_ents, int32_t world, int32_t rank, const int32_t* d_triples, int64_t b,
                       const int32_t* d_negs, int64_t nneg, int32_t cap, int32_t* d_out_triples, int32_t* d_out_negs,
                       int32_t* d_send_ids, int32_t* d_counts, void* d_work, void* stream);
int amdkge_gather_rows(const float* d_table, int32_t row_floats, const int32_t* d_idx, int64_t n, float* d_out, void* stream);
int amdkge_scatter_add_rows(float* d_table, int32_t row_floats, const int32_t* d_idx, int64_t n, const float* d_src, void* stream);
int amdkge_opt_step_merged(const amdkge_opt* opt, float* d_x, const float* d_grad_parts, int32_t n_parts, int64_t part_stride,
                           float* d_slot0, float* d_slot1, int64_t n_elems, double* d_reg_loss, void* stream);
int amdkge_synth_triples(uint64_t seed, int64_t first_row, int64_t n, int64_t n_ents, int64_t n_rels, int32_t* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Session layer: HOST pointers in, HOST pointers out.  One opaque handle owns the HBM-resident state of a model (both
 * tables, their gradient buffers, the optimizer state, every scratch buffer, one HIP stream) and drives the entry points
 * above, so a host without torch -- the reference's own numpy-level Python, through ctypes -- can train, score and rank:
 *   amdkge_session_create      <-> ScoringBasedEmbeddingModel.__init__ + compile + build (ScoringBasedEmbeddingModel.py:100-187,1145-1216)
 *   amdkge_session_set/get_rows<-> EmbeddingLookupLayer weights, get_embeddings (:2214-2277), save/load_weights (:1046-1143)
 *   amdkge_session_train_step  <-> one train_step of fit() on one batch (:370-429)
 *   amdkge_session_score       <-> predict (:1694-1734)
 *   amdkge_session_rank        <-> evaluate: make_test_function + get_ranks (:1387-1465,1684; AbstractScoringLayer.py:156-422)
 * Calls are synchronous (results are in the host buffers on return); a handle is used by one host thread at a time; the
 * caller owns every host pointer, the library owns all device memory.  Errors: return codes + amdkge_last_error(). */
typedef struct amdkge_session amdkge_session;

typedef struct amdkge_session_config {
    amdkge_model model;
    amdkge_loss loss;          /* d_focus_w is ignored here (FocusE weights are passed per step) */
    amdkge_opt opt;            /* iteration is ignored (the session counts); reg_lambda = lambda of the ENTITY table */
    float rel_reg_lambda;      /* lambda of the relation table (same p as the entity table) */
    int32_t eta;               /* corruptions per positive */
    uint64_t seed;             /* negatives: Philox key; step counter = number of train steps done so far */
    int32_t device;            /* HIP device ordinal */
    int32_t flags;             /* AMDKGE_TILED_* of the train steps: 0, POS_ATOMIC (skewed graphs, all positives' rows through
                                * atomics) or DETERMINISTIC; HOT_ROWS is switched on by amdkge_session_set_hot_rows */
} amdkge_session_config;

enum { AMDKGE_TABLE_ENT = 0, AMDKGE_TABLE_REL = 1, AMDKGE_TABLE_ENT_SLOT0 = 2, AMDKGE_TABLE_ENT_SLOT1 = 3,
       AMDKGE_TABLE_REL_SLOT0 = 4, AMDKGE_TABLE_REL_SLOT1 = 5 };
enum { AMDKGE_CORRUPT_S = 1, AMDKGE_CORRUPT_O = 2, AMDKGE_CORRUPT_S_O = 3 /* "s,o": two columns */,
       AMDKGE_CORRUPT_S_PLUS_O = 4 /* "s+o": one column, ScoringBasedEmbeddingModel.py:1459-1463 */ };

/* Tables start at zero (optimizer state at the Keras initial values); fill them with amdkge_session_set_rows. */
int amdkge_session_create(const amdkge_session_config* cfg, amdkge_session** out);
void amdkge_session_destroy(amdkge_session* s);
/* rows [row0, row0 + nrows) of a table <- host fp32 [nrows, internal_k] */
int amdkge_session_set_rows(amdkge_session* s, int32_t table, int64_t row0, int64_t nrows, const float* host);
/* host fp32 [nrows, internal_k] <- rows ids[0..nrows) (ids != NULL) or rows [row0, row0 + nrows) */
int amdkge_session_get_rows(amdkge_session* s, int32_t table, const int32_t* ids, int64_t row0, int64_t nrows, float* host);
/* One training step on B positives (host int32 [B,3]); focus_w: NULL or host fp32 [B] (FocusE, needs
 * cfg.loss.focus_nonlinearity); *loss_out (may be NULL) = data loss + regulariser terms of this batch. */
int amdkge_session_train_step(amdkge_session* s, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out);
/* Skewed graphs: declare up to 64 hot entity rows (AMDKGE_TILED_HOT_ROWS); n = 0 switches the feature off again. */
int amdkge_session_set_hot_rows(amdkge_session* s, const int32_t* ids, int32_t n);
int amdkge_session_score(amdkge_session* s, const int32_t* triples, int64_t n, float* scores_out);
/* Ranks of n test triples.  Filters: CSR over the test triples, fs_off / fo_off host int64 [n + 1] into fs_ids / fo_ids
 * (true subjects / objects of each triple; NULL offsets = unfiltered side).  ent_subset: NULL or n_subset entity ids
 * (entities_subset).  ranks_out: host int32 [n, 2] for AMDKGE_CORRUPT_S_O, else [n]. */
int amdkge_session_rank(amdkge_session* s, const int32_t* triples, int64_t n,
                        const int64_t* fs_off, const int32_t* fs_ids, const int64_t* fo_off, const int32_t* fo_ids,
                        const int32_t* ent_subset, int64_t n_subset, int32_t corrupt_side, int32_t strategy,
                        int32_t* ranks_out);
/* Of the last amdkge_session_rank call (its last side): *ran = the count pass was handed a screening / early-exit workspace
 * (amdkge_rank_counts_screened -- the session keeps it), *rechecked_pairs = pairs the exact chain re-checked and *fell_back = the
 * list overflowed into the exact kernel -- of the int8 screening pass for DistMult / ComplEx / HolE, of the exact early exit for
 * TransE / RotatE; both are 0 when the library took the plain kernel inside that workspace (problems too small for either pass,
 * distance-model tables on which the probe expects no early decisions).  Any pointer may be NULL. */
int amdkge_session_screen_stats(const amdkge_session* s, int32_t* ran, int64_t* rechecked_pairs, int32_t* fell_back);

/* ------------------------------------------------------------------------------------------------------------------
 * Session GROUP: the session layer on several GPUs of one node from ONE process (kge_session_group.hip) -- what a host
 * without torch binds to train data-parallel.  One session (replicated tables, optimizer state, stream) per device; the
 * library owns the RCCL communicators (librccl is bound at run time; errors: AMDKGE_ERCCL) and issues the collectives.
 *   devices : n_gpus HIP device ordinals, all distinct (gradient sum = ncclAllReduce over xGMI inside one
 *             ncclGroupStart / End), or all the same (the replicas share one GPU and sum with a kernel: development /
 *             tests on a one-GPU box); NULL = 0 .. n_gpus - 1.  n_gpus = 1 is a plain session.
 *   amdkge_session_group_train_step : ONE global batch of B positives (ScoringBasedEmbeddingModel.train_step,
 *             ScoringBasedEmbeddingModel.py:370-429): replica d computes the gradients of rows [B d / n, B (d + 1) / n) with
 *             the negatives one GPU would draw for the whole batch (keyed by the global corruption row), the dense gradients of
 *             both tables are summed over the replicas, every replica applies the same dense update: replicas stay
 *             bit-identical, and n replicas compute the step of one GPU up to fp32 summation order.  The reference has no
 *             multi-device path at all.  With AMDKGE_TILED_DETERMINISTIC a step whose sorted accumulation fell back in some
 *             tile returns AMDKGE_EUNSUPPORTED after carrying the step out (as amdkge_session_train_step does).  Hot-row
 *             replicas (amdkge_session_set_hot_rows on a replica) apply to single-session steps only: group steps ignore them.
 *             If a replica's share fails, the gradients the earlier replicas already formed are cleared before the error returns.
 *   amdkge_session_group_set_rows   : rows of a table on every replica;   amdkge_session_group_replica : replica i, for
 *             amdkge_session_get_rows / _score / _rank (every replica holds the whole model). */
typedef struct amdkge_session_group amdkge_session_group;
int amdkge_session_group_create(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, amdkge_session_group** out);
/* flags: AMDKGE_GROUP_FORCE_RCCL -- a group of ONE replica takes the multi-replica path too (librccl bound, ncclCommInitAll over
 * the one device, gradient-only kernels, grouped ncclAllReduce of both gradient tables, dense sweeps): every RCCL call of the
 * group step is exercised on a one-GPU box.  amdkge_session_group_info: whether the group sums through RCCL, and ncclGetVersion. */
enum { AMDKGE_GROUP_FORCE_RCCL = 1, AMDKGE_GROUP_ROWS = 2 /* set by amdkge_session_group_create_rows */, AMDKGE_GROUP_GLOBAL_NEGATIVES = 4,
       AMDKGE_GROUP_COLS = 8 /* set by amdkge_session_group_create_cols */,
       /* one host thread per replica in amdkge_session_group_rank even when the replicas share a device (replicas on distinct devices
        * always get one): the path a multi-GPU node takes, testable on one GPU */
       AMDKGE_GROUP_FORCE_THREADS = 16 };
int amdkge_session_group_create_ex(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                   amdkge_session_group** out);
int amdkge_session_group_info(const amdkge_session_group* g, int32_t* uses_rccl, int32_t* rccl_version);
/* ROW-SHARDED group (BASELINE configs[3] / [4]: tables that do not fit, or should not be replicated on, every GPU).  cfg->model.n_ents
 * is the GLOBAL entity count N; replica d owns rows [d * rows_per, min(N, (d + 1) * rows_per)), rows_per = ceil(N / n_gpus) -- the
 * reference's bucket rule owner(e) = e // ceil(N / G) (datasets/graph_partitioner.py:339-344) -- with their optimizer state, the
 * relation table is replicated.  amdkge_session_group_train_step then is the partitioned step of
 * ScoringBasedEmbeddingModel.py:227,259-261 with the partitions on different GPUs: ids routed to their owners on the device
 * (amdkge_shard_route), request ids / rows / returning gradient rows exchanged as equal-split grouped ncclSend / ncclRecv (device
 * copies between replicas that share one device), the fused kernels in their gradient-only form on the local index space, the
 * relation gradient all-reduced, every replica sweeping its own rows.  max_batch: the largest B a step will be given (sizes the
 * request lists at their worst case).  flags: AMDKGE_GROUP_GLOBAL_NEGATIVES = corruptions drawn over all N ids exactly as on one
 * GPU (same Philox rows: n replicas compute one GPU's step up to fp32 summation order; fabric-bound) instead of the default,
 * shard-local negatives (replacement ids from the replica's own rows: what the reference's partitioned training does);
 * AMDKGE_GROUP_FORCE_RCCL as above.  In such a group
 *   amdkge_session_group_set_rows / _get_rows speak GLOBAL row numbers (entity tables and their slots are scattered to / gathered
 *     from the owners; _get_rows also serves replicated groups, from replica 0);
 *   amdkge_session_group_route_overflow reports (and clears) whether a request list overflowed since the last query (cannot happen
 *     with B <= max_batch; kept as the device-side guard it is in the torch host);
 *   amdkge_session_group_replica hands out a replica's LOCAL view (its shard + scratch rows): score / rank through it see only
 *     that shard -- evaluate through amdkge_session_group_rank below.
 * AMDKGE_TILED_DETERMINISTIC is not offered for row-sharded groups (AMDKGE_EUNSUPPORTED). */
int amdkge_session_group_create_rows(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                     int64_t max_batch, amdkge_session_group** out);
/* COLUMN-SHARDED group (ABI 5; kge_train_cols.h): for tables that FIT every GPU (BASELINE configs[1] - [3]) and
 * whose merge bounds data-parallel scaling.  cfg->model is the WHOLE model; replica d holds units [d k / W, (d + 1) k / W) of every
 * entity and relation row (k % n_gpus == 0; the re and im slices of the same units) with the optimizer state of those columns, and
 * every replica processes the WHOLE batch of a step on its slice.  amdkge_session_group_train_step: partial score sums
 * (amdkge_cols_partial_scores), ONE all-reduce of B (1 + eta) floats (ncclAllReduce over xGMI; same-device replicas: a kernel), then
 * loss / backward / gradient merge / regulariser / optimizer locally (amdkge_cols_loss, amdkge_train_step_tiled with
 * AMDKGE_TILED_GIVEN_COEFFS): W replicas compute one GPU's step (the same Philox corruptions) up to fp32 summation order.
 * amdkge_session_group_set_rows / _get_rows take and return WHOLE rows (columns scattered to / gathered from the replicas);
 * evaluation: gather the rows into one session of its own -- amdkge_session_group_rank returns AMDKGE_EUNSUPPORTED for such a group.  Not offered:
 * FocusE, DETERMINISTIC, POS_ATOMIC, hot rows; slices of more than 256 stored units per half.
 * A step that FAILS: before any slice has applied its update the group is left as it was (gradients cleared, workspaces dropped) and the
 * step may be retried; once some slices have been updated the columns of the one model have parted and the group refuses further steps
 * (AMDKGE_EINVAL) until the tables are written again with amdkge_session_group_set_rows -- the whole entity table clears the state --
 * or the group is destroyed and rebuilt. */
int amdkge_session_group_create_cols(const amdkge_session_config* cfg, const int32_t* devices, int32_t n_gpus, int32_t flags,
                                     amdkge_session_group** out);
int amdkge_session_group_get_rows(amdkge_session_group* g, int32_t table, const int32_t* ids, int64_t row0, int64_t nrows, float* host);
int amdkge_session_group_route_overflow(amdkge_session_group* g, int32_t* overflowed);
void amdkge_session_group_destroy(amdkge_session_group* g);
int32_t amdkge_session_group_size(const amdkge_session_group* g);
int amdkge_session_group_replica(amdkge_session_group* g, int32_t i, amdkge_session** out);
int amdkge_session_group_set_rows(amdkge_session_group* g, int32_t table, int64_t row0, int64_t nrows, const float* host);
int amdkge_session_group_train_step(amdkge_session_group* g, const int32_t* triples, int64_t B, const float* focus_w, double* loss_out);
/* evaluate() through a group (ABI 5): the ranks of n test triples, arguments and result exactly as amdkge_session_rank, ids in GLOBAL
 * numbering.  ROW-SHARDED group: the reference's loop over entity partitions (ScoringBasedEmbeddingModel.py:1431-1452) with the
 * partitions on different GPUs -- every replica counts all queries against ITS rows (filter ids restricted to its range,
 * AbstractScoringLayer.py:280-288; an entities_subset contributes its locally owned candidates), the s / o rows the queries need
 * are gathered at their owners and summed bit-wise over the replicas into the scratch rows behind every shard (ncclAllReduce of
 * the int32 patterns; same-device replicas: a kernel), counts and filter subtractions are summed the same way, replica 0 applies the
 * tie strategy and the +1 (:1459-1463,1684).  Queries are processed in chunks of (scratch rows per replica) / 2, i.e. sized by the
 * max_batch the group was created with.  REPLICATED group: the queries are split over the replicas (one host thread per device).
 * Either way the ranks are those of amdkge_session_rank on one session holding the whole table, bit for bit. */
int amdkge_session_group_rank(amdkge_session_group* g, const int32_t* triples, int64_t n,
                              const int64_t* fs_off, const int32_t* fs_ids, const int64_t* fo_off, const int32_t* fo_ids,
                              const int32_t* ent_subset, int64_t n_subset, int32_t corrupt_side, int32_t strategy,
                              int32_t* ranks_out);

#ifdef __cplusplus
}
#endif
#endif /* AMDKGE_H */
