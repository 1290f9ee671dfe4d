/*
 * TEST INFRASTRUCTURE (oracle): CPU restatement of the 1-vs-all corruption scoring of
 * AbstractScoringLayer.get_ranks (/root/reference/ampligraph/latent_features/layers/scoring/
 * AbstractScoringLayer.py:156-258) and the `_get_{subject,object}_corruption_scores` of
 * TransE.py:56-114, DistMult.py:51-99, ComplEx.py:65-151, HolE.py:47-89, RotatE.py:107-217
 * in fp32 with a DECLARED accumulation order: one accumulator per (query, entity), units taken in table
 * order (for [re || im] rows: all re units, then all im units), every step rounded to fp32 --
 *     contraction models   acc = fmaf(q[u], e[u], acc)        (query-vector form of ComplEx.py:93-107,138-150)
 *     TransE               acc = acc + |q[u] + e[u]| (subject side), acc + |q[u] - e[u]| (object side)
 *     RotatE               acc = acc + sqrtf(re*re + im*im)
 * The reference's own order is whatever Eigen's reduce_sum does on the machine it runs on (it differs between its
 * CPU and GPU kernels); the numpy oracle (oracle/kge_oracle.py) accumulates in fp64, which is order-free.  This file
 * is the second oracle mode: the order the HIP rank kernels declare (ampligraph_amd/csrc/kge_rank.hip, rank_op), so
 * that filtered ranks can be compared BIT FOR BIT at full size.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Built by oracle/Makefile with -ffp-contract=off: every rounding point is the one written here; fmaf is the
 * correctly rounded fused operation (hardware FMA in the avx2/fma clone, libm's exact software fmaf otherwise).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { MODE_DOT = 0, MODE_L1 = 1, MODE_ROT_O = 2, MODE_ROT_S = 3, MODE_L1_SUB = 4 };   /* same numbering as kge_rank.hip */

#define TILE 64

static inline int32_t quantise(float x) { return (int32_t)(x * 1000.0f); }   /* AbstractScoringLayer.py:201 (truncation) */

/* scores of ONE query against a tile of up to TILE entities whose rows were transposed to T[plane][u][TILE] */
__attribute__((target_clones("avx2,fma", "default")))
static void tile_scores(int mode, const float* q, int qplane, const float* T, int U, int nplanes, float* acc) {
    for (int j = 0; j < TILE; ++j) acc[j] = 0.f;
    const float* T0 = T;
    const float* T1 = T + (size_t)U * TILE;
    if (mode == MODE_DOT) {
        for (int u = 0; u < U; ++u) {
            const float qu = q[u];
            const float* e = T0 + (size_t)u * TILE;
            for (int j = 0; j < TILE; ++j) acc[j] = __builtin_fmaf(qu, e[j], acc[j]);
        }
    } else if (mode == MODE_L1) {
        for (int u = 0; u < U; ++u) {
            const float qu = q[u];
            const float* e = T0 + (size_t)u * TILE;
            for (int j = 0; j < TILE; ++j) acc[j] = acc[j] + fabsf(qu + e[j]);
        }
    } else if (mode == MODE_L1_SUB) {
        for (int u = 0; u < U; ++u) {
            const float qu = q[u];
            const float* e = T0 + (size_t)u * TILE;
            for (int j = 0; j < TILE; ++j) acc[j] = acc[j] + fabsf(qu - e[j]);
        }
    } else if (mode == MODE_ROT_O) {   /* RotatE.py:209-214: q = s o r */
        for (int u = 0; u < U; ++u) {
            const float q0 = q[u], q1 = q[qplane + u];
            const float* e0 = T0 + (size_t)u * TILE;
            const float* e1 = T1 + (size_t)u * TILE;
            for (int j = 0; j < TILE; ++j) {
                const float re = q0 - e0[j], im = q1 - e1[j];
                acc[j] = acc[j] + sqrtf(re * re + im * im);
            }
        }
    } else {   /* MODE_ROT_S, RotatE.py:151-160: q = (cos, sin, o_re, o_im) */
        for (int u = 0; u < U; ++u) {
            const float c = q[u], s = q[qplane + u], orr = q[2 * qplane + u], oi = q[3 * qplane + u];
            const float* e0 = T0 + (size_t)u * TILE;
            const float* e1 = T1 + (size_t)u * TILE;
            for (int j = 0; j < TILE; ++j) {
                const float re = e0[j] * c - e1[j] * s - orr;
                const float im = e0[j] * s + e1[j] * c - oi;
                acc[j] = acc[j] + sqrtf(re * re + im * im);
            }
        }
    }
    (void)nplanes;
}

/*
 * counts[i][0] += #{j : qpos[i] <  q(score(i, j))},  counts[i][1] += #{j : qpos[i] == q(score(i, j))}
 * over entity rows E[ids ? ids[j] : j], j in [0, m).   Q: [n][qw] query vectors (planes qplane apart),
 * E: [*][K] table rows (planes eplane apart), U units per plane.  score = sgn_scale * acc.
 */
void ro_counts(int mode, const float* Q, int64_t qw, int qplane, const float* E, int64_t K, int eplane, int U,
               const int32_t* ids, int64_t m, const int32_t* qpos, int64_t n, float sgn_scale, int32_t* counts) {
    const int nplanes = (mode == MODE_ROT_O || mode == MODE_ROT_S) ? 2 : 1;
    const int64_t ntiles = (m + TILE - 1) / TILE;
#pragma omp parallel
    {
        float* T = (float*)malloc((size_t)nplanes * U * TILE * sizeof(float));
        int32_t* local = (int32_t*)calloc((size_t)2 * n, sizeof(int32_t));
        float acc[TILE];
#pragma omp for schedule(dynamic, 1)
        for (int64_t t = 0; t < ntiles; ++t) {
            const int64_t j0 = t * TILE;
            const int nj = (int)((m - j0) < TILE ? (m - j0) : TILE);
            for (int j = 0; j < TILE; ++j) {
                const int64_t jj = j0 + (j < nj ? j : nj - 1);   /* ragged tile: repeat the last row, masked below */
                const float* row = E + (ids ? (int64_t)ids[jj] : jj) * K;
                for (int p = 0; p < nplanes; ++p)
                    for (int u = 0; u < U; ++u) T[((size_t)p * U + u) * TILE + j] = row[(size_t)p * eplane + u];
            }
            for (int64_t i = 0; i < n; ++i) {
                tile_scores(mode, Q + i * qw, qplane, T, U, nplanes, acc);
                const int32_t qp = qpos[i];
                int gt = 0, eq = 0;
                for (int j = 0; j < nj; ++j) {
                    const int32_t qs = quantise(sgn_scale * acc[j]);
                    gt += qp < qs;
                    eq += qp == qs;
                }
                local[2 * i] += gt;
                local[2 * i + 1] += eq;
            }
        }
#pragma omp critical
        for (int64_t i = 0; i < 2 * n; ++i) counts[i] += local[i];
        free(T);
        free(local);
    }
}

/* quantised scores of explicit (query, entity row) pairs: the true-positive corruptions of the filter step
 * (AbstractScoringLayer.py:260-307), same chain as above */
void ro_pair_qscores(int mode, const float* Q, int64_t qw, int qplane, const float* E, int64_t K, int eplane, int U,
                     const int64_t* pair_q, const int64_t* pair_e, int64_t npairs, float sgn_scale, int32_t* out) {
    const int nplanes = (mode == MODE_ROT_O || mode == MODE_ROT_S) ? 2 : 1;
#pragma omp parallel
    {
        float* T = (float*)malloc((size_t)nplanes * U * TILE * sizeof(float));
        float acc[TILE];
#pragma omp for schedule(static)
        for (int64_t x = 0; x < npairs; ++x) {
            const float* row = E + pair_e[x] * K;
            for (int p = 0; p < nplanes; ++p)
                for (int u = 0; u < U; ++u) {
                    const float v = row[(size_t)p * eplane + u];
                    for (int j = 0; j < TILE; ++j) T[((size_t)p * U + u) * TILE + j] = v;
                }
            tile_scores(mode, Q + pair_q[x] * qw, qplane, T, U, nplanes, acc);
            out[x] = quantise(sgn_scale * acc[0]);
        }
        free(T);
    }
}
