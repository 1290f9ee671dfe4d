"""numpy restatement of the AmpliGraph hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference code it follows; paths are relative to
/root/reference/.  Arithmetic convention: elementwise ops are done in fp32 at
the same points the reference's TF graph rounds to fp32; reductions over the
embedding dimension are accumulated in fp64 and rounded once to fp32, which
makes the oracle independent of any summation order (the reference's Eigen
tree order is unknowable, the HIP kernels use wave-shuffle / MFMA k-ordered
chains).  Parity of floating-point outputs is therefore "within 1e-5 relative"
(north_star); parity of integer outputs (ranks) is bit-exact on inputs whose
arithmetic is exact in fp32 (dyadic-rational tables) and "equal except on
fragile comparisons" otherwise (see `fragile_rank_mask`).
"""
import math

import numpy as np

from .philox import sample_corruption_draws

F32 = np.float32
MODELS = ("TransE", "DistMult", "ComplEx", "HolE", "RotatE")
COMPARISON_PRECISION = F32(1e3)  # ampligraph/latent_features/layers/scoring/AbstractScoringLayer.py:11


def internal_k(model, k):
    """ComplEx/HolE/RotatE store [re || im] halves: ComplEx.py:37, RotatE.py:57."""
    return 2 * k if model in ("ComplEx", "HolE", "RotatE") else k


def rotate_phase_divisor(k, max_rel_size):
    """RotatE.py:95-98: theta / (embedding_range / pi), embedding_range=(6/(2k*R))**0.5.

    The quotient `embedding_range / pi` is a Python double that TF converts to an
    fp32 constant before the division, hence the F32 cast.
    """
    if max_rel_size is None:
        max_rel_size = 1  # RotatE.py:87-94
    embedding_range = (6 / (2 * k * max_rel_size)) ** 0.5
    return F32(embedding_range / math.pi)


def _sum_k(x):
    return x.astype(np.float64).sum(axis=-1).astype(F32)


def _split(x):
    h = x.shape[-1] // 2
    return x[..., :h], x[..., h:]


# ----------------------------------------------------------------------------
# a1: lookup (layers/encoding/EmbeddingLookupLayer.py:307-342)
# ----------------------------------------------------------------------------
def lookup(ent, rel, triples):
    triples = np.asarray(triples)
    return ent[triples[:, 0]], rel[triples[:, 1]], ent[triples[:, 2]]


# ----------------------------------------------------------------------------
# a4-a8: pointwise scores
# ----------------------------------------------------------------------------
def _rotate_rel(p, k, max_rel_size):
    theta = p[..., :k]  # RotatE.py:79 second half unused
    phi = (theta / rotate_phase_divisor(k, max_rel_size)).astype(F32)
    return np.cos(phi.astype(np.float64)).astype(F32), np.sin(phi.astype(np.float64)).astype(F32)


def compute_scores(model, s, p, o, k=None, max_rel_size=None):
    """`_compute_scores` of TransE.py:37-54, DistMult.py:34-49, ComplEx.py:39-63,
    HolE.py:31-45, RotatE.py:62-105.  s,p,o: (n,K) fp32 -> (n,) fp32."""
    s, p, o = (np.asarray(a, dtype=F32) for a in (s, p, o))
    if model == "TransE":
        return -_sum_k(np.abs(s + p - o))
    if model == "DistMult":
        return _sum_k(s * p * o)
    if model in ("ComplEx", "HolE"):
        sr, si = _split(s)
        pr, pi = _split(p)
        orr, oi = _split(o)
        sc = _sum_k(sr * (pr * orr + pi * oi) + si * (pr * oi - pi * orr))
        if model == "HolE":
            kk = s.shape[-1] // 2
            sc = (F32(2 / kk) * sc).astype(F32)  # HolE.py:45 (2 / (internal_k / 2))
        return sc
    if model == "RotatE":
        kk = s.shape[-1] // 2
        sr, si = _split(s)
        orr, oi = _split(o)
        pr, pi = _rotate_rel(p, kk, max_rel_size)
        re = sr * pr - si * pi - orr
        im = sr * pi + si * pr - oi
        return -_sum_k(np.sqrt(re * re + im * im))
    raise ValueError(model)


# ----------------------------------------------------------------------------
# a9: 1-vs-all corruption scores (n, m)
# ----------------------------------------------------------------------------
def corruption_scores(model, side, s, p, o, ent, max_rel_size=None):
    """`_get_subject_corruption_scores` / `_get_object_corruption_scores`
    (TransE.py:56-114, DistMult.py:51-99, ComplEx.py:65-151, HolE.py:47-89,
    RotatE.py:107-217).  side in {"s","o"}; ent: (m,K).  Returns (n,m) fp32.
    Rounding points follow the reference's op order (e.g. DistMult rounds
    rel*obj before multiplying by the entity row)."""
    s, p, o, ent = (np.asarray(a, dtype=F32) for a in (s, p, o, ent))
    E = ent[None, :, :]
    if model == "TransE":
        if side == "s":
            return -_sum_k(np.abs(E + (p - o)[:, None, :]))
        return -_sum_k(np.abs((s + p)[:, None, :] - E))
    if model == "DistMult":
        if side == "s":
            return _sum_k(E * (p * o)[:, None, :])
        return _sum_k((s * p)[:, None, :] * E)
    if model in ("ComplEx", "HolE"):
        sr, si = _split(s)
        pr, pi = _split(p)
        orr, oi = _split(o)
        er, ei = _split(E)
        if side == "s":
            a = (pr * orr)[:, None, :] + (pi * oi)[:, None, :]
            b = (pr * oi)[:, None, :] - (pi * orr)[:, None, :]
            sc = _sum_k(er * a + ei * b)
        else:
            a = (sr * pr)[:, None, :] - (si * pi)[:, None, :]
            b = (si * pr)[:, None, :] + (sr * pi)[:, None, :]
            sc = _sum_k(a * er + b * ei)
        if model == "HolE":
            kk = s.shape[-1] // 2
            sc = (F32(2 / kk) * sc).astype(F32)
        return sc
    if model == "RotatE":
        kk = s.shape[-1] // 2
        sr, si = _split(s)
        orr, oi = _split(o)
        pr, pi = _rotate_rel(p, kk, max_rel_size)
        er, ei = _split(E)
        if side == "s":
            re = er * pr[:, None, :] - ei * pi[:, None, :] - orr[:, None, :]
            im = er * pi[:, None, :] + ei * pr[:, None, :] - oi[:, None, :]
        else:
            re = (sr * pr - si * pi)[:, None, :] - er
            im = (sr * pi + si * pr)[:, None, :] - ei
        return -_sum_k(np.sqrt(re * re + im * im))
    raise ValueError(model)


# ----------------------------------------------------------------------------
# a10/a11: ranks
# ----------------------------------------------------------------------------
def quantise(x):
    """AbstractScoringLayer.py:201 `tf.cast(score * 1e3, tf.int32)` (truncation)."""
    with np.errstate(invalid="ignore"):
        return (np.asarray(x, dtype=F32) * COMPARISON_PRECISION).astype(np.int32)


def get_ranks(model, s, p, o, ent_matrix, start_ent_id, end_ent_id, filters,
              mapping=None, corrupt_side="s,o", comparison_type="worst",
              max_rel_size=None):
    """AbstractScoringLayer.get_ranks (AbstractScoringLayer.py:156-422).

    filters: [] (unfiltered) or a list (1 or 2 sides) of per-triple int id arrays.
    mapping: None or dict id->position (the DenseHashTable for entities_subset).
    Returns int32 (sides, n), 0-based like the reference.
    """
    tq = quantise(compute_scores(model, s, p, o, max_rel_size=max_rel_size))
    n = tq.shape[0]
    out = []
    filter_index = 0
    for side in ("s", "o"):
        if side not in corrupt_side:
            continue
        cq = quantise(corruption_scores(model, side, s, p, o, ent_matrix, max_rel_size))
        if comparison_type == "best":  # :221-227
            rank = (tq[:, None] < cq).sum(1).astype(np.int32)
        elif comparison_type == "middle":  # :232-244
            rank = (tq[:, None] < cq).sum(1).astype(np.int32)
            eq = (tq[:, None] == cq).sum(1)
            rank = rank + np.ceil(eq / 2).astype(np.int32)
        else:  # :252-258
            rank = (tq[:, None] <= cq).sum(1).astype(np.int32)
        if len(filters) > 0:
            if side == "o":  # :370-375
                filter_index = 0 if (corrupt_side in ("s", "o") and len(filters) == 1) else 1
            for i in range(n):
                ids = np.asarray(filters[filter_index][i], dtype=np.int64).reshape(-1)
                if mapping:  # :266-275
                    ids = np.array([mapping.get(int(x), -1) for x in ids], dtype=np.int64)
                    ids = ids[ids >= 0]
                ids = ids[(ids >= start_ent_id) & (ids <= end_ent_id)] - start_ent_id  # :280-288
                higher = int((tq[i] <= cq[i, ids]).sum())  # :292-303, always "<="
                rank[i] -= higher
        out.append(rank)
    return np.stack(out).astype(np.int32)


def evaluate_ranks(model, ent, rel, triples, filters_s=None, filters_o=None,
                   corrupt_side="s,o", ranking_strategy="worst", entities_subset=None,
                   max_rel_size=None, batch=256):
    """make_test_function + evaluate glue (ScoringBasedEmbeddingModel.py:1387-1465,
    1672-1692): ranks (n, sides) int32, 1-based; "s+o" sums the sides."""
    triples = np.asarray(triples, dtype=np.int64)
    n = triples.shape[0]
    if entities_subset is not None and len(entities_subset) > 0:
        subset = np.asarray(entities_subset, dtype=np.int64)
        ent_matrix = ent[subset]
        mapping = {}
        for pos, e in enumerate(subset):  # DenseHashTable.insert: last wins
            mapping[int(e)] = pos
    else:
        ent_matrix = ent
        mapping = None
    use_filter = filters_s is not None or filters_o is not None
    out = []
    for b0 in range(0, n, batch):
        tb = triples[b0:b0 + batch]
        s, p, o = lookup(ent, rel, tb)
        if use_filter:
            fl = []
            if "s" in corrupt_side:
                fl.append(filters_s[b0:b0 + batch])
            if "o" in corrupt_side:
                fl.append(filters_o[b0:b0 + batch])
        else:
            fl = []
        r = get_ranks(model, s, p, o, ent_matrix, 0, ent_matrix.shape[0] - 1, fl, mapping,
                      corrupt_side, ranking_strategy, max_rel_size)
        r = r.T
        if corrupt_side == "s+o":
            r = r.sum(1, keepdims=True)
        out.append(r + 1)  # :1684
    if not out:
        return np.zeros((0, 1 if corrupt_side in ("s", "o", "s+o") else 2), dtype=np.int32)
    return np.concatenate(out).astype(np.int32)


def fragile_rank_mask(model, ent, rel, triples, side, rel_tol=4e-6, max_rel_size=None,
                      ent_matrix=None, batch=128):
    """For each test triple: number of corruptions whose quantised comparison with the
    positive could flip under a relative score perturbation of `rel_tol` (fp32
    summation-order noise).  Used by parity tests on non-exact inputs: GPU and oracle
    ranks may differ by at most this many counts for that triple."""
    triples = np.asarray(triples, dtype=np.int64)
    E = ent if ent_matrix is None else ent_matrix
    out = np.zeros(triples.shape[0], dtype=np.int64)
    for b0 in range(0, triples.shape[0], batch):
        s, p, o = lookup(ent, rel, triples[b0:b0 + batch])
        ps = compute_scores(model, s, p, o, max_rel_size=max_rel_size).astype(np.float64) * 1e3
        cs = corruption_scores(model, side, s, p, o, E, max_rel_size).astype(np.float64) * 1e3
        # a comparison is robust if the two truncated values stay ordered for any
        # perturbation: i.e. trunc(ps +- d) vs trunc(cs +- d) cannot change.
        dp = np.abs(ps) * rel_tol + 1e-9
        dc = np.abs(cs) * rel_tol + 1e-9
        lo_p, hi_p = np.trunc(ps - dp), np.trunc(ps + dp)
        lo_c, hi_c = np.trunc(cs - dc), np.trunc(cs + dc)
        # "<=" / "<" / "==" outcome can change only if the intervals of truncated values touch
        touch = (lo_c <= hi_p[:, None]) & (hi_c >= lo_p[:, None])
        unstable = touch & ((lo_c != hi_c) | (lo_p != hi_p)[:, None])
        out[b0:b0 + batch] = unstable.sum(1)
    return out


# ----------------------------------------------------------------------------
# a3: negative sampling (CorruptionGenerationLayerTrain.py:35-94)
# ----------------------------------------------------------------------------
def generate_corruptions(pos, n_ents, eta, seed, step, row_offset=0, b_global=None):
    """Layout: row j*B+i is the j-th corruption of positive i (`tf.tile(pos,[eta,1])`, :52);
    exactly one side replaced (keep_subj ? object<-repl : subject<-repl, :77-88); relation
    untouched; no filtering of true positives.  Random draws come from the shared
    Philox contract (oracle/philox.py) indexed by the *global* corruption row
    j*b_global + (row_offset+i), so that sharding a batch over ranks draws the same
    negatives as one big batch."""
    pos = np.asarray(pos, dtype=np.int32)
    B = pos.shape[0]
    if b_global is None:
        b_global = B
    j = np.repeat(np.arange(eta, dtype=np.uint64), B)
    i = np.tile(np.arange(B, dtype=np.uint64), eta)
    rows = j * np.uint64(b_global) + np.uint64(row_offset) + i
    keep_subj, repl = sample_corruption_draws(rows, step, seed, n_ents)
    data = np.tile(pos, (eta, 1))
    subj = np.where(keep_subj == 1, data[:, 0], repl)
    obj = np.where(keep_subj == 1, repl, data[:, 2])
    return np.stack([subj, data[:, 1], obj], axis=1).astype(np.int32)


# ----------------------------------------------------------------------------
# a12-a16: losses (forward + hand-derived dL/dscore)
# ----------------------------------------------------------------------------
CLIP_LO, CLIP_HI = -75.0, 75.0  # loss_functions.py:32-35
LOSS_DEFAULTS = {
    "pairwise": {"margin": 1.0},           # loss_functions.py:23
    "nll": {},
    "absolute_margin": {"margin": 1.0},
    "self_adversarial": {"margin": 3.0, "alpha": 0.5},  # :26,29
    "multiclass_nll": {},
}


def _log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


def _sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def loss_and_grads(name, scores_pos, scores_neg, eta, params=None, reduction="sum"):
    """Loss.__call__ without regularisation (loss_functions.py:185-225) + per-loss
    `_apply_loss` (:285-308, :359-382, :441-464, :539-574, :629-654).

    scores_neg is the flat (B*eta,) vector; reshaped to (eta, B) as at :211.
    Returns (total fp32 scalar, per_sample (B,), dL/dpos (B,), dL/dneg (B*eta,)) with
    the math done in fp64 from fp32 inputs.
    """
    prm = dict(LOSS_DEFAULTS[name])
    prm.update(params or {})
    P = np.asarray(scores_pos, dtype=F32).astype(np.float64)
    Nn = np.asarray(scores_neg, dtype=F32).astype(np.float64).reshape(eta, -1)
    red_div = 1.0 if reduction == "sum" else float(eta)
    if name == "pairwise":
        h = prm["margin"] - P[None, :] + Nn
        act = (h >= 0)  # tf.maximum passes the gradient to its first argument on ties
        per = np.maximum(h, 0.0).sum(0) / red_div
        dP = -act.sum(0) / red_div
        dN = act / red_div
    elif name == "nll":
        Pc = np.clip(P, CLIP_LO, CLIP_HI)
        Nc = np.clip(Nn, CLIP_LO, CLIP_HI)
        if reduction == "mean":
            red_div = 2.0 * eta  # mean over the concatenated (2*eta, B) tensor, :380-382
        per = (eta * np.log(1 + np.exp(-Pc)) + np.log(1 + np.exp(Nc)).sum(0)) / red_div
        inP = (P >= CLIP_LO) & (P <= CLIP_HI)
        inN = (Nn >= CLIP_LO) & (Nn <= CLIP_HI)
        dP = np.where(inP, -eta * _sigmoid(-Pc), 0.0) / red_div
        dN = np.where(inN, _sigmoid(Nc), 0.0) / red_div
    elif name == "absolute_margin":
        h = prm["margin"] + Nn
        act = (h >= 0)
        per = (np.maximum(h, 0.0) - P[None, :]).sum(0) / red_div
        dP = np.full_like(P, -eta / red_div)
        dN = act / red_div
    elif name == "self_adversarial":
        g, a = prm["margin"], prm["alpha"]
        z = a * Nn
        z = z - z.max(0, keepdims=True)
        w = np.exp(z)
        w /= w.sum(0, keepdims=True)
        ell = _log_sigmoid(-Nn - g)
        lbar = (w * ell).sum(0, keepdims=True)
        per = -_log_sigmoid(g + P) - (w * ell).sum(0) / red_div
        dP = -_sigmoid(-(g + P))
        dN = (w * _sigmoid(Nn + g) - a * w * (ell - lbar)) / red_div
    elif name == "multiclass_nll":
        Pc = np.clip(P, CLIP_LO, CLIP_HI)
        Nc = np.clip(Nn, CLIP_LO, CLIP_HI)
        inP = (P >= CLIP_LO) & (P <= CLIP_HI)
        inN = (Nn >= CLIP_LO) & (Nn <= CLIP_HI)
        eN = np.exp(Nc)
        eP = np.exp(Pc)
        Z = eN.sum(0) / red_div + eP
        per = -np.log(eP / Z)
        dP = np.where(inP, -1.0 + eP / Z, 0.0)
        dN = np.where(inN, eN / Z[None, :] / red_div, 0.0)
    else:
        raise ValueError(name)
    total = F32(per.sum())
    return total, per.astype(F32), dP.astype(F32), dN.reshape(-1).astype(F32)


# ----------------------------------------------------------------------------
# score gradients (derived from the forward code above)
# ----------------------------------------------------------------------------
def score_grads(model, s, p, o, max_rel_size=None):
    """d score / d(s,p,o) rows, fp64 math from fp32 inputs -> three (n,K) fp64 arrays."""
    s, p, o = (np.asarray(a, dtype=F32).astype(np.float64) for a in (s, p, o))
    if model == "TransE":
        d = np.sign(s + p - o)
        return -d, -d, d
    if model == "DistMult":
        return p * o, s * o, s * p
    if model in ("ComplEx", "HolE"):
        sr, si = _split(s)
        pr, pi = _split(p)
        orr, oi = _split(o)
        gs = np.concatenate([pr * orr + pi * oi, pr * oi - pi * orr], -1)
        gp = np.concatenate([sr * orr + si * oi, sr * oi - si * orr], -1)
        go = np.concatenate([sr * pr - si * pi, sr * pi + si * pr], -1)
        if model == "HolE":
            c = float(F32(2 / (s.shape[-1] // 2)))
            gs, gp, go = c * gs, c * gp, c * go
        return gs, gp, go
    if model == "RotatE":
        kk = s.shape[-1] // 2
        sr, si = _split(s)
        orr, oi = _split(o)
        div = float(rotate_phase_divisor(kk, max_rel_size))
        phi = (p[..., :kk].astype(F32) / F32(div)).astype(np.float64)
        c, sn = np.cos(phi), np.sin(phi)
        re = sr * c - si * sn - orr
        im = sr * sn + si * c - oi
        m = np.sqrt(re * re + im * im)
        with np.errstate(divide="ignore", invalid="ignore"):
            gs = np.concatenate([-(re * c + im * sn) / m, -(-re * sn + im * c) / m], -1)
            gth = -(re * (-sr * sn - si * c) + im * (sr * c - si * sn)) / m / div
            go = np.concatenate([re / m, im / m], -1)
        gp = np.concatenate([gth, np.zeros_like(gth)], -1)
        return gs, gp, go
    raise ValueError(model)


# ----------------------------------------------------------------------------
# a2: initialiser (Keras GlorotUniform formula; stream is ours: numpy PCG64)
# ----------------------------------------------------------------------------
def glorot_uniform(rows, cols, rng):
    lim = math.sqrt(6.0 / (rows + cols))
    return rng.uniform(-lim, lim, size=(rows, cols)).astype(F32)


# ----------------------------------------------------------------------------
# a17/a18/a19: one training step with dense Keras-legacy optimizer semantics
# ----------------------------------------------------------------------------
# optimizer state tensors per table ("kind" = update rule: SGD with momentum is "momentum", RMSprop with momentum "rmsprop_mom")
OPT_SLOTS = {"sgd": (), "adagrad": ("a",), "adam": ("m", "v"), "momentum": ("mom",), "rmsprop": ("rms",),
             "rmsprop_mom": ("rms", "mom"), "adadelta": ("acc", "dacc"), "adamax": ("m", "u")}


class TrainState:
    def __init__(self, ent, rel, optimizer="adam", lr=1e-3, **hp):
        """hp: momentum, nesterov, rho, beta_1, beta_2, epsilon (Keras legacy defaults when absent)."""
        self.ent = np.array(ent, dtype=F32)
        self.rel = np.array(rel, dtype=F32)
        self.optimizer = optimizer
        self.lr = lr
        self.hp = dict(hp)
        self.iterations = 0
        self.slots = {}
        for nme in OPT_SLOTS[optimizer]:   # Keras legacy Adagrad: initial_accumulator_value = 0.1
            for tab, arr in (("e", self.ent), ("r", self.rel)):
                self.slots[f"{nme}_{tab}"] = np.full_like(arr, 0.1) if nme == "a" else np.zeros_like(arr)


def focus_transform(x, weight, non_linearity):
    """FocusE: y = f(x) * weight and dy/dx (ScoringBasedEmbeddingModel.py:396-406, non-linearities :492-513; the
    reference's "softplus" is log(1 + 9999 e^x) with custom gradient 1 - 1/(1 + 9999 e^x))."""
    x = np.asarray(x, dtype=F32).astype(np.float64)
    if non_linearity == "linear":
        f, fp = x, np.ones_like(x)
    elif non_linearity == "tanh":
        f = np.tanh(x); fp = 1.0 - f * f
    elif non_linearity == "sigmoid":
        f = 1.0 / (1.0 + np.exp(-x)); fp = f * (1.0 - f)
    elif non_linearity == "softplus":
        e = 9999.0 * np.exp(x); f = np.log(1.0 + e); fp = 1.0 - 1.0 / (1.0 + e)
    else:
        raise ValueError("Invalid focusE non-linearity")
    return (f * weight).astype(F32), fp * weight


def focus_weights(w_mean, beta, eta):
    """compute_focusE_weights (:342-368): (weights_pos (B,), weights_neg (B*eta,) in corruption-row order j*B+i)."""
    w = np.asarray(w_mean, dtype=np.float64)
    return beta + (1.0 - beta) * (1.0 - w), np.tile(beta + (1.0 - beta) * w, eta)


def dense_gradients(model, ent, rel, pos, negs, eta, loss_name, loss_params=None,
                    reduction="sum", max_rel_size=None, reg=None, focus=None, coeffs=None):
    """Forward + backward of ScoringBasedEmbeddingModel.train_step (:370-429): returns
    (total loss fp32, G_ent fp64, G_rel fp64).  Duplicate row ids are summed (Keras
    IndexedSlices dedup).  reg = None or dict(p=..., lam_e=..., lam_r=...) following
    regularizers.py:35-37 applied to the whole tables."""
    pos = np.asarray(pos, dtype=np.int64)
    negs = np.asarray(negs, dtype=np.int64)
    s, p, o = lookup(ent, rel, pos)
    ns, npred, no = lookup(ent, rel, negs)
    sp = compute_scores(model, s, p, o, max_rel_size=max_rel_size)
    sn = compute_scores(model, ns, npred, no, max_rel_size=max_rel_size)
    fac_p = fac_n = None
    if focus is not None:   # focus = (per-positive mean weight (B,), beta, non-linearity name)
        wp, wn = focus_weights(focus[0], focus[1], eta)
        sp, fac_p = focus_transform(sp, wp, focus[2])
        sn, fac_n = focus_transform(sn, wn, focus[2])
    total, per, dP, dN = loss_and_grads(loss_name, sp, sn, eta, loss_params, reduction)
    if focus is not None:
        dP, dN = dP * fac_p, dN * fac_n
    if coeffs is not None:   # (a dict to fill: the loss coefficients dL/dscore of the positives and of the corruptions)
        coeffs["dP"], coeffs["dN"] = np.asarray(dP, dtype=np.float64), np.asarray(dN, dtype=np.float64)
    Ge = np.zeros(ent.shape, dtype=np.float64)
    Gr = np.zeros(rel.shape, dtype=np.float64)
    for tri, (a, b, c), g in ((pos, (s, p, o), dP), (negs, (ns, npred, no), dN)):
        gs, gp, go = score_grads(model, a, b, c, max_rel_size)
        g = g.astype(np.float64)[:, None]
        np.add.at(Ge, tri[:, 0], g * gs)
        np.add.at(Gr, tri[:, 1], g * gp)
        np.add.at(Ge, tri[:, 2], g * go)
    total = float(total)
    if reg is not None:
        for tab, G, terms in zip((ent, rel), (Ge, Gr), reg_terms(reg)):
            for pw, lam in terms:
                x = tab.astype(np.float64)
                total += lam * float((np.abs(x) ** pw).sum())
                G += lam * pw * np.abs(x) ** (pw - 1) * np.sign(x)
    return F32(total), Ge, Gr, (sp, sn, per)


def reg_terms(reg):
    """Regulariser description -> ([(p, lambda), ...] of the entity table, [...] of the relation table).

    reg = dict(p=..., lam_e=..., lam_r=...): one LP term with a shared p (regularizers.py:14-37); optional keys
    terms_e / terms_r = [(p, lambda), ...] replace a table's term list: the reference hands [entity, relation] pairs of
    independent Keras regularisers to the lookup layer (EmbeddingLookupLayer.py:131-155), and tf.keras 'l1_l2' is two terms."""
    if reg is None:
        return [], []
    pw = reg.get("p", 2)
    te = reg.get("terms_e", [(pw, reg.get("lam_e", 0.0))])
    tr = reg.get("terms_r", [(pw, reg.get("lam_r", 0.0))])
    return [(int(p_), float(l_)) for p_, l_ in te if l_], [(int(p_), float(l_)) for p_, l_ in tr if l_]


def apply_optimizer(state, Ge, Gr, beta1=None, beta2=None, eps=None):
    """Keras *legacy* update rules (tensorflow==2.15 keras/optimizers/legacy/{adam,adagrad,gradient_descent,rmsprop,
    adadelta,adamax}.py and the fused kernels they call, core/kernels/training_ops.cc; third-party, not vendored in
    /root/reference -- parity unpinned).  Dense over all rows, i.e. the non-lazy behaviour of optimizer_v2
    Adam._resource_apply_sparse: every row's state decays and every row moves each step.  Reached from optimizers.py:166-168
    (any legacy optimizer name is accepted there, :57-67)."""
    hp = getattr(state, "hp", {})
    b1 = float(hp.get("beta_1", 0.9) if beta1 is None else beta1)
    b2 = float(hp.get("beta_2", 0.999) if beta2 is None else beta2)
    beta1, beta2 = F32(b1), F32(b2)
    eps = F32(hp.get("epsilon", 1e-7) if eps is None else eps)
    state.iterations += 1
    t = state.iterations
    lr = F32(state.lr)
    kind = state.optimizer
    tabs = ((state.ent, Ge.astype(F32), "e"), (state.rel, Gr.astype(F32), "r"))
    sl = lambda nme, tab: state.slots[f"{nme}_{tab}"]   # noqa: E731
    if kind == "adam":
        lr_t = F32(float(lr) * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
        # one_minus_beta_t = 1 - beta_t is formed from the fp32 hyper-parameter tensors (adam.py _prepare_local)
        omb1, omb2 = F32(1) - beta1, F32(1) - beta2
        for x, g, tab in tabs:
            m, v = sl("m", tab), sl("v", tab)
            m[...] = m * beta1 + g * omb1
            v[...] = v * beta2 + (g * g) * omb2
            x -= (lr_t * m) / (np.sqrt(v) + eps)
    elif kind == "adagrad":
        for x, g, tab in tabs:
            a = sl("a", tab)
            a += g * g
            x -= lr * g / (np.sqrt(a) + eps)
    elif kind == "sgd":
        for x, g, tab in tabs:
            x -= lr * g
    elif kind == "momentum":      # ResourceApplyKerasMomentum
        mom, nesterov = F32(hp.get("momentum", 0.9)), bool(hp.get("nesterov", False))
        for x, g, tab in tabs:
            a = sl("mom", tab)
            a[...] = a * mom - lr * g
            x += (a * mom - lr * g) if nesterov else a
    elif kind in ("rmsprop", "rmsprop_mom"):
        rho = F32(hp.get("rho", 0.9))
        omr = F32(1.0 - float(rho))
        for x, g, tab in tabs:
            r = sl("rms", tab)
            r += (g * g - r) * omr
            if kind == "rmsprop":   # RMSprop._resource_apply_dense without momentum: epsilon outside the root
                x -= lr * g / (np.sqrt(r) + eps)
            else:                   # ResourceApplyRMSProp: epsilon inside the root
                mo = sl("mom", tab)
                mo[...] = mo * F32(hp.get("momentum", 0.9)) + (lr * g) / np.sqrt(r + eps)
                x -= mo
    elif kind == "adadelta":      # ResourceApplyAdadelta
        rho = F32(hp.get("rho", 0.95))
        omr = F32(1.0 - float(rho))
        for x, g, tab in tabs:
            a, d = sl("acc", tab), sl("dacc", tab)
            a[...] = a * rho + (g * g) * omr
            u = np.sqrt(d + eps) * (F32(1.0) / np.sqrt(a + eps)) * g
            x -= u * lr
            d[...] = d * rho + (u * u) * omr
    elif kind == "adamax":        # ResourceApplyAdaMax
        lr_t = F32(float(lr) / (1.0 - float(beta1) ** t))
        omb1 = F32(1.0 - float(beta1))
        for x, g, tab in tabs:
            m, u = sl("m", tab), sl("u", tab)
            m += (g - m) * omb1
            u[...] = np.maximum(beta2 * u, np.abs(g))
            x -= lr_t * m / (u + eps)
    else:
        raise ValueError(state.optimizer)


def touched_rows(n_ents, pos, negs, dN, scale=1.0, nonfinite=None):
    """Which ENTITY rows a step touches in touched-rows mode (include/amdkge.h, amdkge_opt.lazy): the s and o of every positive,
    and the replacement row of every corruption whose coefficient g = dL/dscore * score_sign * score_scale is at least fp32's
    smallest NORMAL number in magnitude -- the forward kernel drops every other entry (kge_train_kernel.h: "inactive margin /
    clipped corruption / a coefficient that underflows fp32").  A coefficient the fp64
    restatement still resolves (1e-41 for a corruption that scores 90 below its positive) therefore does NOT touch its row --
    the difference VERDICT r4 #9 asked about: RotatE k = 1000 under rules without damping reaches that regime at the third step
    (profiles/r05a_diag_rotate_rules2.jsonl: the 12 rows the fp64 mask moved and the engine did not all had |g| < 4e-39).
    Round 6: a NaN coefficient is an entry (the kernel's test is `!(|g| < tiny)`), and so is the masked zero of a corruption whose
    score -- or the hinge argument it enters -- is not finite (`nonfinite`, (B*eta,) bool; kge_train_kernel.h masked_zero): the
    reference multiplies that zero by the score's Jacobian, 0 * NaN = NaN."""
    pos, negs = np.asarray(pos, dtype=np.int64), np.asarray(negs, dtype=np.int64)
    mask = np.zeros(n_ents, dtype=bool)
    mask[pos[:, 0]] = True
    mask[pos[:, 2]] = True
    if len(negs):
        data = np.tile(pos, (len(negs) // max(len(pos), 1), 1))
        # (the entry's coefficient g = dL/dscore * score_sign * score_scale -- HolE: 2 / k -- against fp32's smallest normal number)
        with np.errstate(invalid="ignore"):
            live = ~(np.abs(np.asarray(dN, dtype=np.float64) * float(scale)) < float(np.finfo(np.float32).tiny))
        if nonfinite is not None:
            live |= np.asarray(nonfinite, dtype=bool)
        repl = np.where(negs[:, 0] != data[:, 0], negs[:, 0], negs[:, 2])   # (a corruption that redraws the same id: its own row)
        mask[repl[live]] = True
    return mask


def apply_optimizer_lazy(state, Ge, Gr, reg=None, ent_mask=None):
    """Touched-rows mode of the engine (amdkge_opt.lazy, include/amdkge.h) -- NOT a reference behaviour: the reference's
    optimizer is dense (optimizers.py:136-168).  Untouched rows keep x and every slot; the other rows get the regulariser
    gradient and the ordinary update rule (TF-Addons LazyAdam semantics, generalised).  Entity rows: ent_mask (touched_rows
    above: rows that received an entry); without it, and for the relation table (whose sweep reads the accumulated gradient
    row, kge_opt.h opt_rows_kernel): rows whose data-gradient row is not entirely zero in fp32.
    Ge, Gr: data gradients WITHOUT the regulariser term.  Returns the regulariser loss over the touched rows."""
    Ge, Gr = np.array(Ge, dtype=np.float64), np.array(Gr, dtype=np.float64)
    tiny = float(np.finfo(np.float32).tiny)
    masks = [np.any(np.abs(Ge) >= tiny, axis=1) if ent_mask is None else np.asarray(ent_mask, dtype=bool), np.any(np.abs(Gr) >= tiny, axis=1)]
    reg_loss = 0.0
    if reg is not None:
        for x, G, mask, terms in zip((state.ent, state.rel), (Ge, Gr), masks, reg_terms(reg)):
            xx = x.astype(np.float64)[mask]
            for pw, lam in terms:
                reg_loss += lam * float((np.abs(xx) ** pw).sum())
                G[mask] += lam * pw * np.abs(xx) ** (pw - 1) * np.sign(xx)
    keep = [(state.ent.copy(), {k: v.copy() for k, v in state.slots.items() if k.endswith("_e")}),
            (state.rel.copy(), {k: v.copy() for k, v in state.slots.items() if k.endswith("_r")})]
    apply_optimizer(state, Ge, Gr)
    for (x0, sl0), x, mask in zip(keep, (state.ent, state.rel), masks):
        x[~mask] = x0[~mask]
        for k, v in sl0.items():
            state.slots[k][~mask] = v[~mask]
    return reg_loss


def train_step(state, model, pos, eta, loss_name, seed, step, n_ents=None, loss_params=None,
               reduction="sum", max_rel_size=None, reg=None, row_offset=0, b_global=None,
               negs=None, focus=None, lazy=False):
    if n_ents is None:
        n_ents = state.ent.shape[0]
    if negs is None:
        negs = generate_corruptions(pos, n_ents, eta, seed, step, row_offset, b_global)
    if lazy:
        co = {}
        loss, Ge, Gr, (sp, sn, _) = dense_gradients(model, state.ent, state.rel, pos, negs, eta, loss_name,
                                                    loss_params, reduction, max_rel_size, None, focus, coeffs=co)
        scale = 2.0 / float(state.ent.shape[1] // 2) if model == "HolE" else 1.0
        # (the hinge argument of the pairwise loss holds the positive's score too; self_adversarial has no mask)
        nonfin = ~np.isfinite(sn) | (np.tile(~np.isfinite(sp), eta) if loss_name == "pairwise" else False)
        return float(loss) + apply_optimizer_lazy(state, Ge, Gr, reg, touched_rows(state.ent.shape[0], pos, negs, co["dN"], scale,
                                                                                   None if loss_name == "self_adversarial" else nonfin))
    loss, Ge, Gr, _ = dense_gradients(model, state.ent, state.rel, pos, negs, eta, loss_name,
                                      loss_params, reduction, max_rel_size, reg, focus)
    apply_optimizer(state, Ge, Gr)
    return loss


# ----------------------------------------------------------------------------
# calibration: CalibrationLayer (layers/calibration/calibrate.py:33-129)
# ----------------------------------------------------------------------------
def platt_init(pos_size, neg_size=0, positive_base_rate=None):
    """w, b, (label_pos, label_neg), neg_size, base rate as CalibrationLayer.__init__ / call set them (:33-56,95-106)."""
    neg_size = pos_size if neg_size == 0 else neg_size
    if positive_base_rate is not None:
        if positive_base_rate <= 0 or positive_base_rate >= 1:
            raise ValueError("Positive_base_rate must be a value between 0 and 1.")
    else:
        assert pos_size > 0 and neg_size > 0, "Positive size must be > 0."
        positive_base_rate = pos_size / (pos_size + neg_size)
    b0 = float(F32(math.log((neg_size + 1.0) / (pos_size + 1.0))))
    labels = (F32((pos_size + 1.0) / (pos_size + 2.0)), F32(1.0 / (neg_size + 2.0)))
    return 0.0, b0, labels, neg_size, positive_base_rate


def platt_proba(scores, w, b):
    """CalibrationLayer.call(training=0) (:90,128-129): sigmoid(-(w*s + b))."""
    x = -(F32(w) * np.asarray(scores, dtype=F32) + F32(b))
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def platt_loss_and_grads(scores_pos, scores_neg, w, b, labels, positive_base_rate):
    """CalibrationLayer.call(training=1) (:84-127) and its gradient w.r.t. (w, b): weighted mean of
    tf.nn.sigmoid_cross_entropy_with_logits(labels, logits) = max(x,0) - x*z + log(1+exp(-|x|))."""
    sp, sn = np.asarray(scores_pos, dtype=np.float64), np.asarray(scores_neg, dtype=np.float64)
    s = np.concatenate([sp, sn])
    z = np.concatenate([np.full(sp.shape, float(labels[0])), np.full(sn.shape, float(labels[1]))])
    wt = np.concatenate([np.full(sp.shape, sn.shape[0] / sp.shape[0]),
                         np.full(sn.shape, (1.0 - positive_base_rate) / positive_base_rate)])
    x = -(float(w) * s + float(b))
    loss = np.mean(wt * (np.maximum(x, 0.0) - x * z + np.log1p(np.exp(-np.abs(x)))))
    d = wt * (1.0 / (1.0 + np.exp(-x)) - z) / s.shape[0]
    return float(loss), float(np.sum(-s * d)), float(np.sum(-d))


def adam_scalar_step(params, grads, slots, t, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam() defaults (third-party, TF 2.15) on python floats: used by calibrate (:2062)."""
    alpha = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    out = []
    for i, (p, g) in enumerate(zip(params, grads)):
        m, v = slots[i]
        m += (g - m) * (1.0 - beta1)
        v += (g * g - v) * (1.0 - beta2)
        slots[i] = (m, v)
        out.append(p - m * alpha / (math.sqrt(v) + eps))
    return out


# ----------------------------------------------------------------------------
# a21/a22: host data semantics
# ----------------------------------------------------------------------------
def first_seen_index(triples):
    """InMemory.update_dictionary_mappings (datasets/data_indexer.py:373-399): scan rows,
    subject then object get the next entity id when first seen; relations separately."""
    ents, rels = {}, {}
    for s, p, o in triples:
        if s not in ents:
            ents[s] = len(ents)
        if o not in ents:
            ents[o] = len(ents)
        if p not in rels:
            rels[p] = len(rels)
    return ents, rels


def to_indexes(triples, ents, rels):
    """get_indexes_from_a_dictionary (data_indexer.py:485-549): rows with unknown keys dropped."""
    out = []
    for s, p, o in triples:
        if s in ents and p in rels and o in ents:
            out.append((ents[s], rels[p], ents[o]))
    return np.array(out, dtype=np.int32).reshape(-1, 3)


def filter_sets(test, filter_datasets):
    """_get_complementary_subjects/_objects (datasets/graph_data_loader.py:287-350,382-439):
    per test triple the *set* of s' with (s',p,o) in any filter dataset, resp. o' with
    (s,p,o')."""
    po, sp = {}, {}
    for D in filter_datasets:
        for s, p, o in np.asarray(D)[:, :3]:
            po.setdefault((int(p), int(o)), set()).add(int(s))
            sp.setdefault((int(s), int(p)), set()).add(int(o))
    fs, fo = [], []
    for s, p, o in np.asarray(test)[:, :3]:
        fs.append(np.array(sorted(po.get((int(p), int(o)), ())), dtype=np.int32))
        fo.append(np.array(sorted(sp.get((int(s), int(p)), ())), dtype=np.int32))
    return fs, fo


# ----------------------------------------------------------------------------
# evaluation/metrics.py:58-62,108-112,188-192
# ----------------------------------------------------------------------------
def mrr_score(ranks):
    r = np.asarray(ranks).reshape(-1)
    return float(np.sum(1.0 / r) / len(r))


def mr_score(ranks):
    r = np.asarray(ranks).reshape(-1)
    return float(np.sum(r) / len(r))


def hits_at_n_score(ranks, n):
    r = np.asarray(ranks).reshape(-1)
    return float(np.sum(r <= n) / len(r))
