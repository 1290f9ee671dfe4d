"""TEST INFRASTRUCTURE -- second oracle mode for fit(): one training step in the kernels' DECLARED fp32 order.

`oracle/kge_oracle.py` restates ScoringBasedEmbeddingModel.train_step (ScoringBasedEmbeddingModel.py:370-429) with fp64
accumulation.  For the models that are smooth in their parameters that is enough: GPU and oracle trajectories stay within
1e-5.  TransE is not smooth (d|x|/dx = sign(x), TransE.py:51-53; the pairwise hinge, loss_functions.py:302-308, adds a second
discontinuity): a score that differs in its last bit flips a hinge term, an embedding that differs in its last bit flips a sign,
and two fp32 evaluations of one schedule in different summation orders part (tests/test_gpu_learning.py measures it).  As for
the ranks (oracle/rank_ordered.py) ONE order has to be fixed for "identical" to be testable -- the one the train kernels declare
(ampligraph_amd/csrc/kge_train_kernel.h, kge_train_tiled.hip, kge_opt.h):

  * d = fl(fl(s + p) - o) per unit (TransE.py:51-53), for a corruption fl(fl(s + p) - e) (object replaced) or
    fl(fl(e + p) - o) (subject replaced), rounded where the reference rounds;
  * a score = -(sum of |d|): lane l of a wave holds units 4 l .. 4 l + 3 and adds them in that order, the 64 lane sums are
    combined by the wave64 DPP tree (rank_ordered.wave_sum); rows of up to 256 units (one quad per lane);
  * the pairwise hinge in fp32: h_j = fl(fl(margin - P) + n_j), active iff h_j >= 0 (tf.maximum passes the gradient to its
    first argument on ties), per-positive loss = the wave64 tree over the lanes' max(h_j, 0) (lane j % 64 adds its terms in
    increasing j), reduction "sum";
  * with that loss every gradient entry is an INTEGER (a signed count of active hinge terms per unit: dL/dscore is -1 / +1,
    d score / d row is -/+ sign(d)), so the row sums are exact in fp32 in ANY order -- bucket order, atomics and the
    deterministic mode's sorted order all give the same bits;
  * Adam per element as kge_opt.h opt_elem writes it (Keras legacy rule, optimizers.py:136-168): m = fl(fl(m b1) + fl(g (1 - b1))),
    v = fl(fl(v b2) + fl(fl(g g) (1 - b2))), x = fl(x - fl(fl(lr_t m) / fl(sqrt(v) + eps))), lr_t formed in fp64 from the fp32
    hyper-parameters and rounded once, dense over both tables.

Only the fp64 sum of the per-positive losses is order-dependent (~1e-16 relative; nothing feeds back).  Pinned on the CPU
(tests/test_oracle_train_ordered.py): equal to kge_oracle.train_step wherever no decision is within rounding of its boundary
(dyadic tables: bit for bit), and the wave tree against an independent restatement.
"""
import math

import numpy as np

from . import kge_oracle as O
from .philox import sample_corruption_draws
from .rank_ordered import wave_sum

F32 = np.float32


def _lane_sums(absd):
    """[n, K] per-unit |d| -> [n, 64]: lane l adds its units 4 l .. 4 l + 3 in order (from 0); K <= 256, K % 4 == 0."""
    n, K = absd.shape
    assert K % 4 == 0 and K <= 256, "ordered TransE step: rows of up to 256 units (one quad per lane)"
    q = absd.reshape(n, K // 4, 4)
    acc = np.zeros((n, K // 4), dtype=F32)
    for u in range(4):
        acc = (acc + q[:, :, u]).astype(F32)
    lanes = np.zeros((n, 64), dtype=F32)
    lanes[:, :K // 4] = acc
    return lanes


def transe_scores(s, p, o):
    """-(sum |fl(fl(s + p) - o)|) in the declared order; s, p, o fp32 [n, K]."""
    d = ((s + p).astype(F32) - o).astype(F32)
    return (F32(-1.0) * wave_sum(_lane_sums(np.abs(d)))).astype(F32), d


class AdamState:
    """Both tables with their Adam slots, fp32; lr, b1, b2, eps are the fp32 hyper-parameters the descriptor carries."""

    def __init__(self, ent, rel, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.ent, self.rel = np.array(ent, dtype=F32), np.array(rel, dtype=F32)
        self.m = [np.zeros_like(self.ent), np.zeros_like(self.rel)]
        self.v = [np.zeros_like(self.ent), np.zeros_like(self.rel)]
        self.lr, self.b1, self.b2, self.eps = F32(lr), F32(beta1), F32(beta2), F32(eps)
        self.iterations = 0

    def apply(self, Ge, Gr):
        """kge_opt.h: fill_opt_args (lr_t, 1 - beta in fp64 from the fp32 values, rounded once) + opt_elem<ADAM>."""
        self.iterations += 1
        t = float(self.iterations)
        b1, b2, lr = float(self.b1), float(self.b2), float(self.lr)
        lr_t = F32(lr * math.sqrt(1.0 - math.pow(b2, t)) / (1.0 - math.pow(b1, t)))
        omb1, omb2 = F32(1.0 - b1), F32(1.0 - b2)
        for x, g, m, v in ((self.ent, Ge, self.m[0], self.v[0]), (self.rel, Gr, self.m[1], self.v[1])):
            g = g.astype(F32)
            m[...] = ((m * self.b1).astype(F32) + (g * omb1).astype(F32)).astype(F32)
            v[...] = ((v * self.b2).astype(F32) + ((g * g).astype(F32) * omb2).astype(F32)).astype(F32)
            x[...] = (x - ((lr_t * m).astype(F32) / (np.sqrt(v).astype(F32) + self.eps).astype(F32)).astype(F32)).astype(F32)


def transe_pairwise_step(state, pos, eta, seed, step, margin=1.0, n_ents=None, row_offset=0, b_global=None, return_grads=False):
    """One step of TransE / pairwise (reduction "sum") / Adam on `pos` (int [B, 3]) in the declared order; corruptions from the
    shared Philox contract (oracle/philox.py, rows j * b_global + row_offset + i).  Updates `state` in place; returns the
    batch loss (fp64 sum of the fp32 per-positive losses)."""
    pos = np.asarray(pos, dtype=np.int64)
    B = pos.shape[0]
    ent, rel = state.ent, state.rel
    N = ent.shape[0] if n_ents is None else int(n_ents)
    bg = B if b_global is None else int(b_global)
    margin = F32(margin)
    s, p, o = ent[pos[:, 0]], rel[pos[:, 1]], ent[pos[:, 2]]
    P, d_pos = transe_scores(s, p, o)
    Ge = np.zeros(ent.shape, dtype=np.float64)   # integers throughout: exact in fp32 as in fp64, in any order
    Gr = np.zeros(rel.shape, dtype=np.float64)
    n_act = np.zeros(B, dtype=np.int64)
    h_lanes = np.zeros((B, 64), dtype=F32)   # lane j % 64 adds max(h_j, 0) for its j in increasing order
    sp = (s + p).astype(F32)
    mP = (margin - P).astype(F32)
    for j in range(eta):
        rows = np.uint64(j) * np.uint64(bg) + np.uint64(row_offset) + np.arange(B, dtype=np.uint64)
        keep, repl = sample_corruption_draws(rows, step, seed, N)
        e = ent[repl]
        keep = keep.astype(bool)
        # object replaced: fl(fl(s + p) - e); subject replaced: fl(fl(e + p) - o)
        d = np.where(keep[:, None], (sp - e).astype(F32), ((e + p).astype(F32) - o).astype(F32))
        n_j = (F32(-1.0) * wave_sum(_lane_sums(np.abs(d)))).astype(F32)
        h = (mP + n_j).astype(F32)
        act = h >= 0
        h_lanes[:, j % 64] = (h_lanes[:, j % 64] + np.maximum(h, F32(0))).astype(F32)
        n_act += act
        # dL/dn_j = +1 for an active term; d n_j / d(row) = -/+ sign(d): replaced row gets +sign(d) (object) or -sign(d) (subject)
        sg = np.sign(d).astype(np.float64) * act[:, None]
        np.add.at(Ge, repl[keep], sg[keep])            # (s, p, e): d/de = +sign(d)
        np.add.at(Ge, pos[keep, 0], -sg[keep])         #            d/ds = -sign(d)
        np.add.at(Ge, repl[~keep], -sg[~keep])         # (e, p, o): d/de = -sign(d)
        np.add.at(Ge, pos[~keep, 2], sg[~keep])        #            d/do = +sign(d)
        np.add.at(Gr, pos[:, 1], -sg)                  # d/dp = -sign(d) on either side
    # positive: dL/dP = -(number of active terms); d P / d(s, p, o) = (-, -, +) sign(d)
    sgp = np.sign(d_pos).astype(np.float64) * n_act[:, None].astype(np.float64)
    np.add.at(Ge, pos[:, 0], sgp)
    np.add.at(Gr, pos[:, 1], sgp)
    np.add.at(Ge, pos[:, 2], -sgp)
    per = wave_sum(h_lanes)
    assert np.abs(Ge).max() < 2 ** 24 and np.abs(Gr).max() < 2 ** 24
    if return_grads:   # (tests: the step's ingredients, nothing applied)
        return float(per.astype(np.float64).sum()), Ge, Gr
    state.apply(Ge, Gr)
    return float(per.astype(np.float64).sum())


def replay_learning(model, loss, seed, cfg, planted_kg, initialise, epochs=None):
    """The schedule of tests/test_gpu_learning.py (planted graph, Glorot tables as the drop-in class draws them, sequential
    batches) through transe_pairwise_step -> (loss history, state, id triples of train / test)."""
    assert model == "TransE" and loss == "pairwise"
    d = planted_kg(model, seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)
    ents, rels = O.first_seen_index(train)
    Xi = O.to_indexes(train, ents, rels)
    N, R, K = len(ents), len(rels), O.internal_k(model, cfg["k"])
    rng = np.random.Generator(np.random.PCG64(seed))
    st = AdamState(initialise("glorot_uniform", (N, K), rng), initialise("glorot_uniform", (R, K), rng), cfg["lr"])
    steps = (len(Xi) + cfg["batch"] - 1) // cfg["batch"]
    hist = []
    for ep in range(cfg["epochs"] if epochs is None else epochs):
        tot = 0.0
        for b in range(steps):
            tot += transe_pairwise_step(st, Xi[b * cfg["batch"]:(b + 1) * cfg["batch"]], cfg["eta"], seed, ep * steps + b)
        hist.append(tot / steps)
    return np.asarray(hist), st, Xi, O.to_indexes(test, ents, rels)
