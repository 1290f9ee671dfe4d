"""End-to-end LEARNING parity: fit() + evaluate() of the drop-in class on the GPU against the oracle replaying the same
schedule, on a planted-structure graph where filtered MRR really rises (tests/planted.py) -- loss history within 1e-4 and
filtered MRR within +-0.002 (BASELINE.json north_star) for the models that are smooth in their parameters, measured bounds for
TransE (see CASES), several seeds."""
import numpy as np
import pytest
from margins import within

from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu

EPOCHS, BATCH, ETA, K, LR = 40, 1024, 5, 16, 2e-2


# Tolerances.  Contraction models (smooth in the parameters): loss history within 1e-4 (measured <= 1.1e-5 over 160 Adam
# steps) and filtered MRR within +-0.002 (measured <= 1.3e-6) -- the north_star's bars.  TransE is not smooth: d|x|/dx =
# sign(x) flips wherever a unit of s + p - o sits within fp32 summation noise of 0, and the pairwise hinge adds a second
# discontinuity, so two fp32 evaluations of the SAME schedule in different summation orders follow diverging trajectories
# (measured on MI355X vs the oracle: first 5 epochs within 1e-5 (nll) / 4e-4 (pairwise), after 40 epochs 4.7e-4..6.8e-4 / 0.5..0.8 %
# in the loss, 0.002..0.008 in MRR, both runs equally good).  The reference has the same property between its own CPU and GPU
# kernels.  The bars below are the measured drift with headroom, and the early epochs are held tight.
CASES = [("ComplEx", "multiclass_nll", 1e-4, 1e-4, 2e-3, 0.15), ("DistMult", "self_adversarial", 1e-4, 1e-4, 2e-3, 0.05),
         ("TransE", "nll", 2e-3, 5e-5, 2.5e-2, 0.15), ("TransE", "pairwise", 2e-2, 2e-3, 2.5e-2, 0.15),
         # round 3: the two remaining models.  HolE = ComplEx's score scaled by 2/k: same bars (measured 1e-5 / 4e-4 MRR).  RotatE's
         # gradient z / |z| is ill-conditioned where a unit's modulus is ~0 (no epsilon, RotatE.py:102-104), so two fp32 / fp64
         # evaluations of the same schedule part slowly, like TransE's: measured on MI355X vs the fp64-accumulating oracle, first 5
         # epochs within 1e-6 for both losses; after 160 Adam steps self_adversarial (configs[4]'s loss) 2.4e-5 .. 3.0e-4 in the
         # loss and 8e-4 .. 2.1e-3 in MRR -- at the north_star's +-0.002, bar = measured x 2.  With nll this graph trains into a
         # regime where MRR itself is chaotic: loss within 0.15 .. 0.32 %, but MRR 0.22 .. 0.30 across the ORACLE's own three seeds
         # and 0.01 .. 0.05 between the two runs of one seed; there the test holds the loss and "both learn", not an MRR distance.
         # Round 4: the per-seed MRR bars of the distance models are what ONE seed can be held to -- measured over 512 seeds on MI355X
         # (test_mean_mrr_over_seeds_matches_oracle, profiles/r04b_pytest_gpu.log): per-seed |MRR(GPU) - MRR(oracle)| sd 0.0064 / 0.0055 /
         # 0.0023, max 0.0225 / 0.0214 / 0.0205 (TransE nll / TransE pairwise / RotatE self_adversarial; heavy-tailed: a trajectory
         # that parts early parts far), and the default mode is not bitwise reproducible, so a bar at twice a 3-seed maximum (round 3:
         # 0.004 for RotatE) fails one run in a dozen.  The north_star's +-0.002 is asserted on the MEAN over the seeds, below.
         ("HolE", "self_adversarial", 1e-4, 1e-4, 2e-3, 0.05), ("RotatE", "self_adversarial", 1e-3, 1e-5, 2.5e-2, 0.05),
         ("RotatE", "nll", 1e-2, 1e-5, None, 0.05)]


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("model,loss,loss_tol,early_tol,mrr_tol,mrr_min", CASES)
def test_fit_learns_and_matches_oracle_replay(gpu_lib, model, loss, loss_tol, early_tol, mrr_tol, mrr_min, seed):
    from planted import planted_kg
    from test_gpu_model import oracle_replay

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers

    d = planted_kg(model, seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)   # labels, like the reference's datasets
    m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type=model, seed=seed)
    m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=loss)
    h = m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False)
    st, Xi, hist = oracle_replay(model, train, K, ETA, loss, "adam", LR, BATCH, EPOCHS, seed=seed)
    got = np.asarray(h.history["loss"])
    drift = float(np.max(np.abs(got - hist) / np.abs(hist)))
    # filtered evaluation, GPU tables through the GPU path vs the oracle's tables through the oracle
    ranks = m.evaluate(test, use_filter={"train": train, "test": test}, corrupt_side="s,o", verbose=False)
    ents, rels = O.first_seen_index(train)
    ti, tri = O.to_indexes(test, ents, rels), O.to_indexes(train, ents, rels)
    fs, fo = O.filter_sets(ti, [tri, ti])
    R = len(rels)
    ref = O.evaluate_ranks(model, st.ent, st.rel, ti, fs, fo, "s,o", "worst", max_rel_size=R)
    rng = np.random.default_rng(1)
    untrained = O.evaluate_ranks(model, O.glorot_uniform(len(ents), st.ent.shape[1], rng), O.glorot_uniform(R, st.rel.shape[1], rng),
                                 ti, fs, fo, "s,o", "worst", max_rel_size=R)
    mrr_g, mrr_o, mrr_0 = O.mrr_score(ranks), O.mrr_score(ref), O.mrr_score(untrained)
    gold = _golden()
    if f"{model}/{loss}" in gold:   # the many-seed test below compares with these frozen oracle values: same replay, same bits
        assert abs(gold[f"{model}/{loss}"][seed, 0] - mrr_o) <= 1e-12 and abs(gold[f"{model}/{loss}"][seed, 3] - hist[-1]) <= 1e-9 * abs(hist[-1])
    report = dict(loss_drift=drift, first_epochs_drift=float(np.max(np.abs(got[:5] - hist[:5]) / np.abs(hist[:5]))),
                  mrr_gpu=mrr_g, mrr_oracle=mrr_o, mrr_untrained=mrr_0, loss_first=float(got[0]), loss_last=float(got[-1]))
    print("learning parity", model, loss, seed, report)
    assert drift <= loss_tol and report["first_epochs_drift"] <= early_tol, report
    assert got[-1] < (0.6 if mrr_tol is not None else 0.75) * got[0], report   # the loss really goes down (RotatE / nll: by a third in 40 epochs)
    assert mrr_o > 3 * mrr_0 and mrr_o > mrr_min, report            # learnable structure: MRR rises well above chance
    if mrr_tol is None:   # chaotic regime (see CASES): both runs learn, no distance asserted
        assert mrr_g > 3 * mrr_0 and mrr_g > mrr_min, report
        return
    assert abs(mrr_g - mrr_o) <= mrr_tol, report
    assert abs(O.hits_at_n_score(ranks, 10) - O.hits_at_n_score(ref, 10)) <= max(1e-2, 4 * mrr_tol), report


def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "learning_mrr_v1.npz"))


# VERDICT r3 #1a.  The distance models' trajectories are sensitive to rounding (above), so ONE seed's filtered MRR cannot be held
# to the north_star's +-0.002 by any implementation that is not bit-identical to the one it is compared with: the oracle
# replaying the schedule from tables nudged by one ulp parts from ITSELF by sd 0.0038 (TransE / nll), 0.0009 (TransE / pairwise),
# 0.0024 (RotatE / self_adversarial), 0.0127 (RotatE / nll) per seed.  What the bar can mean for them is that the GPU path has no
# BIAS: the MEAN filtered MRR over many seeds within +-0.002 of the oracle's mean over the same seeds -- with enough seeds that
# the standard error of the mean distance is under a third of the bar (tests/golden/make_learning_golden.py: 512 / 512 / 512 / 2048 seeds).  The
# oracle's side is frozen in tests/golden/learning_mrr_v1.npz (re-derived on the CPU by tests/test_golden.py, and by the
# per-seed test above for seeds 0..2).
@pytest.mark.parametrize("model,loss", [("TransE", "nll"), ("TransE", "pairwise"), ("RotatE", "self_adversarial"), ("RotatE", "nll")])
def test_mean_mrr_over_seeds_matches_oracle(gpu_lib, model, loss):
    from planted import planted_kg

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers

    G = _golden()
    gold = G[f"{model}/{loss}"]   # columns: oracle MRR, hits@10, first-epoch loss, last-epoch loss

    histories = {}
    # Round 6: the one-in-25 000 excursion of round 5 was a NaN, and its birth is named (scripts/nan_hunt.py,
    # profiles/r06a_nan_hunt_*: 4 of 19 000 RotatE fits).  RotatE's modulus has no epsilon (RotatE.py:102-104): a corruption (e, p, e)
    # -- the sampler redrew the entity that is already on the other side, one corruption in N -- whose relation has a unit with a phase
    # so close to 0 that cos rounds to 1 and e sin(phi) falls under the ulp of e evaluates to z = (0, 0) EXACTLY in fp32, sqrt(0) = 0,
    # and the gradient is 0 / 0: NaN in the tables from that step on -- in the reference's own arithmetic as here (tf.sqrt's gradient at
    # 0 is inf, times 2 z = 0).  The loss kernels now report it as the NaN it is (clip_exp, kge_train_kernel.h) instead of 375 per
    # positive.  The oracle replays in fp64 and never meets the exact zero, so a fit that does is re-drawn here (the default mode's
    # arrival-order rounding differs from run to run: the same seed does not meet it twice) and COUNTED: more than two per case would
    # be another mechanism.
    nan_fits = []

    def fit_one(seed, retries=2):
        d = planted_kg(model, seed=seed)
        train, test = d["train"].astype(str), d["test"].astype(str)
        m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type=model, seed=seed)
        m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=loss)
        h = m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"]
        if not np.isfinite(h).all():
            nan_fits.append(dict(seed=int(seed), first_nonfinite_epoch=int(np.nonzero(~np.isfinite(h))[0][0])))
            assert model == "RotatE", ("a non-finite loss outside RotatE's modulus-zero case", nan_fits)
            assert retries > 0, ("the same seed met the modulus-zero case three times", nan_fits)
            return fit_one(seed, retries - 1)
        e_, r_ = m._engine.get_tables()
        assert np.isfinite(e_).all() and np.isfinite(r_).all(), ("non-finite tables behind a finite loss history", seed)
        ranks = m.evaluate(test, use_filter={"train": train, "test": test}, corrupt_side="s,o", verbose=False)
        return (O.mrr_score(ranks), O.hits_at_n_score(ranks, 10), h[0], h[-1]), h

    def fit_seeds(first, count):
        out = np.zeros((count, 4))
        for seed in range(first, first + count):
            out[seed - first], histories[seed] = fit_one(seed)
        return out

    got = fit_seeds(0, len(gold))
    # Sequential second stage (tests/golden/make_learning_golden.py EXTENSIONS): the default mode is not bitwise reproducible, so
    # the mean distance is a random variable; where the bar is only ~3 of its standard errors away (RotatE / nll) a first-stage
    # mean beyond 0.0012 is settled on twice the seeds instead of being left to chance.
    ext = f"{model}/{loss}/ext"
    if ext in G.files and abs(float((got[:, 0] - gold[:, 0]).mean())) > 1.2e-3:
        print("mean MRR over seeds", model, loss, "first stage", float((got[:, 0] - gold[:, 0]).mean()), "-> second stage")
        got = np.concatenate([got, fit_seeds(len(gold), len(G[ext]))])
        gold = np.concatenate([gold, G[ext]])
    n = len(gold)
    # One fit in tens of thousands has ended far off (round 5, one closing run of nine: a last-epoch loss ~50x the oracle's on ONE
    # seed of 2 048; every oracle and every other GPU value of that column lies within 1.6 % of 6 400 -- not reproduced in 8 800
    # further fits, profiles/r05j_*, r05k_*: DESIGN.md section 8 "open").  Such a fit fails this test through the mean below; what the
    # failure message then needs is WHICH seed, its loss history and whether a second fit of the same seed repeats it.
    rel_last = (got[:, 3] - gold[:, 3]) / np.abs(gold[:, 3])
    excursions = []
    for s_ in np.nonzero(np.abs(rel_last) > 0.05)[0][:4]:
        again, h2 = fit_one(int(s_))
        excursions.append(dict(seed=int(s_), last_epoch_loss=float(got[s_, 3]), oracle=float(gold[s_, 3]), mrr=float(got[s_, 0]), mrr_oracle=float(gold[s_, 0]),
                               loss_history_every_4th_epoch=[float(x) for x in histories[int(s_)][::4]],
                               second_fit_last_epoch_loss=float(again[3]), second_fit_history_every_4th_epoch=[float(x) for x in h2[::4]]))
    dm = got[:, 0] - gold[:, 0]
    report = dict(seeds=n, mrr_gpu_mean=float(got[:, 0].mean()), mrr_oracle_mean=float(gold[:, 0].mean()), mean_distance=float(dm.mean()),
                  per_seed_distance_sd=float(dm.std()), per_seed_distance_max=float(np.abs(dm).max()),
                  standard_error=float(dm.std() / np.sqrt(n)), hits10_mean_distance=float((got[:, 1] - gold[:, 1]).mean()),
                  first_epoch_loss_max_rel=float(np.max(np.abs(got[:, 2] - gold[:, 2]) / np.abs(gold[:, 2]))),
                  last_epoch_loss_mean_rel=float(np.mean(rel_last)), last_epoch_loss_max_rel=float(np.max(np.abs(rel_last))),
                  fits_beyond_5_percent_in_the_last_epoch_loss=excursions, fits_redrawn_after_a_modulus_zero_nan=nan_fits)
    print("mean MRR over seeds", model, loss, report)
    assert report["first_epoch_loss_max_rel"] <= {"nll": 5e-5, "pairwise": 4e-4, "self_adversarial": 1e-5}[loss], report   # before any drift: every seed (measured max over 512 / 2 048 seeds: 1.3e-5 / 2.2e-5 / 1.9e-7)
    assert len(nan_fits) <= 2, report
    assert abs(report["mean_distance"]) <= 2e-3, report                       # the north_star's bar, on the mean
    assert abs(report["hits10_mean_distance"]) <= 4e-3, report
    # no bias in the loss either (per seed the pairwise hinge parts by 0.5 .. 0.8 %, the others by <= 0.3 %: CASES above)
    assert abs(report["last_epoch_loss_mean_rel"]) <= (4e-3 if loss == "pairwise" else 1e-3), report


@pytest.mark.parametrize("loss", ["nll", "pairwise"])
def test_transe_drift_is_sensitivity_to_one_ulp(gpu_lib, loss):
    """VERDICT r2 (weak 3): is TransE's distance from the oracle replay (4.7e-4 .. 6.8e-4 in the nll loss, 0.5 .. 0.8 % in the
    pairwise loss after 160 Adam steps) a defect of these kernels, or the model's own sensitivity to rounding?  The GPU against
    ITSELF: the same schedule from initial tables that differ by one unit in the last place (a relative 2^-23, the size of an
    fp32-vs-fp64 rounding difference), and the same tables with the two orders of the fp32 additions into a gradient row
    (arrival order / the deterministic mode's sorted order).  Measured on MI355X, nll: the one-ulp nudge parts the loss
    histories by 5.7e-4 .. 9.0e-4 and the two orders by 6.9e-4 .. 7.4e-4 -- the size of the distance to the oracle -- with the first five epochs
    within 1e-8; the same nudge on a smooth model (ComplEx) moves its history by 1.3e-7.  sign(s + p - o) flips wherever a
    unit sits within rounding noise of 0, and which units do is trajectory dependent (the pairwise run from THESE tables happens
    not to amplify within 160 steps: 2e-8): the bars are asserted, the amplification is reported."""
    from planted import planted_kg

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers

    def run(model, the_loss, nudge, deterministic=False):
        d = planted_kg(model, seed=0)
        train = d["train"].astype(str)
        ne, nr = len(np.unique(train[:, [0, 2]])), len(np.unique(train[:, 1]))
        kk = K if model == "TransE" else 2 * K
        rng = np.random.default_rng(5)
        E0 = rng.uniform(-1, 1, size=(ne, kk)).astype(np.float32) * np.float32(np.sqrt(6.0 / (ne + kk)))
        R0 = rng.uniform(-1, 1, size=(nr, kk)).astype(np.float32) * np.float32(np.sqrt(6.0 / (nr + kk)))
        if nudge:   # one ulp up or down, element by element
            E0 = np.nextafter(E0, np.where(rng.random(E0.shape) < 0.5, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
        m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type=model, seed=0)
        m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=the_loss, entity_relation_initializer=[E0, R0],
                  deterministic=deterministic)
        return np.asarray(m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"])

    def dist(x, y):
        return float(np.max(np.abs(x - y) / np.abs(y)))

    a, b = run("TransE", loss, False), run("TransE", loss, True)
    c, e = run("ComplEx", "multiclass_nll", False), run("ComplEx", "multiclass_nll", True)
    o = run("TransE", loss, False, deterministic=True)
    print("TransE", loss, "one-ulp nudge of the initial entity table: loss histories part by", dist(a, b), "(first 5 epochs",
          dist(a[:5], b[:5]), "); ComplEx:", dist(c, e), "; TransE, two summation orders:", dist(a, o))
    bar = {"nll": 6e-3, "pairwise": 6e-2}[loss]                 # 3 x the bars of CASES (GPU vs oracle): the default mode's own run-to-run
                                                                # spread is part of what is measured here (5.7e-4 .. 9.0e-4 over three runs)
    assert dist(a[:5], b[:5]) <= {"nll": 5e-5, "pairwise": 2e-3}[loss] and dist(a[:5], o[:5]) <= {"nll": 5e-5, "pairwise": 2e-3}[loss]
    assert dist(a, b) <= bar and dist(a, o) <= bar              # never further apart than the GPU is from the oracle
    assert dist(c, e) <= 1e-5                                   # a smooth model does not amplify the nudge
    if loss == "nll" and not max(dist(a, b), dist(a, o)) > 100 * dist(c, e):
        # measured three times out of three (see above); which units flip is trajectory dependent and the default mode is not
        # bitwise reproducible, so a run that happens not to amplify is reported, not failed
        import warnings

        warnings.warn("TransE nll: this run did not amplify the one-ulp nudge (%.3g, %.3g)" % (dist(a, b), dist(a, o)))


# VERDICT r3 #1b: the rigorous half.  oracle/train_ordered.py restates the TransE / pairwise / Adam step in the kernels' DECLARED
# fp32 order (the unit order of a lane, the wave64 tree of a score, the fp32 hinge, opt_elem's Adam).  With the pairwise loss every
# gradient entry is an integer, so the row sums do not depend on the order of their additions and BOTH train paths (the default
# one with its arrival-order buckets / atomics, and the deterministic mode) must reproduce the ordered replay BIT FOR BIT: same
# hinge terms active, same signs flipped, same tables after 160 Adam steps -- hence the same ranks and the same MRR, per seed.
@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_transe_pairwise_fit_equals_ordered_oracle_bit_for_bit(gpu_lib, seed, deterministic):
    from planted import LEARNING, planted_kg

    from oracle import rank_ordered as RO
    from oracle import train_ordered as TO

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers
    from ampligraph_amd.latent_features.initializers import initialise

    d = planted_kg("TransE", seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)
    m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type="TransE", seed=seed)
    m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss="pairwise", deterministic=deterministic)
    got = np.asarray(m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"])
    # (k = 16: the default StepLoop takes the atomic-scatter kernels, one unit per lane; deterministic mode the owner-computes pair,
    # one quad per lane -- two declared score trees, oracle/train_ordered._lane_sums)
    hist, st, Xi, ti = TO.replay_learning("TransE", "pairwise", seed, LEARNING, planted_kg, initialise, layout="quad" if deterministic else "unit")
    ents, rels = O.first_seen_index(train)
    names_e = np.array(sorted(ents, key=ents.get))
    names_r = np.array(sorted(rels, key=rels.get))
    E = m.get_embeddings(names_e, embedding_type="e")
    Rm = m.get_embeddings(names_r, embedding_type="r")
    diff_e, diff_r = int((E != st.ent).sum()), int((Rm != st.rel).sum())
    report = dict(seed=seed, deterministic=deterministic, loss_history_max_rel=float(np.max(np.abs(got - hist) / np.abs(hist))),
                  entity_elements_differing=diff_e, relation_elements_differing=diff_r,
                  max_abs_diff=float(max(np.abs(E - st.ent).max(), np.abs(Rm - st.rel).max())))
    print("TransE pairwise vs ordered oracle", report)
    assert diff_e == 0 and diff_r == 0, report                       # the tables after 160 Adam steps: the same bits
    # the loss history: the same fp32 per-positive losses summed in fp64 (measured 0.0 in deterministic mode; restated with the wrong
    # lane layout the default path's history sat 1.2e-8 off with the tables still bit-identical -- scores differing in their last bit
    # rarely flip a hinge term)
    assert within("learning/det_fit_vs_ordered_oracle/loss_history", report["loss_history_max_rel"], 1e-12), report
    ranks = m.evaluate(test, use_filter={"train": train, "test": test}, corrupt_side="s,o", verbose=False)
    fs, fo = O.filter_sets(ti, [Xi, ti])
    ref = RO.evaluate_ranks("TransE", st.ent, st.rel, ti, fs, fo, corrupt_side="s,o", ranking_strategy="worst")
    assert np.array_equal(ranks, ref) and O.mrr_score(ranks) == O.mrr_score(ref)


RULES = [("sgd", {}, "sgd", (None, None)), ("adagrad", {}, "adagrad", (None, None)),
         ("sgd", {"momentum": 0.7, "nesterov": True}, "momentum", (0.7, 1.0)), ("sgd", {"momentum": 0.9}, "momentum", (0.9, 0.0)),
         ("rmsprop", {}, "rmsprop", (0.9, 0.0)), ("rmsprop", {"momentum": 0.5}, "rmsprop_mom", (0.9, 0.5)),
         ("adadelta", {}, "adadelta", (0.95, 0.0)), ("adamax", {}, "adamax", (0.9, 0.999))]


@pytest.mark.parametrize("loss", ["pairwise", "absolute_margin"])
@pytest.mark.parametrize("name,hp,kind,desc", RULES)
def test_transe_integer_gradient_fits_are_bitwise_for_every_update_rule(gpu_lib, name, hp, kind, desc, loss):
    """The same statement as above for the other seven update rules of kge_opt.h and for the second loss whose gradient is
    integer-valued (absolute_margin, loss_functions.py:458-464): fit() on the GPU == oracle/train_ordered.py's restatement of
    opt_elem<KIND>, bit for bit, after 10 epochs (40 steps) -- the optimizer ARITHMETIC of the kernels is pinned to a CPU
    restatement operation by operation (what it restates, the Keras-legacy rules, is checked against kge_oracle on the CPU)."""
    from planted import LEARNING, planted_kg

    from oracle import train_ordered as TO

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers
    from ampligraph_amd.latent_features.initializers import initialise

    seed, epochs = 5, 10
    lr = {"sgd": 1e-3, "adadelta": 1.0}.get(name, LR)   # (integer gradients of size ~B: plain SGD needs a small step)
    d = planted_kg("TransE", seed=seed)
    train = d["train"].astype(str)
    m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type="TransE", seed=seed)
    m.compile(optimizer=optimizers.get(name, dict(hp, learning_rate=lr)), loss=loss)
    got = np.asarray(m.fit(train, batch_size=BATCH, epochs=epochs, verbose=False).history["loss"])
    hist, st, Xi, ti = TO.replay_learning("TransE", loss, seed, dict(LEARNING, lr=lr), planted_kg, initialise, epochs=epochs, opt=kind, opt_hp=desc, layout="unit")
    ents, rels = O.first_seen_index(train)
    E = m.get_embeddings(np.array(sorted(ents, key=ents.get)), embedding_type="e")
    Rm = m.get_embeddings(np.array(sorted(rels, key=rels.get)), embedding_type="r")
    report = dict(rule=kind, loss=loss, entity_elements_differing=int((E != st.ent).sum()), relation_elements_differing=int((Rm != st.rel).sum()),
                  max_abs_diff=float(max(np.abs(E - st.ent).max(), np.abs(Rm - st.rel).max())),
                  loss_history_max_rel=float(np.max(np.abs(got - hist) / np.abs(hist))))
    print("TransE integer-gradient fit vs ordered oracle", report)
    assert report["entity_elements_differing"] == 0 and report["relation_elements_differing"] == 0, report
    assert within("learning/det_fit_vs_ordered_oracle/loss_history", report["loss_history_max_rel"], 1e-12), report


@pytest.mark.parametrize("seed,loss", [(0, "nll"), (1, "nll"), (2, "nll"), (3, "nll"), (0, "self_adversarial"), (1, "self_adversarial"),
                                       (0, "multiclass_nll"), (1, "multiclass_nll")])
def test_transe_nll_deterministic_fit_equals_ordered_oracle_bit_for_bit(gpu_lib, seed, loss):
    """TransE with the losses that have transcendentals (nll, self_adversarial -- the online softmax of the single-pass kernel --,
    multiclass_nll) -- real-valued gradient coefficients, so every fp32 addition's place matters -- in DETERMINISTIC mode against
    oracle/train_ordered.transe_step_det: the loss terms from the mode's declared transcendentals (IEEE operations only, the same
    in numpy), the forward kernel's per-side sums in corruption order, the tile pass adding each row's entries sorted by (positive,
    role, bits of g), the relation gradient in batch order, opt_elem's Adam.  160 Adam steps: both tables bit-identical, hence the
    same filtered ranks and the same MRR -- the north_star's +-0.002 holds per seed with 0.002 to spare."""
    from planted import LEARNING, planted_kg

    from oracle import rank_ordered as RO
    from oracle import train_ordered as TO

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers
    from ampligraph_amd.latent_features.initializers import initialise

    d = planted_kg("TransE", seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)
    m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type="TransE", seed=seed)
    m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=loss, deterministic=True)
    got = np.asarray(m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"])
    hist, st, Xi, ti = TO.replay_learning("TransE", loss, seed, LEARNING, planted_kg, initialise)
    ents, rels = O.first_seen_index(train)
    E = m.get_embeddings(np.array(sorted(ents, key=ents.get)), embedding_type="e")
    Rm = m.get_embeddings(np.array(sorted(rels, key=rels.get)), embedding_type="r")
    report = dict(seed=seed, loss=loss, entity_elements_differing=int((E != st.ent).sum()), relation_elements_differing=int((Rm != st.rel).sum()),
                  max_abs_diff=float(max(np.abs(E - st.ent).max(), np.abs(Rm - st.rel).max())),
                  loss_history_max_rel=float(np.max(np.abs(got - hist) / np.abs(hist))),
                  first_epoch_loss_rel=float(abs(got[0] - hist[0]) / abs(hist[0])))
    print("TransE transcendental loss (deterministic) vs ordered oracle", report)
    assert report["entity_elements_differing"] == 0 and report["relation_elements_differing"] == 0, report
    assert report["loss_history_max_rel"] <= 1e-6, report   # (multiclass: the per-positive log of the loss VALUE is libm's on either side)
    ranks = m.evaluate(test, use_filter={"train": train, "test": test}, corrupt_side="s,o", verbose=False)
    fs, fo = O.filter_sets(ti, [Xi, ti])
    ref = RO.evaluate_ranks("TransE", st.ent, st.rel, ti, fs, fo, corrupt_side="s,o", ranking_strategy="worst")
    assert np.array_equal(ranks, ref) and O.mrr_score(ranks) == O.mrr_score(ref)


DET_CASES = [(0, "ComplEx", "self_adversarial"), (1, "ComplEx", "self_adversarial"), (0, "ComplEx", "nll"), (1, "ComplEx", "multiclass_nll"),
                 (0, "DistMult", "self_adversarial"), (1, "DistMult", "nll"), (0, "HolE", "self_adversarial"), (1, "HolE", "multiclass_nll"),
                 (0, "RotatE", "self_adversarial"), (1, "RotatE", "self_adversarial"), (0, "RotatE", "nll"), (1, "RotatE", "multiclass_nll")]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,model,loss", DET_CASES)
def test_deterministic_fit_equals_ordered_oracle_bit_for_bit(gpu_lib, seed, model, loss):
    """DETERMINISTIC mode against oracle/train_ordered.trilinear_step_det / rotate_step_det: ComplEx (the model of BASELINE
    configs[1], with its self-adversarial loss), DistMult, HolE, and RotatE (configs[4]'s model: correctly rounded cos / sin of the
    fp32 phase, IEEE sqrt and division, unit vectors summed per side in groups of three).  Side rows A = d/do(s, p), B = d/ds(p, o) from
    grad_unit's products; a corruption's score one fmaf chain per lane over (unit, component) + the wave tree; sum_j c_j e_j per
    side by fmaf in corruption order with the online-softmax rescale; the row gradients from grad_unit on the sums; the tile
    pass adding fl(g A) / fl(g B) per entry in sorted order; the relation gradient in batch order; opt_elem's Adam.  160 Adam
    steps: both tables bit-identical to the numpy replay, hence identical filtered ranks and MRR."""
    from planted import LEARNING, planted_kg

    from oracle import rank_ordered as RO
    from oracle import train_ordered as TO

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers
    from ampligraph_amd.latent_features.initializers import initialise

    d = planted_kg(model, seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)
    m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type=model, seed=seed)
    m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=loss, deterministic=True)
    got = np.asarray(m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"])
    hist, st, Xi, ti = TO.replay_learning(model, loss, seed, LEARNING, planted_kg, initialise)
    ents, rels = O.first_seen_index(train)
    E = m.get_embeddings(np.array(sorted(ents, key=ents.get)), embedding_type="e")
    Rm = m.get_embeddings(np.array(sorted(rels, key=rels.get)), embedding_type="r")
    report = dict(seed=seed, model=model, loss=loss, entity_elements_differing=int((E != st.ent).sum()),
                  relation_elements_differing=int((Rm != st.rel).sum()),
                  max_abs_diff=float(max(np.abs(E - st.ent).max(), np.abs(Rm - st.rel).max())),
                  loss_history_max_rel=float(np.max(np.abs(got - hist) / np.abs(hist))),
                  first_epoch_loss_rel=float(abs(got[0] - hist[0]) / abs(hist[0])))
    print("deterministic fit vs ordered oracle", report)
    assert report["entity_elements_differing"] == 0 and report["relation_elements_differing"] == 0, report
    # the loss VALUE: identical but for multiclass_nll, whose per-positive log(Z) is libm's logf on the device and numpy's log
    # here (measured 1.4e-6 / 1.9e-6 of the epoch means, profiles/r04l_pytest_trilinear_det.log); nothing of it feeds back
    assert report["loss_history_max_rel"] <= (5e-6 if loss == "multiclass_nll" else 0.0), report
    ranks = m.evaluate(test, use_filter={"train": train, "test": test}, corrupt_side="s,o", verbose=False)
    fs, fo = O.filter_sets(ti, [Xi, ti])
    ref = RO.evaluate_ranks(model, st.ent, st.rel, ti, fs, fo, corrupt_side="s,o", ranking_strategy="worst", max_rel_size=len(rels))
    assert np.array_equal(ranks, ref) and O.mrr_score(ranks) == O.mrr_score(ref)
