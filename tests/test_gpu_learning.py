"""End-to-end LEARNING parity: fit() + evaluate() of the drop-in class on the GPU against the oracle replaying the same
schedule, on a planted-structure graph where filtered MRR really rises (tests/planted.py) -- loss history within 1e-4,
filtered MRR within +-0.002 (BASELINE.json north_star), several seeds, a contraction model and a distance model."""
import numpy as np
import pytest

from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu

EPOCHS, BATCH, ETA, K, LR = 40, 1024, 5, 16, 2e-2


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("model,loss", [("ComplEx", "multiclass_nll"), ("TransE", "pairwise")])
def test_fit_learns_and_matches_oracle_replay(gpu_lib, model, loss, seed):
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
    assert np.allclose(got, hist, rtol=1e-4), float(np.max(np.abs(got - hist) / np.abs(hist)))
    assert got[-1] < 0.5 * got[0]                                   # the loss really goes down
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
    assert mrr_o > 5 * mrr_0 and mrr_o > 0.15, (mrr_o, mrr_0)       # learnable structure: MRR rises well above chance
    assert abs(mrr_g - mrr_o) <= 2e-3, (mrr_g, mrr_o)
    assert abs(O.hits_at_n_score(ranks, 10) - O.hits_at_n_score(ref, 10)) <= 1e-2
