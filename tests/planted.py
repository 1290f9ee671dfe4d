"""Planted-structure synthetic knowledge graph (TEST INFRASTRUCTURE): triples that a KGE model can actually learn.

Ground-truth embeddings of a small rank are drawn, every (subject, relation) pair links to the objects the ground-truth
model scores highest, so held-out triples are predictable from the training ones -- unlike the uniform-random graphs
bench.py times, where filtered MRR is noise by construction."""
import numpy as np


def planted_kg(model="ComplEx", n_ents=300, n_rels=6, k_true=6, per_pair=3, n_test=600, seed=0):
    """-> dict(train, test (int32 id triples, every test entity / relation also in train), n_ents, n_rels)."""
    rng = np.random.default_rng(seed)
    if model == "TransE":
        E = rng.normal(size=(n_ents, k_true))
        Rm = rng.normal(size=(n_rels, k_true)) * 1.5
        score = lambda s, r: -np.abs(E[s][:, None, :] + Rm[r][None, None, :] - E[None, :, :]).sum(-1)   # noqa: E731
    else:   # bilinear-diagonal ground truth with complex phases (what ComplEx / DistMult / HolE can represent)
        E = rng.normal(size=(n_ents, k_true)) + 1j * rng.normal(size=(n_ents, k_true))
        Rm = np.exp(1j * rng.uniform(0, 2 * np.pi, size=(n_rels, k_true)))
        score = lambda s, r: np.real((E[s] * Rm[r][None, :])[:, None, :] * np.conj(E)[None, :, :]).sum(-1)   # noqa: E731
    tri = []
    all_s = np.arange(n_ents)
    for r in range(n_rels):
        sc = score(all_s, r)                       # (n_ents, n_ents)
        sc[all_s, all_s] = -np.inf                 # no self loops
        top = np.argsort(-sc, axis=1)[:, :per_pair]
        for j in range(per_pair):
            tri.append(np.stack([all_s, np.full(n_ents, r), top[:, j]], 1))
    tri = np.concatenate(tri).astype(np.int32)
    tri = tri[rng.permutation(len(tri))]
    # hold out triples whose entities stay covered by the rest
    cnt = np.bincount(np.concatenate([tri[:, 0], tri[:, 2]]), minlength=n_ents)
    test_idx = []
    for i, (s, _, o) in enumerate(tri):
        if len(test_idx) == n_test:
            break
        if cnt[s] > 2 and cnt[o] > 2:
            cnt[s] -= 1
            cnt[o] -= 1
            test_idx.append(i)
    mask = np.zeros(len(tri), dtype=bool)
    mask[test_idx] = True
    return {"train": tri[~mask], "test": tri[mask], "n_ents": n_ents, "n_rels": n_rels}


# ---- the learning-parity schedule of tests/test_gpu_learning.py, replayed by the oracle (CPU) ---------------------------------
LEARNING = dict(epochs=40, batch=1024, eta=5, k=16, lr=2e-2)


def oracle_learning_run(model, loss, seed, epochs=None):
    """The oracle's replay of test_gpu_learning's schedule on planted_kg(model, seed): Glorot tables drawn as the drop-in
    class draws them, Adam, `epochs` passes in sequential batches -> (loss history, filtered ranks (n_test, 2), state)."""
    from oracle import kge_oracle as O

    from ampligraph_amd.latent_features.initializers import initialise

    cfg = LEARNING
    d = planted_kg(model, seed=seed)
    train, test = d["train"].astype(str), d["test"].astype(str)
    ents, rels = O.first_seen_index(train)
    Xi = O.to_indexes(train, ents, rels)
    N, R, K = len(ents), len(rels), O.internal_k(model, cfg["k"])
    rng = np.random.Generator(np.random.PCG64(seed))
    st = O.TrainState(initialise("glorot_uniform", (N, K), rng), initialise("glorot_uniform", (R, K), rng), "adam", cfg["lr"])
    steps = (len(Xi) + cfg["batch"] - 1) // cfg["batch"]
    hist = []
    for ep in range(cfg["epochs"] if epochs is None else epochs):
        tot = 0.0
        for s in range(steps):
            tot += float(O.train_step(st, model, Xi[s * cfg["batch"]:(s + 1) * cfg["batch"]], cfg["eta"], loss, seed,
                                      ep * steps + s, max_rel_size=R))
        hist.append(tot / steps)
    ti = O.to_indexes(test, ents, rels)
    fs, fo = O.filter_sets(ti, [Xi, ti])
    ranks = O.evaluate_ranks(model, st.ent, st.rel, ti, fs, fo, "s,o", "worst", max_rel_size=R)
    return np.asarray(hist), ranks, st
