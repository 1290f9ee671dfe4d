"""Benchmark-shaped synthetic knowledge graphs (no network: the reference's dataset loaders are
figshare downloads, /root/reference/ampligraph/datasets/datasets.py:689,707).  Shapes follow
/root/reference/docs/ampligraph.datasets.rst:49-62 and ScoringBasedEmbeddingModel.py:955-958,1603."""
import numpy as np

SYNTH_SHAPES = {
    # name: (n_entities, n_relations, n_train, n_valid, n_test)
    "synth-fb15k237": (14505, 237, 272115, 17526, 20438),
    "synth-wn18rr": (40943, 11, 86835, 3034, 2924),
    "synth-yago310": (123182, 37, 1079040, 5000, 5000),
    # one GPU's share of BASELINE configs[4] (50 M entities / 8 GPUs; triples cut down, the table size is what matters)
    "synth-c5-shard": (6_250_000, 1000, 4_000_000, 1000, 1000),
    "synth-c5-small": (1_000_000, 1000, 2_000_000, 1000, 1000),
}


def make_synthetic_kg(name="synth-fb15k237", seed=0, popularity="uniform"):
    """Unique int32 triples, split train/valid/test; every entity and relation appears in train.

    popularity: "uniform" (cache-unfriendly primary) or "zipf" (entity/relation popularity ~ 1/rank)."""
    N, R, n_tr, n_va, n_te = SYNTH_SHAPES[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    total = n_tr + n_va + n_te
    if popularity == "zipf":
        pe = 1.0 / np.arange(1, N + 1)
        pe /= pe.sum()
        pr = 1.0 / np.arange(1, R + 1)
        pr /= pr.sum()
    else:
        pe = pr = None
    keys = np.zeros(0, dtype=np.int64)
    while keys.size < total:
        m = int((total - keys.size) * 1.3) + 1024
        s = rng.choice(N, size=m, p=pe).astype(np.int64)
        p = rng.choice(R, size=m, p=pr).astype(np.int64)
        o = rng.choice(N, size=m, p=pe).astype(np.int64)
        keys = np.unique(np.concatenate([keys, (s * R + p) * N + o]))
    keys = rng.permutation(keys)[:total]
    X = np.stack([keys // (R * N), (keys // N) % R, keys % N], 1).astype(np.int32)
    train, valid, test = X[:n_tr].copy(), X[n_tr:n_tr + n_va], X[n_tr + n_va:]
    # force every entity / relation into train (overwrite the subject / relation of the first rows); graphs with more
    # entities than 2 * triples cannot cover them all (synth-c5-*): as many as fit
    missing_e = np.setdiff1d(np.arange(N, dtype=np.int32), np.union1d(train[:, 0], train[:, 2]))[:max(0, n_tr - R)]
    train[:missing_e.size, 0] = missing_e
    missing_r = np.setdiff1d(np.arange(R, dtype=np.int32), train[:, 1])
    train[missing_e.size:missing_e.size + missing_r.size, 1] = missing_r
    return {"train": train, "valid": valid.copy(), "test": test.copy(), "n_ents": N, "n_rels": R}
