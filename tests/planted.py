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
