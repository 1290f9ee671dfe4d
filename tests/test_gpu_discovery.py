"""Discovery on the device: the selection / score / neighbour kernels through the C ABI against numpy, and
query_topn / find_nearest_neighbours of the drop-in surface against brute force -- single GPU and row-sharded."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from test_gpu_kernels import dev, make_engine, rand_triples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,k", [(3, 10, 4), (5, 14505, 10), (2, 100000, 1024), (4, 7, 7), (1, 3, 5), (6, 5000, 100),
                                   (2, 5000, 3000), (2, 1500, 2000)])   # k > 1024: the full-sort path (any top_n, like the reference)
@pytest.mark.parametrize("largest", [True, False])
def test_topk_rows_against_numpy(gpu_lib, n, m, k, largest):
    eng, _, _ = make_engine("DistMult", 4, 8, 3)
    rng = np.random.default_rng(n * 1000 + k)
    V = rng.normal(size=(n, m)).astype(np.float32)
    V[:, : m // 3] = np.round(V[:, : m // 3], 1)                   # many exact ties
    if m > 5:
        V[0, 2] = np.nan
    scale = rng.uniform(0.5, 2.0, m).astype(np.float32)
    bias = rng.normal(size=m).astype(np.float32)
    for cs, cb in ((None, None), (scale, bias)):
        idx, val = eng.topk_rows(dev(V), k, largest, None if cs is None else dev(cs), None if cb is None else dev(cb))
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        W = V.copy()
        if cs is not None:
            W = (W * cs).astype(np.float32)
            W = (W + cb).astype(np.float32)
        for i in range(n):
            w = W[i].astype(np.float64)
            w[np.isnan(w)] = -np.inf if largest else np.inf
            order = np.lexsort((np.arange(m), -w if largest else w))   # best first, ties by increasing column
            kk = min(k, m)
            assert np.array_equal(idx[i, :kk], order[:kk]), (i, idx[i, :8], order[:8])
            got = val[i, :kk]
            ok = ~np.isnan(W[i][order[:kk]])
            assert np.array_equal(got[ok], W[i][order[:kk]][ok])
            assert (idx[i, kk:] == -1).all()


@pytest.mark.parametrize("model,k", [("TransE", 50), ("DistMult", 37), ("ComplEx", 200), ("HolE", 10), ("RotatE", 33)])
def test_corruption_scores_against_oracle(gpu_lib, model, k):
    from ampligraph_amd import _ffi

    N, R, n = 700, 5, 37
    eng, ent, rel = make_engine(model, k, N, R, scale=0.3)
    rng = np.random.default_rng(2)
    X = rand_triples(rng, n, N, R)
    sub = np.sort(rng.choice(N, 200, replace=False)).astype(np.int32)
    s, p, o = O.lookup(ent, rel, X)
    for side, nm in ((_ffi.SIDE_S, "s"), (_ffi.SIDE_O, "o")):
        for ids in (None, sub):
            E = ent if ids is None else ent[ids]
            ref = O.corruption_scores(model, nm, s, p, o, E, R)
            kk = 20
            pos, val = eng.corruption_topk(dev(X), side, kk, ent_ids=None if ids is None else dev(ids))
            pos, val = pos.cpu().numpy(), val.cpu().numpy()
            tol = 2e-5 * np.abs(ref).max()
            for i in range(n):
                assert np.allclose(val[i], ref[i, pos[i]], rtol=2e-5, atol=tol)           # the reported scores are the scores of the reported rows
                assert val[i, -1] >= np.sort(ref[i])[::-1][kk - 1] - 2 * tol               # ... and they are the best ones
                assert (np.diff(val[i]) <= 0).all()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_nearest_rows_against_numpy(gpu_lib, metric):
    eng, ent, rel = make_engine("ComplEx", 50, 3000, 3, scale=0.5)
    rng = np.random.default_rng(4)
    qid = rng.choice(3000, 40, replace=False)
    sub = np.sort(rng.choice(3000, 900, replace=False)).astype(np.int32)
    for ids in (None, sub):
        pos, got = eng.nearest_rows(eng.ent[dev(qid.astype(np.int64))], 8, metric, ent_ids=None if ids is None else dev(ids))
        pos, got = pos.cpu().numpy(), got.cpu().numpy().astype(np.float64)
        E = (ent if ids is None else ent[ids]).astype(np.float64)
        Q = ent[qid].astype(np.float64)
        if metric == "cosine":
            D = 1.0 - (Q @ E.T) / np.linalg.norm(Q, axis=1)[:, None] / np.linalg.norm(E, axis=1)[None, :]
        else:
            D = np.sqrt(np.maximum(((Q[:, None, :] - E[None, :, :]) ** 2).sum(-1), 0))
        want = np.sort(D, axis=1)[:, :8]
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6), np.abs(got - want).max()
        assert np.allclose(D[np.arange(40)[:, None], pos], want, rtol=1e-5, atol=1e-6)
        if ids is None:
            assert (pos[:, 0] == qid).all()          # a point is its own nearest neighbour


def _fit_model(dist=None, sharding=None):
    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel

    rng = np.random.default_rng(0)
    N, R = 120, 4
    X = np.stack([rng.integers(0, N, 900), rng.integers(0, R, 900), rng.integers(0, N, 900)], 1)
    X = np.char.add(np.array(["e", "r", "e"]), X.astype(str))
    m = ScoringBasedEmbeddingModel(eta=3, k=10, scoring_type="ComplEx", seed=2)
    if dist is not None:
        m._dist_override = dist
    kw = dict(entity_sharding="rows", sharded_negatives="global") if sharding else {}
    m.compile(optimizer="adam", loss="nll", **kw)
    m.fit(X, batch_size=300, epochs=2, verbose=False)
    return m, X


def _check_discovery(m, X):
    from ampligraph_amd.discovery import find_nearest_neighbours, query_topn

    ents = np.unique(np.concatenate([X[:, 0], X[:, 2]]))
    # tail completion == brute force over predict()
    Y, S = query_topn(m, top_n=7, head="e5", relation="r1")
    cand = np.stack([np.full(len(ents), "e5"), np.full(len(ents), "r1"), ents], 1)
    sc = m.predict(cand)
    order = np.argsort(-sc, kind="stable")[:7]
    assert np.allclose(S, sc[order], rtol=1e-5, atol=1e-6) and (np.diff(S) <= 0).all()
    assert np.allclose(m.predict(Y), S, rtol=1e-5, atol=1e-6) and (Y[:, 0] == "e5").all() and (Y[:, 1] == "r1").all()
    # head completion among a subset
    subset = ents[::3]
    Y, S = query_topn(m, top_n=5, relation="r2", tail="e9", ents_to_consider=list(subset))
    sc = m.predict(np.stack([subset, np.full(len(subset), "r2"), np.full(len(subset), "e9")], 1))
    assert np.allclose(S, np.sort(sc)[::-1][:5], rtol=1e-5, atol=1e-6) and set(Y[:, 0]) <= set(subset)
    # relation completion
    Y, S = query_topn(m, top_n=3, head="e1", tail="e2")
    sc = m.predict(np.stack([np.full(4, "e1"), np.array(["r0", "r1", "r2", "r3"]), np.full(4, "e2")], 1))
    assert np.allclose(S, np.sort(sc)[::-1][:3], rtol=1e-5, atol=1e-6)
    # nearest neighbours, both metrics, whole table and subset
    E = m.get_embeddings(ents).astype(np.float64)
    q = ["e3", "e77", "e40"]
    Q = m.get_embeddings(np.array(q)).astype(np.float64)
    nb, dist = find_nearest_neighbours(m, q, n_neighbors=6)
    D = np.linalg.norm(Q[:, None, :] - E[None, :, :], axis=2)
    assert np.allclose(dist, np.sort(D, axis=1)[:, :6], rtol=1e-5, atol=1e-6) and (nb[:, 0] == np.array(q)).all()
    nb, dist = find_nearest_neighbours(m, q, n_neighbors=4, entities_subset=list(subset), metric="cosine")
    Es = m.get_embeddings(subset).astype(np.float64)
    Dc = 1 - (Q @ Es.T) / np.linalg.norm(Q, axis=1)[:, None] / np.linalg.norm(Es, axis=1)[None, :]
    assert np.allclose(dist, np.sort(Dc, axis=1)[:, :4], rtol=1e-5, atol=2e-6) and set(nb.reshape(-1)) <= set(subset)
    return query_topn(m, top_n=7, head="e5", relation="r1")


def test_discovery_surface_single_gpu(gpu_lib):
    m, X = _fit_model()
    _check_discovery(m, X)


def test_query_topn_beyond_1024(gpu_lib):
    """top_n larger than the streaming selection's 1024: the reference accepts any top_n (argsort over all candidates,
    discovery.py:985-1168); ours must too, with the same order as brute force over predict()."""
    from ampligraph_amd.discovery import query_topn
    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel

    rng = np.random.default_rng(5)
    N, R = 3000, 3
    X = np.stack([rng.permutation(N), rng.integers(0, R, N), rng.permutation(N)], 1)
    X = np.char.add(np.array(["e", "r", "e"]), X.astype(str))
    m = ScoringBasedEmbeddingModel(eta=2, k=8, scoring_type="DistMult", seed=1)
    m.compile(optimizer="adam", loss="nll")
    m.fit(X, batch_size=1000, epochs=1, verbose=False)
    ents = np.unique(np.concatenate([X[:, 0], X[:, 2]]))
    Y, S = query_topn(m, top_n=2500, head="e7", relation="r1")
    sc = m.predict(np.stack([np.full(len(ents), "e7"), np.full(len(ents), "r1"), ents], 1))
    assert Y.shape == (2500, 3) and len(set(Y[:, 2])) == 2500
    assert np.allclose(S, np.sort(sc)[::-1][:2500], rtol=1e-5, atol=1e-6) and (np.diff(S) <= 0).all()
    assert np.allclose(m.predict(Y), S, rtol=1e-5, atol=1e-6)


def test_discovery_surface_row_sharded(gpu_lib):
    """The same checks with the entity table row-sharded over two engines on the one GPU (in-process rendezvous)."""
    from threaded_dist import ThreadedWorld

    def body(dist):
        m, X = _fit_model(dist, sharding=True)
        return _check_discovery(m, X)

    res = ThreadedWorld(2).run(body)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])   # replicas agree
