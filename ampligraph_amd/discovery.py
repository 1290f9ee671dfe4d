"""Discovery helpers that are thin callers of the hot path: query_topn
(/root/reference/ampligraph/discovery/discovery.py:985-1168) and find_nearest_neighbours (:1171-1244).

The reference materialises one STRING triple per candidate and calls model.predict; here the candidate triples are
built as int32 ids on the device, scored by one launch of the libamdkge score kernel (the fixed s / p rows stay in
L2) and only the top_n ids / scores travel back.  Same arguments, validation and error behaviour."""
import numpy as np


def _known(indexer, values, type_of):
    vals = np.asarray(values).reshape(-1)
    return len(indexer.get_indexes(vals, type_of)) == len(vals)


def query_topn(model, top_n=10, head=None, relation=None, tail=None, ents_to_consider=None, rels_to_consider=None):
    """Score every completion of the two given triple elements and return the top_n (triples (n,3) of raw labels,
    scores (n,) float32), ordered by decreasing score.  True statements are not filtered out (as in the reference)."""
    import torch

    if not model.is_fitted:
        raise ValueError("Model is not fitted.")
    if not np.sum([head is None, relation is None, tail is None]) == 1:
        raise ValueError("Exactly one of `head`, `relation` or `tail` arguments must be None.")
    ix = model.data_indexer
    if head and not _known(ix, [head], "e"):
        raise ValueError("Head entity `{}` not seen by model".format(head))
    if relation and not _known(ix, [relation], "r"):
        raise ValueError("Relation `{}` not seen by model".format(relation))
    if tail and not _known(ix, [tail], "e"):
        raise ValueError("Tail entity `{}` not seen by model".format(tail))
    if ents_to_consider is not None:
        if head and tail:
            raise ValueError("Cannot specify `ents_to_consider` and both `subject` and `object` arguments.")
        if not isinstance(ents_to_consider, (list, np.ndarray)):
            raise ValueError("`ents_to_consider` must be a list or numpy array.")
        if not _known(ix, ents_to_consider, "e"):
            raise ValueError("Entities in `ents_to_consider` have not been seen by the model.")
    if rels_to_consider is not None:
        if relation:
            raise ValueError("Cannot specify both `rels_to_consider` and `relation` arguments.")
        if not isinstance(rels_to_consider, (list, np.ndarray)):
            raise ValueError("`rels_to_consider` must be a list or numpy array.")
        if not _known(ix, rels_to_consider, "r"):
            raise ValueError("Relations in `rels_to_consider` have not been seen by the model.")
    if getattr(model, "_spec", None) is not None:
        raise NotImplementedError("query_topn with a row-sharded entity table")
    eng = model._engine
    dev = eng.device

    def ids(values, type_of, count):
        if values is None or len(values) == 0:
            return torch.arange(count, dtype=torch.int32, device=dev)
        return torch.as_tensor(np.asarray(ix.get_indexes(np.asarray(values), type_of), dtype=np.int32)).to(dev)

    one = lambda v, t: int(ix.get_indexes(np.asarray([v]), t)[0])
    if relation is None:
        cand = ids(rels_to_consider, "r", model._n_rels)
        cols = [torch.full_like(cand, one(head, "e")), cand, torch.full_like(cand, one(tail, "e"))]
    else:
        cand = ids(ents_to_consider, "e", model._n_ents)
        r = torch.full_like(cand, one(relation, "r"))
        cols = [torch.full_like(cand, one(head, "e")), r, cand] if head else [cand, r, torch.full_like(cand, one(tail, "e"))]
    tri = torch.stack(cols, 1).contiguous()
    scores = eng.score(tri)
    n = min(int(top_n), int(tri.shape[0]))
    top_s, top_i = torch.topk(scores, n, largest=True, sorted=True)
    out = tri[top_i.long()].cpu().numpy()
    return ix.get_indexes(out, "t", "ind2raw"), top_s.cpu().numpy().astype(np.float32)


def find_nearest_neighbours(kge_model, entities, n_neighbors=10, entities_subset=None, metric="euclidean"):
    """k nearest neighbours of `entities` in embedding space (:1171-1244; the reference delegates to
    sklearn.neighbors.NearestNeighbors on the host).  Distances on the device for "euclidean" / "cosine"; other
    sklearn metrics fall back to sklearn on the downloaded embeddings.  Returns (neighbour labels, distances), each
    (len(entities), n_neighbors), nearest first."""
    import torch

    assert kge_model.is_fitted, "KGE model is not fit!"
    assert isinstance(entities, (list, np.ndarray)), "Invalid type for entities! Must be a list or np.array"
    ix = kge_model.data_indexer
    if entities_subset is not None:
        assert isinstance(entities_subset, (list, np.ndarray)), "Invalid type for entities_subset! Must be a list or np.array"
        all_neighbors = np.asarray(entities_subset)
        cand = np.asarray(ix.get_indexes(all_neighbors, "e"), dtype=np.int64)
    else:
        cand = np.arange(kge_model._n_ents, dtype=np.int64)
        all_neighbors = ix.get_indexes(cand, "e", "ind2raw")
    assert n_neighbors < len(all_neighbors), "n_neighbors must be less than the number of entities being fit!"
    tab = kge_model._entity_table()
    unpack = kge_model._engine.unpack
    E = unpack(tab[torch.as_tensor(cand).to(tab.device)])
    Q = unpack(tab[torch.as_tensor(np.asarray(ix.get_indexes(np.asarray(entities), "e"), dtype=np.int64)).to(tab.device)])
    if metric in ("euclidean", "l2", "minkowski"):
        d = torch.cdist(Q.double(), E.double()).float()
    elif metric == "cosine":
        qn, en = torch.nn.functional.normalize(Q.double(), dim=1), torch.nn.functional.normalize(E.double(), dim=1)
        d = (1.0 - qn @ en.T).float()
    else:
        from sklearn.neighbors import NearestNeighbors

        knn = NearestNeighbors(n_neighbors=n_neighbors, metric=metric).fit(E.cpu().numpy())
        dist, idx = knn.kneighbors(Q.cpu().numpy())
        return np.asarray(all_neighbors)[idx], dist
    dist, idx = torch.topk(d, int(n_neighbors), dim=1, largest=False, sorted=True)
    return np.asarray(all_neighbors)[idx.cpu().numpy()], dist.cpu().numpy()
