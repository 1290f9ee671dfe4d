"""Discovery helpers on the device: query_topn
(/root/reference/ampligraph/discovery/discovery.py:985-1168) and find_nearest_neighbours (:1171-1244).

The reference materialises one STRING triple per candidate, calls model.predict and argsorts on the host; its nearest
neighbours are sklearn on the host.  Here an entity completion is ONE query through the 1-vs-all corruption-score kernels
of evaluate() (amdkge_corruption_scores: same prep + tile kernels as the ranks) followed by a streaming top-k selection
kernel (amdkge_topk_rows); nearest neighbours are dot products on the same tile kernel with the norms folded into the
selection.  Only top_n ids / scores travel back.  Both work on a row-sharded entity table (per-shard lists, merged).
Same arguments, validation and error behaviour as the reference."""
import numpy as np

from . import _ffi


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
    eng = model._engine
    dev = eng.device
    sp = getattr(model, "_spec", None)
    one = lambda v, t: int(ix.get_indexes(np.asarray([v]), t)[0])   # noqa: E731

    if relation is None:   # complete the relation: a handful of candidates, scored as ordinary triples
        if rels_to_consider is None or len(rels_to_consider) == 0:
            cand = np.arange(model._n_rels, dtype=np.int32)
        else:
            cand = np.asarray(ix.get_indexes(np.asarray(rels_to_consider), "r"), dtype=np.int32)
        tri = np.stack([np.full_like(cand, one(head, "e")), cand, np.full_like(cand, one(tail, "e"))], 1)
        scores = model._score_dev(torch.as_tensor(tri).to(dev))
        n = min(int(top_n), len(cand))
        idx, val = eng.topk_rows(scores.view(1, -1), n)
        out = tri[idx[0].cpu().numpy().astype(np.int64)]
        return ix.get_indexes(out, "t", "ind2raw"), val[0].cpu().numpy().astype(np.float32)

    # complete an entity: 1-vs-all corruption scores of ONE query on the rank kernels + a top-k selection on the device
    side = _ffi.SIDE_O if tail is None else _ffi.SIDE_S
    fixed = one(head if tail is None else tail, "e")
    r_id = one(relation, "r")
    q = np.array([[fixed, r_id, fixed]], dtype=np.int32)   # the replaced column is ignored by the corruption scores
    cand_ids = None
    if ents_to_consider is not None and len(ents_to_consider) > 0:
        cand_ids = np.asarray(ix.get_indexes(np.asarray(ents_to_consider), "e"), dtype=np.int64)
    n_cand = model._n_ents if cand_ids is None else len(cand_ids)
    n = min(int(top_n), n_cand)
    # (top_n > 1024: engine.topk_rows sorts the whole score row on the device instead of the streaming selection)
    if sp is None:
        ids_dev = None if cand_ids is None else torch.as_tensor(cand_ids.astype(np.int32)).to(dev)
        pos, val = eng.corruption_topk(torch.as_tensor(q).to(dev), side, n, ent_ids=ids_dev)
        pos = pos[0].cpu().numpy().astype(np.int64)
        ents = pos if cand_ids is None else cand_ids[pos]
        val = val[0].cpu().numpy()
    else:
        # row-sharded table: every rank selects among ITS rows (the query's own rows are fetched behind the shard), the
        # W partial lists (global ids, scores) are gathered and merged by a second selection
        d = model._dist()
        ql = model._localise(torch.as_tensor(q).to(dev))
        if cand_ids is None:
            loc, n_loc = None, sp.n_local
        else:
            loc = sp.local_subset(torch.as_tensor(cand_ids).to(dev))[0]
            n_loc = int(loc.shape[0])
        k_loc = n
        gid = torch.full((1, k_loc), -1, dtype=torch.int32, device=dev)
        gval = torch.full((1, k_loc), float("-inf"), dtype=torch.float32, device=dev)
        if n_loc > 0:
            kk = min(k_loc, n_loc)
            pos, val = eng.corruption_topk(ql, side, kk, ent_ids=loc, ent_lo=0, ent_hi=n_loc)
            rows = pos[0].to(torch.int64) if loc is None else loc[pos[0].to(torch.int64)].to(torch.int64)
            gid[0, :kk] = (rows + sp.lo).to(torch.int32)
            gval[0, :kk] = val[0]
        parts_i = [torch.empty_like(gid) for _ in range(sp.world)]
        parts_v = [torch.empty_like(gval) for _ in range(sp.world)]
        d.all_gather(parts_i, gid)
        d.all_gather(parts_v, gval)
        ents, val = eng.topk_rows(torch.cat(parts_v, 1).contiguous(), n, payload=torch.cat(parts_i, 1).contiguous())
        ents, val = ents[0].cpu().numpy().astype(np.int64), val[0].cpu().numpy()
    rel_col = np.full(n, r_id, dtype=np.int64)
    fix_col = np.full(n, fixed, dtype=np.int64)
    out = np.stack([fix_col, rel_col, ents], 1) if tail is None else np.stack([ents, rel_col, fix_col], 1)
    return ix.get_indexes(out, "t", "ind2raw"), val.astype(np.float32)


def find_nearest_neighbours(kge_model, entities, n_neighbors=10, entities_subset=None, metric="euclidean"):
    """k nearest neighbours of `entities` in embedding space (:1171-1244; the reference delegates to
    sklearn.neighbors.NearestNeighbors on the host).  "euclidean" / "cosine": on the device -- dot products on the rank
    tile kernel (GEMM form), norms folded into a per-row top-k selection; works with a row-sharded table (partial lists per
    shard, merged).  Other sklearn metrics fall back to sklearn on the downloaded embeddings.  Returns (neighbour labels,
    distances), each (len(entities), n_neighbors), nearest first."""
    import torch

    assert kge_model.is_fitted, "KGE model is not fit!"
    assert isinstance(entities, (list, np.ndarray)), "Invalid type for entities! Must be a list or np.array"
    ix = kge_model.data_indexer
    if entities_subset is not None:
        assert isinstance(entities_subset, (list, np.ndarray)), "Invalid type for entities_subset! Must be a list or np.array"
        all_neighbors = np.asarray(entities_subset)
        cand = np.asarray(ix.get_indexes(all_neighbors, "e"), dtype=np.int64)
    else:
        cand = None
        all_neighbors = None
    n_all = kge_model._n_ents if cand is None else len(cand)
    assert n_neighbors < n_all, "n_neighbors must be less than the number of entities being fit!"
    eng = kge_model._engine
    dev = eng.device
    sp = getattr(kge_model, "_spec", None)
    qid = np.asarray(ix.get_indexes(np.asarray(entities), "e"), dtype=np.int64)
    k = int(n_neighbors)
    if metric not in ("euclidean", "l2", "minkowski", "cosine") or k > 1024:
        from sklearn.neighbors import NearestNeighbors

        labels = all_neighbors if cand is not None else ix.get_indexes(np.arange(kge_model._n_ents), "e", "ind2raw")
        E = kge_model.get_embeddings(labels)
        knn = NearestNeighbors(n_neighbors=n_neighbors, metric=metric).fit(E)
        dist, idx = knn.kneighbors(kge_model.get_embeddings(np.asarray(entities)))
        return np.asarray(labels)[idx], dist
    met = "cosine" if metric == "cosine" else "euclidean"
    if sp is None:
        Q = eng.ent[torch.as_tensor(qid).to(dev)]
        ids_dev = None if cand is None else torch.as_tensor(cand.astype(np.int32)).to(dev)
        pos, dist = eng.nearest_rows(Q, k, met, ent_ids=ids_dev)
        pos = pos.cpu().numpy().astype(np.int64)
        ids = pos if cand is None else cand[pos]
        dist = dist.cpu().numpy()
    else:
        d = kge_model._dist()
        fake = np.stack([qid, np.zeros_like(qid), qid], 1).astype(np.int32)
        ql = kge_model._localise(torch.as_tensor(fake).to(dev))          # query rows, fetched behind the shard where remote
        Q = eng.ent[ql[:, 0].to(torch.int64)]
        if cand is None:
            loc, n_loc = None, sp.n_local
        else:
            loc = sp.local_subset(torch.as_tensor(cand).to(dev))[0]
            n_loc = int(loc.shape[0])
        nq = len(qid)
        gid = torch.full((nq, k), -1, dtype=torch.int32, device=dev)
        gd = torch.full((nq, k), float("inf"), dtype=torch.float32, device=dev)
        if n_loc > 0:
            kk = min(k, n_loc)
            pos, dl = eng.nearest_rows(Q, kk, met, ent_ids=loc, ent_lo=0, ent_hi=n_loc)
            rows = pos.to(torch.int64) if loc is None else loc[pos.to(torch.int64)].to(torch.int64)
            gid[:, :kk] = (rows + sp.lo).to(torch.int32)
            gd[:, :kk] = dl
        parts_i = [torch.empty_like(gid) for _ in range(sp.world)]
        parts_d = [torch.empty_like(gd) for _ in range(sp.world)]
        d.all_gather(parts_i, gid)
        d.all_gather(parts_d, gd)
        ids, dist = eng.topk_rows(torch.cat(parts_d, 1).contiguous(), k, largest=False, payload=torch.cat(parts_i, 1).contiguous())
        ids, dist = ids.cpu().numpy().astype(np.int64), dist.cpu().numpy()
    labels = ix.get_indexes(ids.reshape(-1), "e", "ind2raw").reshape(ids.shape)
    return labels, dist.astype(np.float32)
