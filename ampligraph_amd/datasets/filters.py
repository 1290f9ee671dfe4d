"""True-positive filter sets for evaluate(use_filter=...), as sort-based CSR instead of the
reference's pandas groupby + Python `sum(lists, [])` per batch
(/root/reference/ampligraph/datasets/graph_data_loader.py:287-350,382-439).

Semantics kept: for a test triple (s,p,o) the subject-side filter is the SET {s' : (s',p,o) in any
filter dataset} and the object-side filter the SET {o' : (s,p,o') in any filter dataset}; all ids are
the training id map's.  The index is built once per evaluate() call; a test triple's filter is a
[lo, hi) range into one shared id array, so no per-batch host work remains."""
import numpy as np


class FilterIndex:
    def __init__(self, datasets, n_ents, n_rels, engine=None):
        """engine=None: the index is built on the host (numpy sorts; the checker of the device build and what GPU-less callers
        get).  With a KgeEngine the id triples are uploaded once and the index is built ON THE DEVICE by amdkge_filter_build
        (kge_filter.hip: key generation, radix sort, scan, scatter) -- evaluate() then does no host sort at all; the host arrays
        (po_keys, ...) are downloaded lazily only if somebody asks for them."""
        if engine is not None:
            self._init_device(datasets, n_ents, n_rels, engine)
            return
        X = np.concatenate([np.asarray(d)[:, :3].astype(np.int64) for d in datasets], 0) if len(datasets) else \
            np.zeros((0, 3), dtype=np.int64)
        self.n_ents, self.n_rels = int(n_ents), int(n_rels)
        N, R = self.n_ents, self.n_rels
        if R * N * N >= 2 ** 63:   # (python ints: no wrap) the packed int64 sort keys below would overflow silently
            raise ValueError(f"FilterIndex: n_rels * n_ents^2 = {R * N * N} does not fit the packed int64 keys "
                             "(n_ents up to ~96 M at 1 000 relations)")
        s, p, o = X[:, 0], X[:, 1], X[:, 2]
        # subject side: group by (p,o), values s (unique)
        k_s = np.unique((p * N + o) * N + s)
        self.po_keys, self.po_start = np.unique(k_s // N, return_index=True)
        self.po_start = np.append(self.po_start, k_s.size).astype(np.int64)
        self.s_ids = (k_s % N).astype(np.int32)
        # object side: group by (s,p), values o (unique)
        k_o = np.unique((s * R + p) * N + o)
        self.sp_keys, self.sp_start = np.unique(k_o // N, return_index=True)
        self.sp_start = np.append(self.sp_start, k_o.size).astype(np.int64)
        self.o_ids = (k_o % N).astype(np.int32)

    def _init_device(self, datasets, n_ents, n_rels, engine):
        import torch

        self.n_ents, self.n_rels = int(n_ents), int(n_rels)
        N, R = self.n_ents, self.n_rels
        if R * N * N >= 2 ** 63:
            raise ValueError(f"FilterIndex: n_rels * n_ents^2 = {R * N * N} does not fit the packed int64 keys "
                             "(n_ents up to ~96 M at 1 000 relations)")
        dev = engine.device
        parts = []
        for d in datasets:
            if isinstance(d, torch.Tensor):
                parts.append(d[:, :3].to(device=dev, dtype=torch.int32))
            else:
                parts.append(torch.as_tensor(np.ascontiguousarray(np.asarray(d)[:, :3], dtype=np.int32)).to(dev))
        X = torch.cat(parts, 0).contiguous() if parts else torch.zeros(0, 3, dtype=torch.int32, device=dev)
        built = {sd: engine.filter_build(X, sd, N, R) for sd in ("s", "o")}
        (pk, ps, si), (sk, ss, oi) = built["s"], built["o"]
        one = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev = {"device": str(dev), "po_keys": pk, "po_start": ps, "s_ids": si if si.numel() else one,
                     "sp_keys": sk, "sp_start": ss, "o_ids": oi if oi.numel() else one}
        self._dev_sizes = {"s_ids": int(si.numel()), "o_ids": int(oi.numel())}

    def __getattr__(self, name):
        # host views of a device-built index, on demand (tests, bench's host-side range lookups)
        if name in ("po_keys", "po_start", "sp_keys", "sp_start", "s_ids", "o_ids") and "_dev" in self.__dict__:
            t = self.__dict__["_dev"][name]
            if name in ("s_ids", "o_ids"):
                t = t[:self.__dict__["_dev_sizes"][name]]
            a = t.cpu().numpy()
            self.__dict__[name] = a
            return a
        raise AttributeError(name)

    @staticmethod
    def _ranges(keys, start, q):
        if keys.size == 0:
            z = np.zeros(q.shape[0], dtype=np.int64)
            return z, z.copy()
        pos = np.searchsorted(keys, q)
        pos_c = np.minimum(pos, keys.size - 1)
        hit = keys[pos_c] == q
        lo = np.where(hit, start[pos_c], 0).astype(np.int64)
        hi = np.where(hit, start[pos_c + 1], 0).astype(np.int64)
        return lo, hi

    def subject_ranges(self, triples):
        t = np.asarray(triples)[:, :3].astype(np.int64)
        return self._ranges(self.po_keys, self.po_start, t[:, 1] * self.n_ents + t[:, 2])

    def object_ranges(self, triples):
        t = np.asarray(triples)[:, :3].astype(np.int64)
        return self._ranges(self.sp_keys, self.sp_start, t[:, 0] * self.n_rels + t[:, 1])

    def device_filter(self, engine, triples_dev, side):
        """(lo, hi, ids) device tensors for amdkge_rank_filter: the range lookup of subject_ranges / object_ranges done on
        the engine's device by amdkge_filter_ranges (keys, starts and ids are uploaded once per index and kept), so an
        evaluate() call does no per-triple host work.  triples_dev: (n,3) int32 device tensor; side "s" | "o"."""
        import torch

        device = engine.device
        cache = self.__dict__.setdefault("_dev", {})
        if cache.get("device") != str(device):
            # another device than the one the index lives on (or a host-built index): the host copies first -- for a
            # device-built index they exist only through __getattr__, which reads the very cache replaced here
            host = {nm: getattr(self, nm) for nm in ("po_keys", "po_start", "sp_keys", "sp_start", "s_ids", "o_ids")}
            fresh = {"device": str(device)}
            for nm in ("po_keys", "po_start", "sp_keys", "sp_start"):
                fresh[nm] = torch.as_tensor(host[nm]).to(device)
            for nm in ("s_ids", "o_ids"):
                fresh[nm] = torch.as_tensor(host[nm] if host[nm].size else np.zeros(1, np.int32)).to(device)
            cache.clear()
            cache.update(fresh)
            self.__dict__["_dev_sizes"] = {"s_ids": int(host["s_ids"].size), "o_ids": int(host["o_ids"].size)}
        keys, start, ids = (cache["po_keys"], cache["po_start"], cache["s_ids"]) if side == "s" else \
            (cache["sp_keys"], cache["sp_start"], cache["o_ids"])
        lo, hi = engine.filter_ranges(keys, start, triples_dev, 1 if side == "s" else 2, self.n_ents, self.n_rels)
        return lo, hi, ids

    def as_lists(self, triples):
        """Materialise per-triple id arrays (what the reference yields as a RaggedTensor); tests only."""
        slo, shi = self.subject_ranges(triples)
        olo, ohi = self.object_ranges(triples)
        return ([self.s_ids[a:b] for a, b in zip(slo, shi)], [self.o_ids[a:b] for a, b in zip(olo, ohi)])
