"""numpy-in / numpy-out driver of the library's session layer (include/amdkge.h, amdkge_session_*): the whole model lives
behind one opaque handle inside libamdkge (tables, optimizer state, scratch, stream); this class only marshals numpy
arrays.  It is what a host WITHOUT torch would write against the C ABI (option B) -- the drop-in class
ampligraph_amd.latent_features.ScoringBasedEmbeddingModel keeps its device-resident path through engine.KgeEngine.

    s = Session("ComplEx", k=200, n_ents=N, n_rels=R, eta=20, loss=loss_functions.get("self_adversarial"),
                optimizer=optimizers.get("adam"), seed=0)
    s.set_rows("ent", ent0); s.set_rows("rel", rel0)
    loss = s.train_step(triples_int32)          # one batch of ScoringBasedEmbeddingModel.train_step (:370-429)
    scores = s.score(triples_int32)             # predict (:1694-1734)
    ranks = s.rank(test, filters_s=(off, ids), filters_o=(off, ids), corrupt_side="s,o")   # evaluate (:1516-1692)
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def _config(scoring_type, k, n_ents, n_rels, eta, loss, optimizer, regularizer, rel_regularizer, seed, device, pos_atomic,
            focus_nonlinearity, deterministic):
    cfg = _ffi.SessionConfig()
    cfg.model = _ffi.Model(_ffi.SCORING_TYPES[scoring_type], int(k), int(n_ents), int(n_rels), int(n_rels), 0)
    cfg.loss = loss.to_ffi()
    if focus_nonlinearity is not None:
        cfg.loss.focus_nonlinearity = _ffi.FOCUS_NONLINEARITY[focus_nonlinearity]
    from .engine import reg_fields

    cfg.opt = optimizer.to_ffi(1, 2)
    cfg.opt.reg_p, cfg.opt.reg_lambda, cfg.opt.reg2_p, cfg.opt.reg2_lambda = reg_fields(regularizer, 2)
    rr = rel_regularizer if rel_regularizer is not None else regularizer
    rp1, rl1, rp2, rl2 = reg_fields(rr, 2)
    if rl1 == 0.0 and rl2 != 0.0:
        rp1, rl1, rl2 = rp2, rl2, 0.0
    cfg.rel_reg_lambda, cfg.opt.rel_reg_p, cfg.opt.rel_reg2_p, cfg.opt.rel_reg2_lambda = rl1, rp1, rp2, rl2
    cfg.eta, cfg.seed, cfg.device, cfg.flags = int(eta), int(seed), int(device), (1 if pos_atomic else 0) | (2 if deterministic else 0)
    return cfg


def _csr_pair(f):
    return (None, None) if f is None else (np.ascontiguousarray(f[0], dtype=np.int64), _i32(f[1]))


def _rank_call(fn, handle, t, fs, fo, entities_subset, corrupt_side, ranking_strategy):
    n = int(t.shape[0])
    sub = _i32(entities_subset) if entities_subset is not None and len(entities_subset) else None
    cols = 2 if corrupt_side == "s,o" else 1
    out = np.empty((n, cols), dtype=np.int32)
    check(fn(handle, _p(t), n, _p(fs[0]), _p(fs[1]), _p(fo[0]), _p(fo[1]), _p(sub), int(sub.shape[0]) if sub is not None else 0,
             _ffi.CORRUPT_SIDES[corrupt_side], _ffi.RANK_STRATEGY[ranking_strategy], _p(out)))
    return out


class Session:
    def __init__(self, scoring_type, k, n_ents, n_rels, eta, loss, optimizer, regularizer=None, rel_regularizer=None, seed=0,
                 device=0, pos_atomic=False, focus_nonlinearity=None, deterministic=False, _handle=None):
        self.lib = _ffi.lib()
        self.K = int(self.lib.amdkge_internal_k(_ffi.SCORING_TYPES[scoring_type], int(k)))
        self.n_ents, self.n_rels = int(n_ents), int(n_rels)
        if _handle is not None:   # a replica of a SessionGroup: the group owns the handle
            self._h, self._owned = _handle, False
            return
        self._owned = True
        cfg = _config(scoring_type, k, n_ents, n_rels, eta, loss, optimizer, regularizer, rel_regularizer, seed, device, pos_atomic,
                      focus_nonlinearity, deterministic)
        self._h = C.c_void_p()
        check(self.lib.amdkge_session_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and getattr(self, "_owned", True):
            self.lib.amdkge_session_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the library handle may already be gone
            pass

    def set_rows(self, table, values, row0=0):
        v = np.ascontiguousarray(values, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.K:
            raise ValueError(f"rows must have {self.K} floats")
        check(self.lib.amdkge_session_set_rows(self._h, _ffi.TABLES[table], int(row0), int(v.shape[0]), _p(v)))

    def get_rows(self, table, ids=None, row0=0, nrows=None):
        if ids is not None:
            ids = _i32(ids)
            out = np.empty((ids.shape[0], self.K), dtype=np.float32)
            check(self.lib.amdkge_session_get_rows(self._h, _ffi.TABLES[table], _p(ids), 0, int(ids.shape[0]), _p(out)))
            return out
        if nrows is None:
            nrows = (self.n_ents if table.startswith("ent") else self.n_rels) - int(row0)
        out = np.empty((int(nrows), self.K), dtype=np.float32)
        check(self.lib.amdkge_session_get_rows(self._h, _ffi.TABLES[table], None, int(row0), int(nrows), _p(out)))
        return out

    def set_hot_rows(self, ids):
        """Skewed graphs: up to 64 hot entity rows get replica rows (amdkge_session_set_hot_rows); None / [] switches it off."""
        a = _i32(ids if ids is not None else [])
        check(self.lib.amdkge_session_set_hot_rows(self._h, _p(a) if a.shape[0] else None, int(a.shape[0])))

    def train_step(self, triples, focus_w=None):
        t = _i32(triples)
        fw = np.ascontiguousarray(focus_w, dtype=np.float32) if focus_w is not None else None
        loss = C.c_double(0.0)
        check(self.lib.amdkge_session_train_step(self._h, _p(t), int(t.shape[0]), _p(fw), C.byref(loss)))
        return float(loss.value)

    def score(self, triples):
        t = _i32(triples)
        out = np.empty(t.shape[0], dtype=np.float32)
        check(self.lib.amdkge_session_score(self._h, _p(t), int(t.shape[0]), _p(out)))
        return out

    def rank(self, triples, filters_s=None, filters_o=None, entities_subset=None, corrupt_side="s,o", ranking_strategy="worst"):
        """filters_*: None or (offsets int64 [n+1], ids int32) CSR over the test triples (FilterIndex ranges work after
        np.concatenate; see datasets/filters.py).  Returns int32 (n, 1|2) like evaluate()."""
        t = _i32(triples)
        fs, fo = _csr_pair(filters_s), _csr_pair(filters_o)
        return _rank_call(self.lib.amdkge_session_rank, self._h, t, fs, fo, entities_subset, corrupt_side, ranking_strategy)


    def screen_stats(self):
        """Of the last rank() (its last side): None when the count pass was handed no screening / early-exit workspace, else (pairs
        the exact chain re-checked, fell back to the exact kernel?) -- contraction models: the int8 screening pass; TransE / RotatE:
        the exact early exit ((0, False) when its probe picked the plain kernel, and for problems too small for either pass)."""
        ran, pairs, fb = C.c_int32(0), C.c_int64(0), C.c_int32(0)
        check(self.lib.amdkge_session_screen_stats(self._h, C.byref(ran), C.byref(pairs), C.byref(fb)))
        return (int(pairs.value), bool(fb.value)) if ran.value else None


class SessionGroup:
    """amdkge_session_group_*: the session layer on several GPUs from ONE process, numpy only -- data-parallel training with the
    library's own RCCL communicators (kge_session_group.hip).  `devices`: distinct ordinals (RCCL all-reduce over xGMI) or the
    same ordinal repeated (replicas share one GPU and sum with a kernel: tests on a one-GPU box).  Every replica holds the whole
    model: replica(i) is an ordinary Session view for get_rows / score / rank."""

    def __init__(self, devices, scoring_type, k, n_ents, n_rels, eta, loss, optimizer, regularizer=None, rel_regularizer=None, seed=0,
                 pos_atomic=False, focus_nonlinearity=None, deterministic=False, force_rccl=False, rows=False, max_batch=None,
                 global_negatives=False, cols=False, force_threads=False):
        """force_rccl: AMDKGE_GROUP_FORCE_RCCL -- a group of one replica still binds librccl and sums its gradients through a
        one-rank ncclAllReduce (first contact with RCCL on a one-GPU box).  rows=True: the entity table ROW-SHARDED over the
        replicas (amdkge_session_group_create_rows; n_ents is the global count, max_batch the largest batch of a step,
        global_negatives reproduces one GPU's corruptions instead of drawing shard-local ones); set_rows / get_rows then speak
        global row numbers.  cols=True: every table COLUMN-sharded over the replicas (amdkge_session_group_create_cols: replica d
        holds k / W units of every row and processes the whole batch on them; one all-reduce of the partial scores per step);
        set_rows / get_rows take and return whole rows.  force_threads: AMDKGE_GROUP_FORCE_THREADS -- rank() drives every replica
        from a host thread of its own although the replicas share a device (replicas on distinct devices always get one)."""
        self.lib = _ffi.lib()
        self._args = (scoring_type, int(k), int(n_ents), int(n_rels), int(eta), loss, optimizer)
        self.K = int(self.lib.amdkge_internal_k(_ffi.SCORING_TYPES[scoring_type], int(k)))
        cfg = _config(scoring_type, k, n_ents, n_rels, eta, loss, optimizer, regularizer, rel_regularizer, seed, 0, pos_atomic,
                      focus_nonlinearity, deterministic)
        dev = _i32(list(devices))
        self._g = C.c_void_p()
        self.rows, self.n_ents, self.n_rels = bool(rows), int(n_ents), int(n_rels)
        self.cols = bool(cols)
        if cols:
            check(self.lib.amdkge_session_group_create_cols(C.byref(cfg), _p(dev), int(dev.shape[0]), 1 if force_rccl else 0, C.byref(self._g)))
        elif rows:
            if max_batch is None:
                raise ValueError("rows=True needs max_batch (the largest batch a step will be given)")
            check(self.lib.amdkge_session_group_create_rows(C.byref(cfg), _p(dev), int(dev.shape[0]), (1 if force_rccl else 0) | (4 if global_negatives else 0) | (16 if force_threads else 0),
                                                            int(max_batch), C.byref(self._g)))
        else:
            check(self.lib.amdkge_session_group_create_ex(C.byref(cfg), _p(dev), int(dev.shape[0]), (1 if force_rccl else 0) | (16 if force_threads else 0),
                                                          C.byref(self._g)))
        self.size = int(self.lib.amdkge_session_group_size(self._g))

    def info(self):
        """(sums through RCCL?, ncclGetVersion code or 0)"""
        u, v = C.c_int32(0), C.c_int32(0)
        check(self.lib.amdkge_session_group_info(self._g, C.byref(u), C.byref(v)))
        return bool(u.value), int(v.value)

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self.lib.amdkge_session_group_destroy(self._g)
        self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def replica(self, i):
        h = C.c_void_p()
        check(self.lib.amdkge_session_group_replica(self._g, int(i), C.byref(h)))
        st, k, ne, nr, eta, loss, opt = self._args
        return Session(st, k, ne, nr, eta, loss, opt, _handle=h)

    def set_rows(self, table, values, row0=0):
        v = np.ascontiguousarray(values, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.K:
            raise ValueError(f"rows must have {self.K} floats")
        check(self.lib.amdkge_session_group_set_rows(self._g, _ffi.TABLES[table], int(row0), int(v.shape[0]), _p(v)))

    def get_rows(self, table, ids=None, row0=0, nrows=None):
        """Rows of a table in GLOBAL numbering (a row-sharded group gathers entity rows from their owners)."""
        if ids is not None:
            ids = _i32(ids)
            out = np.empty((ids.shape[0], self.K), dtype=np.float32)
            check(self.lib.amdkge_session_group_get_rows(self._g, _ffi.TABLES[table], _p(ids), 0, int(ids.shape[0]), _p(out)))
            return out
        if nrows is None:
            nrows = (self.n_ents if table.startswith("ent") else self.n_rels) - int(row0)
        out = np.empty((int(nrows), self.K), dtype=np.float32)
        check(self.lib.amdkge_session_group_get_rows(self._g, _ffi.TABLES[table], None, int(row0), int(nrows), _p(out)))
        return out

    def route_overflow(self):
        f = C.c_int32(0)
        check(self.lib.amdkge_session_group_route_overflow(self._g, C.byref(f)))
        return bool(f.value)

    def train_step(self, triples, focus_w=None):
        t = _i32(triples)
        fw = np.ascontiguousarray(focus_w, dtype=np.float32) if focus_w is not None else None
        loss = C.c_double(0.0)
        check(self.lib.amdkge_session_group_train_step(self._g, _p(t), int(t.shape[0]), _p(fw), C.byref(loss)))
        return float(loss.value)

    def rank(self, triples, filters_s=None, filters_o=None, entities_subset=None, corrupt_side="s,o", ranking_strategy="worst"):
        """evaluate() through the group (amdkge_session_group_rank): arguments and result as Session.rank, ids in global numbering.
        A row-sharded group counts every query against every shard and sums the per-shard counts (the reference's partition
        loop, ScoringBasedEmbeddingModel.py:1431-1452, across GPUs); a replicated group splits the queries over its replicas."""
        t = _i32(triples)
        return _rank_call(self.lib.amdkge_session_group_rank, self._g, t, _csr_pair(filters_s), _csr_pair(filters_o), entities_subset,
                          corrupt_side, ranking_strategy)
