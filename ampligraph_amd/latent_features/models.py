"""ScoringBasedEmbeddingModel: the reference's public model class
(/root/reference/ampligraph/latent_features/models/ScoringBasedEmbeddingModel.py:47) re-hosted on the
MI355X engine.  Same constructor / compile / fit / predict / evaluate / get_embeddings surface,
argument meaning and error behaviour; the Keras/TF machinery underneath is replaced by libamdkge
(HIP kernels, C ABI) driven from plain Python.  No TensorFlow, no CPU fallback.

Intentional, documented differences: negatives come from a counter-based Philox stream
keyed by (seed, step, row) instead of TF's stateful stream; tables are always HBM-resident, so
`partitioning_k > 1` is rejected; FocusE's structural weight really decays per epoch (in the reference the
tf.function captures the first epoch's value at trace time).
"""
import json
import os

import numpy as np

from .. import _ffi
from ..datasets.filters import FilterIndex
from ..datasets.indexer import DataIndexer
from ..evaluation.metrics import hits_at_n_score, mr_score, mrr_score
from ..trainer import StepLoop
from . import loss_functions, optimizers, regularizers
from .initializers import RowSource, initialise, stream_cost

SCORING_LAYER_REGISTRY = dict(_ffi.SCORING_TYPES)  # AbstractScoringLayer.py:15 (Random is out of scope)


class History:
    """Keras-History look-alike returned by fit(): `.history[name]` lists, `.epoch` list."""

    def __init__(self):
        self.history = {}
        self.epoch = []

    def _log(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


def _load_triples(x):
    """Accept what the reference's GraphDataLoader accepts for in-memory data
    (datasets/graph_data_loader.py:129-181): ndarray, list, DataFrame, or a csv/tsv file name."""
    if isinstance(x, str):
        import pandas as pd

        x = pd.read_csv(x, sep="\t", header=None, dtype=str).values
    elif hasattr(x, "values") and not isinstance(x, np.ndarray):
        x = x.values
    x = np.asarray(x)
    if x.ndim != 2 or x.shape[1] < 3:
        raise ValueError("triples must have shape (n, 3) (or (n, >3) with numeric edge values)")
    return x


class _NoDist:
    """Stand-in that makes _dist() report a single rank (used while a rank evaluates its own share)."""

    def get_world_size(self):
        return 1


class ScoringBasedEmbeddingModel:
    def __init__(self, eta, k, scoring_type="DistMult", seed=0, max_ent_size=None, max_rel_size=None):
        if scoring_type not in SCORING_LAYER_REGISTRY:
            raise KeyError(scoring_type)  # the reference indexes SCORING_LAYER_REGISTRY directly (:146)
        self.eta = int(eta)
        self.k = int(k)
        self.scoring_type = scoring_type
        self.seed = int(seed)
        self.max_ent_size = max_ent_size
        self.max_rel_size = max_rel_size
        self.internal_k = 2 * self.k if scoring_type in ("ComplEx", "HolE", "RotatE") else self.k
        self.data_indexer = None
        self.is_fitted = False
        self.is_calibrated = False
        self.is_compiled = False
        self.use_focusE = False
        self.history = None
        self.stop_training = False
        self._engine = None
        self._loop = None
        self._spec = None            # ShardSpec in row-sharded mode
        self._sharding = "replicated"
        self._sharded_negatives = "local"
        self._dist_override = None   # tests: an object with the torch.distributed collective surface
        self._full_ent = None        # row-sharded mode: cached gathered entity table

    # ------------------------------------------------------------------------------------ config (:56-98)
    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def get_config(self):
        """The constructor arguments, like the reference's Keras config (:84-98)."""
        return {"eta": self.eta, "k": self.k, "scoring_type": self.scoring_type, "seed": self.seed,
                "max_ent_size": self.max_ent_size, "max_rel_size": self.max_rel_size}

    def get_invalid_keys(self, X, data_type="raw", **kwargs):
        """:2279-2306 -- subjects / predicates / objects of X that the model's id maps do not know."""
        return self.data_indexer.get_invalid_keys(X, data_type, **kwargs)

    def save(self, filepath, **kwargs):
        """Keras' model.save(filepath) in the reference (SavedModel directory + metadata, :1002-1044): here the same directory
        form written by ampligraph_amd.utils.save_model (flat .npz / .json)."""
        from ..utils.model_utils import save_model

        save_model(self, filepath)

    # ------------------------------------------------------------------------------------ compile
    def compile(self, optimizer="adam", loss=None, entity_relation_initializer="glorot_uniform",
                entity_relation_regularizer=None, **kwargs):
        """:1145-1152.  optimizer: name | OptimizerWrapper; loss: name | Loss; initializer: one value or
        [entity_init, relation_init]; regularizer: None | 'LP'/'l1'/'l2'/'l3' | LPRegularizer | pair.

        Extra keywords of this engine: optimizer_mode="dense" (default, the reference's semantics) | "lazy" (touched rows
        only; see amdkge_opt.lazy); deterministic=True: bitwise reproducible training (AMDKGE_TILED_DETERMINISTIC).  Multi-GPU, one process per GPU under torch.distributed:
        entity_sharding="replicated" (default: tables replicated, gradient all-reduce) | "rows" (entity table
        row-sharded over the ranks, ampligraph_amd/sharded.py; sharded_negatives="local" | "global" chooses where its corruptions
        are drawn from) | "columns" (tables that fit every GPU: rank r TRAINS on k / W units of every row and the whole batch, one
        all-reduce of the partial scores per step -- ampligraph_amd/colsharded.py.  Memory and host cost of that mode: every rank
        keeps the WHOLE tables and their whole optimizer slots next to its slice -- predict / evaluate / save_weights / callbacks
        read them -- and they are refreshed from the slices through host numpy (all_gather + column merge) after each fit(),
        before each validation pass and, when callbacks are given, before the callbacks of every epoch)."""
        optimizer_mode = kwargs.pop("optimizer_mode", "dense")
        if optimizer_mode not in ("dense", "lazy"):
            raise ValueError("optimizer_mode must be 'dense' (the reference's behaviour) or 'lazy' (touched rows only)")
        self._deterministic = bool(kwargs.pop("deterministic", False))
        self._sharding = kwargs.pop("entity_sharding", "replicated")
        self._sharded_negatives = kwargs.pop("sharded_negatives", "local")
        if self._sharding not in ("replicated", "rows", "columns"):
            raise ValueError("entity_sharding must be 'replicated', 'rows' or 'columns'")
        if self._sharded_negatives not in ("local", "global"):
            raise ValueError("sharded_negatives must be 'local' or 'global'")
        self.optimizer = optimizers.get(optimizer)
        if optimizer_mode == "lazy":
            # DEVIATION from the reference (its optimizer is dense, optimizers.py:136-168): only rows that received a
            # gradient this step are updated and regularised -- TF-Addons LazyAdam semantics (amdkge_opt.lazy)
            self.optimizer.lazy = True
        if loss is None:
            raise ValueError("compile(): a loss is required")
        self.loss = loss_functions.get(loss)
        ini = entity_relation_initializer
        self._initializers = list(ini) if isinstance(ini, (list, tuple)) else [ini, ini]
        if len(self._initializers) != 2:
            raise ValueError("entity_relation_initializer must be one value or a pair [entity, relation]")
        if self.scoring_type == "RotatE":  # :1312-1315
            r = self._initializers[1]
            assert isinstance(r, str) and r.lower() == "glorot_uniform", \
                "RotatE requires the relation embeddings to be initialised with GlorotUniform"
        reg = entity_relation_regularizer
        regs = list(reg) if isinstance(reg, (list, tuple)) else [reg, reg]
        if len(regs) != 2:   # EmbeddingLookupLayer.py:147-150
            raise AssertionError("Incorrect length for regularizer. Expected 2, got {}".format(len(regs)))
        # independent per table, either may be None, each a sum of at most two LP terms (EmbeddingLookupLayer.py:131-155)
        self._regularizers = [regularizers.get(r) for r in regs]
        self.is_compiled = True

    def _assert_compile_was_called(self):
        if not self.is_compiled:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")

    # ------------------------------------------------------------------------------------ engine
    EVAL_CHUNK_SHARDED = 4096   # test triples per sharded evaluation / prediction pass (2 scratch rows each)

    def _build(self, n_ents, n_rels, batch_size=None, tables=None):
        import torch

        from ..engine import KgeEngine  # raises loudly without the HIP library / a GPU

        self.max_ent_size, self.max_rel_size = int(n_ents), int(n_rels)
        self._n_ents, self._n_rels = int(n_ents), int(n_rels)
        self._filter_cache = (None, None)   # a new id map invalidates cached filter indexes
        self._loop = None                   # the step loop is bound to the engine built below
        K = self.internal_k
        if tables is None:
            cost = stream_cost(self._initializers[0], (n_ents, K))
            if cost is not None:   # a rank can draw just ITS rows of the whole-table stream (initializers.RowSource)
                ent_rows = RowSource(self._initializers[0], (n_ents, K), self.seed, 0).rows
                rel = RowSource(self._initializers[1], (n_rels, K), self.seed, cost).rows(0, n_rels)
            else:
                rng = np.random.Generator(np.random.PCG64(self.seed))
                ent = initialise(self._initializers[0], (n_ents, K), rng)
                rel = initialise(self._initializers[1], (n_rels, K), rng)
                ent_rows = lambda lo, hi: ent[lo:hi]   # noqa: E731
        else:
            ent, rel = tables
            ent_rows = ent if callable(ent) else (lambda lo, hi: ent[lo:hi])
        d = self._dist()
        self._spec = None
        lo, hi = 0, int(n_ents)
        if self._sharding == "rows" and d is not None:
            from ..sharded import ShardedStepLoop, ShardSpec

            # same initial values as one GPU: rows [lo, hi) of the whole-table draw
            sp = self._spec = ShardSpec(n_ents, d.get_world_size(), d.get_rank())
            per_rank = -(-int(batch_size or 1000) // sp.world)
            cap = max(ShardedStepLoop.rows_needed(per_rank, self.eta, self._sharded_negatives, sp.world, n_ents),
                      2 * self.EVAL_CHUNK_SHARDED)
            self._engine = KgeEngine(self.scoring_type, self.k, sp.n_local + cap, n_rels, max_rel_size=n_rels)
            lo, hi = sp.lo, sp.hi
        else:
            self._engine = KgeEngine(self.scoring_type, self.k, n_ents, n_rels, max_rel_size=n_rels)
        self._upload_rows(self._engine.ent, ent_rows, lo, hi)
        self._engine.pack(rel, out=self._engine.rel)
        self._full_ent = None
        self._col_engine = None
        if self._sharding == "columns" and d is not None:
            from ..colsharded import check_columns

            W = d.get_world_size()
            check_columns(self.scoring_type, self.k, W, lambda kk: int(self._engine.lib.amdkge_padded_k(kk)))
            # the slice this rank trains on; filled from the whole tables (and optimizer slots) when fit() starts
            self._col_engine = KgeEngine(self.scoring_type, self.k // W, n_ents, n_rels, max_rel_size=n_rels, k_full=self.k)
            self._cols_pull = True

    def _upload_rows(self, dst, rows, lo, hi, chunk_elems=1 << 24):
        """dst[0 : hi-lo] <- rows(lo, hi) (dense rows, packed into the engine's stored layout), in host chunks of
        <= chunk_elems floats (C5 shards do not fit host RAM twice)."""
        eng = self._engine
        step = max(1, chunk_elems // eng.K)
        for r0 in range(int(lo), int(hi), step):
            r1 = min(int(hi), r0 + step)
            blk = np.ascontiguousarray(rows(r0, r1), dtype=np.float32)
            if blk.shape != (r1 - r0, eng.K):
                raise ValueError(f"table rows have shape {blk.shape}, expected {(r1 - r0, eng.K)}")
            eng.pack(blk, out=dst[r0 - lo:r1 - lo])

    def _ensure_shard_capacity(self, batch_size):
        """Row-sharded mode, continued training: the scratch rows behind the shard were sized for the first fit()'s
        batch (or for the default batch by load_weights); a larger batch gets a larger engine, tables and optimizer
        state carried over."""
        if self._spec is None:
            return
        from ..engine import KgeEngine
        from ..sharded import ShardedStepLoop

        sp, old = self._spec, self._engine
        per_rank = -(-int(batch_size) // sp.world)
        need = max(ShardedStepLoop.rows_needed(per_rank, self.eta, self._sharded_negatives, sp.world, sp.n_ents),
                   2 * self.EVAL_CHUNK_SHARDED)
        if need <= int(old.ent.shape[0]) - sp.n_local:
            return
        new = KgeEngine(self.scoring_type, self.k, sp.n_local + need, self._n_rels, max_rel_size=self._n_rels)
        new.ent[:sp.n_local].copy_(old.ent[:sp.n_local])
        new.rel.copy_(old.rel)
        old_slots = dict(getattr(old, "slots", {}))
        self._engine = new
        self._loop = self._make_loop() if self.is_compiled else None   # allocates fresh slots on the new engine
        for name, t in old_slots.items():
            if name in new.slots:
                (new.slots[name][:sp.n_local] if name.endswith("_e") else new.slots[name]).copy_(
                    t[:sp.n_local] if name.endswith("_e") else t)
        self._full_ent = None

    def _dist(self):
        if self._dist_override is not None:
            return self._dist_override if self._dist_override.get_world_size() > 1 else None
        import torch.distributed as dist

        return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None

    # ---- column-sharded training (entity_sharding="columns"): whole tables <-> this rank's slice ----
    def _cols_pull_now(self):
        """whole tables (+ optimizer slots) of self._engine -> this rank's column slice (local)."""
        from ..colsharded import column_slice

        d, col, eng = self._dist(), self._col_engine, self._engine
        W, r = d.get_world_size(), d.get_rank()
        ent, rel = eng.get_tables()
        col.set_tables(column_slice(ent, self.scoring_type, self.k, W, r), column_slice(rel, self.scoring_type, self.k, W, r))
        for name, t in getattr(col, "slots", {}).items():
            if name in getattr(eng, "slots", {}):
                col.pack(column_slice(eng.unpack(eng.slots[name]).cpu().numpy(), self.scoring_type, self.k, W, r), out=t)
        self._cols_pull = False

    def _cols_push(self):
        """the ranks' column slices (+ optimizer slots) -> the whole tables every rank keeps (collective)."""
        import torch

        from ..colsharded import column_merge

        if getattr(self, "_col_engine", None) is None:
            return
        d, col, eng = self._dist(), self._col_engine, self._engine
        W = d.get_world_size()

        def gathered(stored):
            mine = col.unpack(stored).contiguous()
            parts = [torch.empty_like(mine) for _ in range(W)]
            d.all_gather(parts, mine)
            return column_merge([p_.cpu().numpy() for p_ in parts], self.scoring_type)

        eng.set_tables(gathered(col.ent), gathered(col.rel))
        for name, t in getattr(col, "slots", {}).items():
            if name in getattr(eng, "slots", {}):
                eng.pack(gathered(t), out=eng.slots[name])
        self._full_ent = None

    def _make_loop(self):
        reg = self._regularizers[0]
        if getattr(self, "_col_engine", None) is not None:
            from ..colsharded import ColumnStepLoop

            if getattr(self, "_deterministic", False):
                raise ValueError("entity_sharding='columns': deterministic mode is not offered")
            loop = ColumnStepLoop(self._col_engine, self.eta, self.loss, self.optimizer, reg, self.seed, self._dist())
            loop.reg_rel = self._regularizers[1]
            self._engine.prepare_training(self.optimizer.name)   # the whole tables' optimizer slots: what checkpoints hold
            return loop
        if self._spec is not None:
            from ..sharded import ShardedStepLoop

            loop = ShardedStepLoop(self._engine, self._spec, self.eta, self.loss, self.optimizer, reg, self.seed,
                                   self._dist(), negatives=self._sharded_negatives)
        else:
            loop = StepLoop(self._engine, self.eta, self.loss, self.optimizer, reg, self.seed, self._dist())
        loop.reg_rel = self._regularizers[1]   # the relation table's own regulariser (or None)
        if getattr(self, "_deterministic", False):
            loop.deterministic = True   # AMDKGE_TILED_DETERMINISTIC: bitwise reproducible tables (include/amdkge.h)
        return loop

    def _entity_table(self):
        """(N, Ks) entity table in the engine's STORED layout as a device tensor (row-sharded mode: gathered once and
        cached until the next fit); engine.unpack() gives dense rows."""
        if self._spec is None:
            return self._engine.ent
        if self._full_ent is None:
            import torch

            sp, eng = self._spec, self._engine
            mine = torch.zeros(sp.rows_per, eng.Ks, dtype=eng.ent.dtype, device=eng.ent.device)
            mine[:sp.n_local] = eng.ent[:sp.n_local]
            parts = [torch.empty_like(mine) for _ in range(sp.world)]
            self._dist().all_gather(parts, mine)
            self._full_ent = torch.cat(parts)[:sp.n_ents]
        return self._full_ent

    def _localise(self, Xd):
        """Row-sharded mode: fetch the remote s/o rows of the int32 device triples Xd behind the shard and return
        the triples re-indexed into the local index space."""
        import torch

        from ..sharded import RowExchange

        sp, eng = self._spec, self._engine
        x = Xd.to(torch.int64)
        n = int(x.shape[0])
        ids = torch.cat([x[:, 0], x[:, 2]])
        remote = (ids < sp.lo) | (ids >= sp.hi)
        rid, rinv = torch.unique(ids[remote], return_inverse=True)
        ex = RowExchange(sp, self._dist(), rid)
        ex.fetch(eng.ent, sp.n_local)
        loc = ids - sp.lo
        loc[remote] = sp.n_local + ex.slots()[rinv]
        return torch.stack([loc[:n], x[:, 1], loc[n:]], 1).to(torch.int32).contiguous()

    # ------------------------------------------------------------------------------------ fit
    def fit(self, x=None, batch_size=1000, epochs=100, verbose=True, callbacks=None, validation_split=0.0,
            validation_data=None, shuffle=True, initial_epoch=0, validation_batch_size=10,
            validation_corrupt_side="s,o", validation_freq=10, validation_burn_in=0, validation_filter=False,
            validation_entities_subset=None, partitioning_k=1, focusE=False, focusE_params={}):
        """:544-883.  Batches are sequential, un-shuffled slices of `x` with a short last batch
        (graph_data_loader.py:472-523); `shuffle` is accepted and, like in the reference (:553), unused."""
        import torch

        self._assert_compile_was_called()
        if partitioning_k != 1:
            raise NotImplementedError("partitioning_k > 1: tables are HBM-resident on MI355X, graph partitioning "
                                      "with disk swapping is out of scope (SURVEY.md section 2, rows 15-16)")
        X = _load_triples(x)
        if validation_split:   # :719-728: carve the validation set out of x without creating unseen entities
            assert isinstance(x, np.ndarray), "Validation split supported for numpy arrays only!"
            from ..evaluation.protocol import train_test_split_no_unseen

            X, validation_data = train_test_split_no_unseen(X, test_size=validation_split, seed=self.seed,
                                                            allow_duplication=False)
        # FocusE (:714-790): active only when asked for AND the data carries numeric columns behind s, p, o
        self.use_focusE = bool(focusE) and X.shape[1] > 3
        focus_w = None
        if self.use_focusE:
            assert isinstance(focusE_params, dict), "focusE parameters need to be in a dict!"
            nl = focusE_params.get("non_linearity", "linear")
            if nl not in _ffi.FOCUS_NONLINEARITY:
                raise ValueError("Invalid focusE non-linearity")
            stop_epoch = focusE_params.get("stop_epoch", 251)
            assert stop_epoch >= 0, "Invalid value for focusE stop_epoch: expected a value >=0 but got {}".format(stop_epoch)
            structural_wt = focusE_params.get("structural_wt", 0.001)
            assert 0 <= structural_wt <= 1, "Invalid focusE 'structural_wt' passed! It has to belong to [0,1]."
            self.focusE_params = {"non_linearity": nl, "stop_epoch": stop_epoch, "structural_wt": structural_wt}
            focus_w = np.ascontiguousarray(X[:, 3:].astype(np.float32).mean(axis=1), dtype=np.float32)   # :360
        elif X.shape[1] > 3:
            print("Data shape is {}: not only triples were given, but focusE is not active!".format(X.shape[1]))
        if self.data_indexer is None or not self.is_fitted:
            self.data_indexer = DataIndexer(X)
            Xi = self.data_indexer.get_indexes(X[:, :3])
            self._build(self.data_indexer.get_entities_count(), self.data_indexer.get_relations_count(), batch_size)
        else:  # continue training (initial_epoch > 0): same id map, same tables
            Xi = self.data_indexer.get_indexes(X[:, :3])
            self._ensure_shard_capacity(batch_size)
        eng = self._engine
        if self._loop is None:
            self._loop = self._make_loop()
        loop = self._loop
        if getattr(self, "_col_engine", None) is not None:
            if self.use_focusE:
                raise ValueError("entity_sharding='columns': FocusE is not offered")
            self._cols_pull_now()   # (a fresh model, a checkpoint just loaded, or the tables the previous fit() left)
        # negatives are keyed by a step counter that continues where the optimizer's iteration count stands: a run resumed
        # from a checkpoint draws what the uninterrupted run would have drawn, and a second fit() draws fresh negatives
        rng_base = int(self.optimizer.iterations)
        self._full_ent = None
        if hasattr(loop, "configure_for_data"):
            loop.configure_for_data(Xi, batch_size)
        train = torch.as_tensor(np.ascontiguousarray(Xi, dtype=np.int32)).to(eng.device)
        n = int(train.shape[0])
        focus_dev = None
        if focus_w is not None:
            if focus_w.shape[0] != n:
                raise ValueError("FocusE: rows with unknown entities/relations were dropped; weights no longer align")
            focus_dev = torch.as_tensor(focus_w).to(eng.device)
        batch_size = int(batch_size)
        steps = (n + batch_size - 1) // batch_size
        self.history = History()
        cbs = list(callbacks or [])
        for cb in cbs:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin()
        self.stop_training = False
        if isinstance(validation_entities_subset, str) and validation_entities_subset == "all":
            validation_entities_subset = None
        # column-sharded training: are the whole tables (what evaluate / callbacks / checkpoints read) current with the slices?
        cols_current = False
        for epoch in range(int(initial_epoch), int(epochs)):
            self.current_epoch = epoch
            loop.reset_loss()
            focus = None
            if focus_dev is not None:   # update_focusE_params (:536-542): linear decay of the structural weight
                fp = self.focusE_params
                if fp["stop_epoch"] > 0:
                    fp["structural_wt"] = max(1.0 - epoch / fp["stop_epoch"], 0.001)
            first = 0
            if (epoch == int(initial_epoch) and getattr(loop, "auto_tune", False) and getattr(loop, "world", 1) > 1
                    and focus_dev is None and steps >= 24):
                # AMDKGE_DP_MERGE=auto: the first steps of this epoch are run under each gradient-merge schedule and the
                # fastest is kept (StepLoop.tune_merge; every schedule computes the same update).  The epoch's reported
                # loss then covers the remaining steps only.
                base = rng_base
                first = loop.tune_merge(lambda s_: train[(s_ - base) * batch_size:(s_ - base + 1) * batch_size], base)
                loop.auto_tune = False
            for step in range(first, steps):
                b0 = step * batch_size
                if focus_dev is not None:
                    focus = (focus_dev[b0:b0 + batch_size], self.focusE_params["structural_wt"], self.focusE_params["non_linearity"])
                    loop.step(train[b0:b0 + batch_size], rng_base + (epoch - int(initial_epoch)) * steps + step, focus)
                else:
                    loop.step(train[b0:b0 + batch_size], rng_base + (epoch - int(initial_epoch)) * steps + step)
            cols_current = cols_current and steps <= first
            logs = {"loss": loop.mean_batch_loss()}
            validate = (epoch >= (validation_burn_in - 1) and validation_data is not None
                        and (epoch + 1) % int(validation_freq) == 0)
            if validate:
                self.is_fitted = True
                self._cols_push()   # (column-sharded training: refresh the whole tables the evaluation reads)
                cols_current = True
                ranks = self.evaluate(validation_data, batch_size=validation_batch_size or batch_size,
                                      use_filter=validation_filter, dataset_type="valid",
                                      corrupt_side=validation_corrupt_side,
                                      entities_subset=validation_entities_subset, verbose=False)
                logs.update({"val_mrr": mrr_score(ranks), "val_mr": mr_score(ranks),
                             "val_hits@1": hits_at_n_score(ranks, 1), "val_hits@10": hits_at_n_score(ranks, 10),
                             "val_hits@100": hits_at_n_score(ranks, 100)})  # :1857-1863
            self.history._log(epoch, logs)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
            if cbs and not cols_current:
                # callbacks read AND write the whole tables (EarlyStopping snapshots them, and restores the best ones just before it
                # stops the run): they must see this epoch's, and a push after them would overwrite what they restored
                self._cols_push()
                cols_current = True
            for cb in cbs:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        if not cols_current:
            self._cols_push()
        for cb in cbs:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end()
        self.is_fitted = True
        return self.history

    # ------------------------------------------------------------------------------------ filters
    def _filter_index(self, use_filter, Xi):
        """FilterIndex for evaluate(): the union of the given datasets (dict) or the evaluated data itself (True), all
        indexed with the training id map (graph_data_loader.py:184-190,652-653).  The id mapping is host work (labels), the
        index itself (sort + CSR) is built on the device (amdkge_filter_build); the last one is cached by content checksum, so
        validation during fit() and repeated evaluate() calls reuse it."""
        try:   # (content checksum of the filter arrays: xxh3 runs at ~10 GB/s, zlib's crc32 at ~2)
            from xxhash import xxh3_64_intdigest as _digest
        except Exception:   # pragma: no cover
            from zlib import crc32 as _digest

        if isinstance(use_filter, dict):
            arrays = [_load_triples(v)[:, :3] for v in use_filter.values()]
            key = []
            for a in arrays:
                a = np.ascontiguousarray(a)
                if a.dtype == object:
                    key = None
                    break
                key.append((a.shape, str(a.dtype), _digest(a.view(np.uint8).reshape(-1))))
            key = None if key is None else ("dict", tuple(key), self._n_ents, self._n_rels)
            if key is not None and getattr(self, "_filter_cache", (None, None))[0] == key:
                return self._filter_cache[1]
            # (id mapping on the host -- labels are host data --, then the index itself is built on the device: kge_filter.hip)
            fi = FilterIndex([self.data_indexer.get_indexes(a) for a in arrays], self._n_ents, self._n_rels, engine=self._engine)
            if key is not None:
                self._filter_cache = (key, fi)
            return fi
        if use_filter:
            return FilterIndex([Xi], self._n_ents, self._n_rels, engine=self._engine)
        return None

    # ------------------------------------------------------------------------------------ predict
    def _index_test(self, x):
        assert self.is_fitted, "Model is not fit on the data yet!"
        X = _load_triples(x)
        return np.ascontiguousarray(self.data_indexer.get_indexes(X[:, :3]), dtype=np.int32)

    def predict(self, x, batch_size=32, verbose=0, callbacks=None):
        """:1736-1823: float32 (n,) scores in input order (rows with unknown keys dropped)."""
        import torch

        Xi = self._index_test(x)
        if Xi.shape[0] == 0:
            return np.zeros(0, dtype=np.float32)
        return self._score_dev(torch.as_tensor(Xi).to(self._engine.device)).cpu().numpy()

    def _score_dev(self, Xd):
        """Scores of the int32 device triples Xd (global ids) as a device tensor."""
        import torch

        if self._spec is None:
            return self._engine.score(Xd)
        outs = []   # row-sharded: every rank fetches the rows it lacks and scores all triples (replicated result)
        for c0 in range(0, int(Xd.shape[0]), self.EVAL_CHUNK_SHARDED):
            outs.append(self._engine.score(self._localise(Xd[c0:c0 + self.EVAL_CHUNK_SHARDED])))
        return torch.cat(outs) if outs else torch.zeros(0, dtype=torch.float32, device=Xd.device)

    # ------------------------------------------------------------------------------------ evaluate
    def evaluate(self, x=None, batch_size=10, verbose=True, use_filter=False, corrupt_side="s,o",
                 entities_subset=None, ranking_strategy="worst", callbacks=None, dataset_type="test"):
        """:1516-1692: int32 ranks (n, 1|2), 1-based, reference tie/filter semantics.  `batch_size` (the reference's default, 10,
        :1519) only chunked the TF graph in the reference; results do not depend on it and it is ignored here."""
        import torch

        assert corrupt_side in ["s", "o", "s,o", "s+o"], "Invalid value for corrupt_side"
        assert ranking_strategy in ["best", "middle", "worst"], "Invalid value for ranking_strategy"
        Xi = self._index_test(x)
        eng = self._engine
        n = Xi.shape[0]
        sides = [sd for sd in ("s", "o") if sd in corrupt_side]
        if n == 0:
            return np.zeros((0, 1 if corrupt_side in ("s", "o", "s+o") else 2), dtype=np.int32)
        if self._spec is not None:
            return self._evaluate_sharded(Xi, sides, use_filter, corrupt_side, entities_subset, ranking_strategy)
        d = self._dist()
        if d is not None and n >= d.get_world_size():
            # replicated tables, several ranks: every rank ranks its contiguous share of the test triples (all filters
            # are built from the full data, so the ranks are those of the single-GPU run) and the shares are gathered
            return self._evaluate_split_queries(d, x, Xi, use_filter, corrupt_side, entities_subset, ranking_strategy)
        # filters (graph_data_loader.py:184-190,652-653): True -> the evaluated data filters itself;
        # dict -> union of the given datasets, all indexed with the training id map
        fi = self._filter_index(use_filter, Xi)
        dev = eng.device
        ent_ids = subset_pos = None
        if entities_subset is not None and len(entities_subset) > 0:
            sub = np.asarray(self.data_indexer.get_indexes(np.asarray(entities_subset), "e"), dtype=np.int32)
            pos = np.full(eng.n_ents, -1, dtype=np.int32)
            pos[sub] = np.arange(sub.shape[0], dtype=np.int32)  # DenseHashTable.insert: last wins (:1639-1643)
            ent_ids, subset_pos = torch.as_tensor(sub).to(dev), torch.as_tensor(pos).to(dev)
        Xd = torch.as_tensor(Xi).to(dev)
        ranks = torch.empty(n, len(sides), dtype=torch.int32, device=dev)
        CH = 1 << 16
        for c0 in range(0, n, CH):
            xs = Xd[c0:c0 + CH]
            jobs = []
            for col, sd in enumerate(sides):
                flt = fi.device_filter(eng, xs, sd) if fi is not None else None   # range lookup on the device
                jobs.append((_ffi.SIDE_S if sd == "s" else _ffi.SIDE_O, flt, ranks[c0:c0 + CH, col], len(sides)))
            eng.rank_sides(xs, jobs, ranking_strategy, ent_ids, subset_pos)   # the sides run beside each other
        r = ranks.cpu().numpy()
        if corrupt_side == "s+o":  # :1459-1463 sums the two 0-based sides, then +1 (:1684)
            r = (r.sum(1, keepdims=True) - 1).astype(np.int32)
        return r

    def _evaluate_split_queries(self, d, x, Xi, use_filter, corrupt_side, entities_subset, ranking_strategy):
        import torch

        from ..trainer import shard_bounds

        W, r = d.get_world_size(), d.get_rank()
        n = Xi.shape[0]
        lo, hi = shard_bounds(n, W, r)
        flt = use_filter
        if use_filter is True:   # "the evaluated data filters itself": keep filtering with the WHOLE evaluated set
            flt = {"self": self.data_indexer.get_indexes(Xi, "t", "ind2raw")}
        override, self._dist_override = self._dist_override, _NoDist()
        try:
            mine = self.evaluate(self.data_indexer.get_indexes(Xi[lo:hi], "t", "ind2raw"), use_filter=flt,
                                 corrupt_side=corrupt_side, entities_subset=entities_subset,
                                 ranking_strategy=ranking_strategy, verbose=False)
        finally:
            self._dist_override = override
        cols = mine.shape[1]
        width = -(-n // W)
        buf = torch.zeros(width, cols, dtype=torch.int32, device=self._engine.device)
        buf[:hi - lo] = torch.as_tensor(mine).to(buf.device)
        parts = [torch.empty_like(buf) for _ in range(W)]
        d.all_gather(parts, buf)
        out = [parts[q][:shard_bounds(n, W, q)[1] - shard_bounds(n, W, q)[0]] for q in range(W)]
        return torch.cat(out).cpu().numpy()

    def _evaluate_sharded(self, Xi, sides, use_filter, corrupt_side, entities_subset, ranking_strategy):
        """Row-sharded evaluate(): every rank counts against ITS rows, counts are summed over ranks
        (the reference's loop over entity partitions, :1431-1452, across GPUs); identical result on every rank."""
        import torch

        from ..sharded import sharded_rank_counts

        eng, d = self._engine, self._dist()
        subset = None
        if entities_subset is not None and len(entities_subset) > 0:
            sub = np.asarray(self.data_indexer.get_indexes(np.asarray(entities_subset), "e"), dtype=np.int64)
            subset = self._spec.local_subset(torch.as_tensor(sub).to(eng.device))
        fi = self._filter_index(use_filter, Xi)
        dev = eng.device
        n = Xi.shape[0]
        ranks = torch.empty(n, len(sides), dtype=torch.int32, device=dev)
        CH = self.EVAL_CHUNK_SHARDED
        Xd = torch.as_tensor(Xi).to(dev)
        for c0 in range(0, n, CH):
            for col, sd in enumerate(sides):
                flt = fi.device_filter(eng, Xd[c0:c0 + CH], sd) if fi is not None else None
                counts, sub = sharded_rank_counts(eng, self._spec, d, Xd[c0:c0 + CH],
                                                  _ffi.SIDE_S if sd == "s" else _ffi.SIDE_O, flt, subset)
                eng.compose_ranks(counts, sub, ranking_strategy, out=ranks[c0:c0 + CH, col], out_stride=len(sides))
        r = ranks.cpu().numpy()
        if corrupt_side == "s+o":
            r = (r.sum(1, keepdims=True) - 1).astype(np.int32)
        return r

    # ------------------------------------------------------------------------------------ accessors
    def is_fit(self):
        return self.is_fitted

    def get_indexes(self, X, type_of="t", order="raw2ind"):
        return self.data_indexer.get_indexes(X, type_of, order)

    def get_count(self, concept_type="e"):
        assert self.is_fitted, "Model is not fit on the data yet!"
        if concept_type == "e":
            return self.data_indexer.get_entities_count()
        if concept_type == "r":
            return self.data_indexer.get_relations_count()
        raise ValueError("Invalid Concept Type (expected 'e' or 'r')")

    def get_train_embedding_matrix_size(self):
        assert self.is_fitted, "Model is not fit on the data yet!"
        return {"e": (self._n_ents, self.internal_k), "r": (self._n_rels, self.internal_k)}

    def get_embeddings(self, entities, embedding_type="e"):
        """:2214-2277."""
        if embedding_type not in ("e", "r"):
            raise ValueError("Invalid entity type: {}".format(embedding_type))
        assert self.is_fitted, "Model is not fit on the data yet!"
        idx = np.asarray(self.data_indexer.get_indexes(np.asarray(entities), embedding_type), dtype=np.int64)
        import torch

        tab = self._entity_table() if embedding_type == "e" else self._engine.rel
        return self._engine.unpack(tab[torch.as_tensor(idx).to(tab.device)]).cpu().numpy()

    # ------------------------------------------------------------------------------------ calibration
    def calibrate(self, X_pos, X_neg=None, positive_base_rate=None, batch_size=32, epochs=50, verbose=0):
        """:1922-2122 -- Platt scaling of the scores (CalibrationLayer, layers/calibration/calibrate.py:11-129): two
        scalars (w, b) fitted with Adam (Keras defaults) on sigmoid cross-entropy, one update per batch of positives.
        With X_neg the negatives are iterated in lock step (their batch size adjusted so that both have the same
        number of batches, :2033-2052); without, one corruption per positive is generated per step
        (corruption_layer(inputs, num_ents, 1), :1886) -- here from the Philox stream keyed by (seed, step).
        Scores, corruptions and the objective/gradient reduction are libamdkge kernels; the scalar Adam runs on host."""
        import math

        import torch

        assert self.is_fitted, "Model is not fit on the data yet!"
        self.is_calibrated = False
        eng = self._engine
        Xp = self._index_test(X_pos)
        pos_size = int(Xp.shape[0])
        neg_size = pos_size
        batch_size = int(batch_size)
        Xn = None
        if X_neg is None:
            assert positive_base_rate is not None, "Please provide the negatives or positive base rate!"
        else:
            Xn = self._index_test(X_neg)
            neg_size = int(Xn.shape[0])
            if positive_base_rate is None:
                positive_base_rate = pos_size / (pos_size + neg_size)
        if positive_base_rate is not None and (positive_base_rate <= 0 or positive_base_rate >= 1):
            raise ValueError("positive_base_rate must be a value between 0 and 1.")
        n_batches = max(1, -(-pos_size // batch_size))
        bs_neg = batch_size
        if Xn is not None and -(-neg_size // batch_size) != n_batches:
            bs_neg = -(-neg_size // n_batches)
        # CalibrationLayer.__init__ / call constants
        w, b = 0.0, float(np.float32(math.log((neg_size + 1.0) / (pos_size + 1.0))))
        label_pos, label_neg = (pos_size + 1.0) / (pos_size + 2.0), 1.0 / (neg_size + 2.0)
        weight_neg = (1.0 - positive_base_rate) / positive_base_rate
        dev = eng.device
        Xpd = torch.as_tensor(Xp).to(dev)
        # row-sharded tables: the same code, scores come from _score_dev (replicated result on every rank)
        sp_all = self._score_dev(Xpd) if pos_size else None    # the embeddings are frozen: score once
        sn_all = self._score_dev(torch.as_tensor(Xn).to(dev)) if Xn is not None and neg_size else None
        slots = [(0.0, 0.0), (0.0, 0.0)]
        t = 0
        for epoch in range(int(epochs)):
            for bi in range(n_batches):
                sp = sp_all[bi * batch_size:(bi + 1) * batch_size]
                if sn_all is not None:
                    sn = sn_all[bi * bs_neg:(bi + 1) * bs_neg]
                else:
                    neg = eng.sample_corruptions(Xpd[bi * batch_size:(bi + 1) * batch_size], 1, self.seed, t,
                                                 sample_range=self._n_ents)
                    sn = self._score_dev(neg)
                if sp.shape[0] == 0:
                    continue
                t += 1
                loss, gw, gb = eng.platt_step(sp, sn, w, b, label_pos, label_neg, sn.shape[0] / sp.shape[0], weight_neg)
                alpha = 0.001 * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)   # tf.keras.optimizers.Adam() defaults
                upd = []
                for i, (p_, g_) in enumerate(((w, gw), (b, gb))):
                    m, v = slots[i]
                    m += (g_ - m) * (1.0 - 0.9)
                    v += (g_ * g_ - v) * (1.0 - 0.999)
                    slots[i] = (m, v)
                    upd.append(p_ - m * alpha / (math.sqrt(v) + 1e-7))
                w, b = upd
            if verbose:
                print(f"calibration epoch {epoch + 1}/{epochs} - loss: {loss:.6f}")
        self.calibration_parameters = {"calib_w": float(w), "calib_b": float(b), "pos_size": pos_size,
                                       "neg_size": neg_size, "positive_base_rate": float(positive_base_rate)}
        self.is_calibrated = True

    def predict_proba(self, x, batch_size=32, verbose=0, callbacks=None):
        """:2124-2212: sigmoid(-(w * score + b)) of the calibrated model (CalibrationLayer.call(training=0))."""
        if not self.is_calibrated:
            raise RuntimeError("Model has not been calibrated. Please call `model.calibrate(...)` before predicting probabilities.")
        s = self.predict(x, batch_size=batch_size).astype(np.float32)
        w = np.float32(self.calibration_parameters["calib_w"])
        b = np.float32(self.calibration_parameters["calib_b"])
        return (1.0 / (1.0 + np.exp((w * s + b).astype(np.float64)))).astype(np.float32)

    # ------------------------------------------------------------------------------------ persistence
    @staticmethod
    def _shard_file(filepath, rank, world):
        return "{}.shard{:03d}-of-{:03d}.npz".format(filepath, rank, world)

    def save_weights(self, filepath):
        """Own flat format (the reference writes a TF checkpoint, :1046-1071): <filepath>.npz holds tables,
        optimizer slots and id maps; <filepath>.json the hyper-parameters.

        Row-sharded mode (collective: call on every rank): rank r writes ITS rows of the entity table and of the
        optimizer slots to <filepath>.shardRRR-of-WWW.npz; rank 0 writes the replicated part (relation table + slots,
        id maps, `shard_world`) to <filepath>.npz and the .json.  Nothing is gathered, so it works at C5 scale;
        load_weights() re-slices, so a checkpoint can be resumed on a different number of GPUs (or on one)."""
        assert self.is_fitted, "Model is not fit on the data yet!"
        eng = self._engine
        arrays = {}
        if self._spec is not None:
            sp = self._spec
            dense = lambda t: eng.unpack(t).cpu().numpy()   # noqa: E731  checkpoints hold dense rows, whatever the engine stores
            mine = {"ent": dense(eng.ent[:sp.n_local]), "lo": np.int64(sp.lo), "hi": np.int64(sp.hi)}
            for kname, t in getattr(eng, "slots", {}).items():
                if kname.endswith("_e"):
                    mine["slot_" + kname] = dense(t[:sp.n_local])
            np.savez(self._shard_file(filepath, sp.rank, sp.world), **mine)
            arrays = {"rel": dense(eng.rel), "shard_world": np.int64(sp.world), "n_ents": np.int64(sp.n_ents)}
            for kname, t in getattr(eng, "slots", {}).items():
                if kname.endswith("_r"):
                    arrays["slot_" + kname] = dense(t)
            self._dist().barrier()
            if sp.rank != 0:
                return
        else:
            if self._loop is not None and hasattr(self._loop, "sync_optimizer_slots"):
                self._loop.sync_optimizer_slots()   # data-parallel sharded merge: collective, call on every rank
            d = self._dist()
            if d is not None and d.get_rank() != 0:
                d.barrier()   # replicated tables: every rank holds the same bytes, rank 0 writes them (barrier: file complete)
                return
            ent, rel = eng.get_tables()
            arrays = {"ent": ent, "rel": rel}
            for kname, t in getattr(eng, "slots", {}).items():
                arrays["slot_" + kname] = eng.unpack(t).cpu().numpy()
        st = self.data_indexer.state()
        arrays["ent_raw"], arrays["rel_raw"] = st["ent_raw"], st["rel_raw"]
        # written under a temporary name and renamed: a reader never sees a torn archive
        with open(filepath + ".npz.tmp", "wb") as f:
            np.savez(f, **arrays)
        os.replace(filepath + ".npz.tmp", filepath + ".npz")
        meta = {"eta": self.eta, "k": self.k, "scoring_type": self.scoring_type, "seed": self.seed,
                "optimizer": self.optimizer.get_config() if self.is_compiled else None,
                "iterations": self.optimizer.iterations if self.is_compiled else 0,
                "loss": {"name": self.loss.name, "params": self.loss._loss_parameters} if self.is_compiled else None,
                "calibration": self.calibration_parameters if self.is_calibrated else None,
                "regularizer": ([None if r is None else {"p": r.p, "lambda": r.lam, "p2": r.p2, "lambda2": r.lam2} for r in self._regularizers]
                                if self.is_compiled else None)}
        with open(filepath + ".json.tmp", "w") as f:
            json.dump(meta, f)
        os.replace(filepath + ".json.tmp", filepath + ".json")
        if self._spec is None and self._dist() is not None:
            self._dist().barrier()

    def load_weights(self, filepath):
        """Whole-table or row-sharded checkpoint -> this model, whatever its own sharding (rows are re-sliced)."""
        import torch

        from ..sharded import ShardSpec

        z = np.load(filepath + ".npz", allow_pickle=False)
        self.data_indexer = DataIndexer.from_state({"ent_raw": z["ent_raw"], "rel_raw": z["rel_raw"]})
        if not self.is_compiled:
            self._initializers = ["zeros", "zeros"]
        saved_world = int(z["shard_world"]) if "shard_world" in z.files else 0
        n_ents = int(z["n_ents"]) if saved_world else int(z["ent"].shape[0])

        def rows_of(key):
            if not saved_world:
                arr = z[key]
                return lambda lo, hi: arr[lo:hi]
            shards = {}

            def rows(lo, hi):
                out = []
                for r in range(saved_world):
                    s = ShardSpec(n_ents, saved_world, r)
                    if s.hi <= lo or s.lo >= hi or s.n_local == 0:
                        continue
                    if r not in shards:
                        shards.clear()   # one shard file in memory at a time
                        shards[r] = np.load(self._shard_file(filepath, r, saved_world), allow_pickle=False)[key]
                    out.append(shards[r][max(lo, s.lo) - s.lo:min(hi, s.hi) - s.lo])
                return np.concatenate(out) if out else np.zeros((0, self.internal_k), np.float32)
            return rows

        self._build(n_ents, z["rel"].shape[0], tables=(rows_of("ent"), z["rel"]))
        slot_keys = [k[5:] for k in z.files if k.startswith("slot_")]
        if saved_world:
            s0 = np.load(self._shard_file(filepath, 0, saved_world), allow_pickle=False)
            slot_keys += [k[5:] for k in s0.files if k.startswith("slot_")]
        if slot_keys and self.is_compiled:
            self._loop = self._make_loop()
            eng = self._engine
            lo, hi = (self._spec.lo, self._spec.hi) if self._spec is not None else (0, n_ents)
            for kname in slot_keys:
                if kname not in eng.slots:
                    continue   # checkpoint of another optimizer
                if kname.endswith("_e"):
                    self._upload_rows(eng.slots[kname], rows_of("slot_" + kname), lo, hi)
                else:
                    eng.pack(z["slot_" + kname], out=eng.slots[kname])
            if os.path.exists(filepath + ".json"):
                self.optimizer.iterations = int(json.load(open(filepath + ".json")).get("iterations", 0))
        if os.path.exists(filepath + ".json"):
            cal = json.load(open(filepath + ".json")).get("calibration")
            if cal:
                self.calibration_parameters, self.is_calibrated = cal, True
        self.is_fitted = True
