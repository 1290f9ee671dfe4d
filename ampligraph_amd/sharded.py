"""Row-sharded entity table across the GPUs of one node (BASELINE.json configs[3], configs[4]).

One process per GPU.  Rank r owns the contiguous id range [r*rows_per, min(N, (r+1)*rows_per)) of the entity
table -- the reference's own bucket rule (`owner(e) = e // ceil(N/G)`,
/root/reference/ampligraph/datasets/graph_partitioner.py:339-344, data_indexer.py:183-194) -- together with the
optimizer slots of those rows; the (small) relation table is replicated.  Positives of every global batch
are split over the ranks like in the replicated mode (trainer.shard_bounds).

A step on rank r:
  1. ids this rank's positives (and, with global negatives, its corruptions) touch are classified by owner;
     remote ones are routed: all_to_all(id counts), all_to_all(ids), owners gather the rows,
     all_to_all(rows) back.  The fetched rows land BEHIND the local shard in the same HBM allocation
     (`ent[n_local + j]`), and the batch is re-indexed into that local index space, so the fused HIP kernels
     run unchanged on "a table of n_local + fetched rows".
  2. the fused train step (kge_train_tiled.hip in its gradient-only form, or kge_train.hip) produces the dense
     gradient of local rows AND of the fetched copies;
  3. gradient rows of the fetched copies travel back to their owners (all_to_all, the reverse route) and are
     added there; the relation gradient is all-reduced (R x K floats);
  4. every rank sweeps ITS rows (optimizer + regulariser) and the replicated relation table.

Negative sampling locality is an explicit flag because it changes the distribution:
  * negatives="global" : replacement ids are U{0..N-1} exactly as on one GPU (same Philox rows), ~(G-1)/G of
    the eta rows per positive are remote -> xGMI-bound, kept for parity (N ranks == 1 rank, tested);
  * negatives="local"  : replacement ids are drawn from the rank's own range (what the reference's partitioned
    training does: corruptions come from the partition's entities, ScoringBasedEmbeddingModel.py:227,259-261);
    only the positives' own remote s/o rows move (<= 2 rows per positive).

The training step (ShardedStepLoop.step) issues only libamdkge kernels (kge_shard.hip: device-side routing with
de-duplication, row gather, gradient scatter-add) and equal-split collectives -- no torch compute op, no host sync.
RowExchange below (torch sort / bincount / index_select, host-known variable splits) remains the exchange of the
evaluation / prediction paths, where a host round trip per chunk of test triples is immaterial.
"""
import torch

from . import _ffi
from .trainer import prefer_tiled, shard_bounds


class ShardSpec:
    def __init__(self, n_ents, world, rank):
        self.n_ents, self.world, self.rank = int(n_ents), int(world), int(rank)
        self.rows_per = (self.n_ents + self.world - 1) // self.world
        self.lo = min(self.n_ents, self.rank * self.rows_per)
        self.hi = min(self.n_ents, (self.rank + 1) * self.rows_per)
        self.n_local = self.hi - self.lo

    def owner(self, ids):
        return torch.div(ids, self.rows_per, rounding_mode="floor")

    def local_subset(self, ids):
        """entities_subset (GLOBAL ids, device tensor, duplicates kept: each is a candidate) -> (the local rows among
        them as int32 candidate list, membership mask over the local rows) for sharded_rank_counts."""
        ids = ids.to(torch.int64)
        mine = ids[(ids >= self.lo) & (ids < self.hi)] - self.lo
        mask = torch.zeros(max(1, self.n_local), dtype=torch.bool, device=ids.device)
        mask[mine] = True
        return mine.to(torch.int32).contiguous(), mask


class RowExchange:
    """Routes a list of global entity ids to their owners and back (one instance per step / evaluation)."""

    def __init__(self, spec, dist, ids):
        """ids: int64 1-D tensor of DISTINCT remote global ids; id j of the list is fetched into slot inv[j]."""
        self.spec, self.dist = spec, dist
        dev = ids.device
        owner = spec.owner(ids)
        self.order = torch.argsort(owner, stable=True)              # send order: grouped by owner rank
        self.inv = torch.empty_like(self.order)
        self.inv[self.order] = torch.arange(ids.numel(), device=dev)
        send_ids = ids[self.order].contiguous()
        send_counts = torch.bincount(owner, minlength=spec.world).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts)
        self.send_counts = [int(c) for c in send_counts.tolist()]   # host sync: split sizes must be host ints
        self.recv_counts = [int(c) for c in recv_counts.tolist()]
        self.requested = torch.empty(sum(self.recv_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(self.requested, send_ids, self.recv_counts, self.send_counts)
        self.requested -= spec.lo                                    # local row index at the owner
        self.n = int(ids.numel())

    def slots(self):
        """slot (row n_local + slot of the workspace) of every id of the constructor, in its original order."""
        return self.inv

    def fetch(self, table, n_local):
        """Owners gather the requested rows; the fetched rows are written to table[n_local : n_local + n]."""
        K = table.shape[1]
        out = table.index_select(0, self.requested)
        self._a2a_rows(table[n_local:n_local + self.n], out, self.send_counts, self.recv_counts, K)

    def _a2a_rows(self, dst, src, dst_counts, src_counts, K):
        # rows travel as flat fp32: split sizes in elements
        self.dist.all_to_all_single(dst.reshape(-1), src.reshape(-1), [c * K for c in dst_counts],
                                    [c * K for c in src_counts])

    def return_grads(self, grad, n_local):
        """Gradient rows of the fetched copies (grad[n_local : n_local + n]) go back to the owners and are added
        to the owners' rows."""
        K = grad.shape[1]
        src = grad[n_local:n_local + self.n].contiguous()
        back = torch.empty(len(self.requested), K, dtype=grad.dtype, device=grad.device)
        self._a2a_rows(back, src, self.recv_counts, self.send_counts, K)
        grad.index_add_(0, self.requested, back)


class ShardedStepLoop:
    """Row-sharded counterpart of trainer.StepLoop (same step()/reset_loss()/mean_batch_loss() surface).

    engine: KgeEngine-like backend whose entity table has `spec.n_local + capacity` rows: the local shard first,
    scratch rows for fetched copies behind it, `capacity // world` of them per peer.

    A step issues only libamdkge kernels and collectives, all on the launch stream, with no host round trip: the ids are
    routed on the device (engine.shard_route), every all_to_all has EQUAL, host-known splits (a fixed number of request
    slots per peer; unused slots carry -1 / zero rows), gathers and scatter-adds are kernels.  The price is bandwidth for
    the unused slots; a peer list that overflows sets a sticky device flag that mean_batch_loss() turns into an error."""

    PHASES = ("route+fetch", "kernels", "return", "sweep")   # what kernel_hook(i) / kernel_hook(i + 1) bracket

    def __init__(self, engine, spec, eta, loss, optimizer, regularizer, seed, dist, negatives="local", capacity=None):
        if negatives not in ("local", "global"):
            raise ValueError("negatives must be 'local' or 'global'")
        self.engine, self.spec, self.dist = engine, spec, dist
        self.eta = int(eta)
        self.loss_ffi = loss.to_ffi()
        self.optimizer = optimizer
        self.reg = regularizer
        self.reg_rel = "same"   # relation table: "same" as the entity table's, None, or its own regulariser (trainer.StepLoop)
        self.seed = int(seed)
        self.negatives = negatives
        self.world, self.rank = spec.world, spec.rank
        self.capacity = int(capacity) if capacity is not None else int(engine.ent.shape[0]) - spec.n_local
        if self.capacity < self.world:
            raise ValueError("engine table is smaller than the local shard plus one scratch row per peer")
        self.cap_peer = self.capacity // self.world
        self.n_steps = 0
        self.use_tiled = prefer_tiled(engine)
        self.kernel_hook = None   # bench.py: callable(i) recording HIP events at the phase boundaries (PHASES)
        self._route_counts = None
        import os

        self.deterministic = os.environ.get("AMDKGE_DETERMINISTIC", "0") == "1"   # see trainer.StepLoop
        engine.prepare_training(optimizer.name)
        if hasattr(optimizer, "bind"):
            optimizer.bind(engine)   # get_weights() / set_weights() of the wrapper read and write the engine's state tensors

    @staticmethod
    def peer_capacity(batch_per_rank, eta, negatives, world=1, n_ents=None, cap_factor=None):
        """Request slots per peer.  Default = the WORST case for every world size: every id of the rank's share remote, distinct
        and owned by ONE peer (bounded by the rows a peer owns).  That case is ordinary, not pathological: ids are numbered
        first-seen and batches are sequential slices of the training set (graph_data_loader.py:472-523), so the early batches
        of an epoch hold almost only low ids, all owned by rank 0.  cap_factor (or AMDKGE_SHARD_CAP_FACTOR) sizes the
        lists at that multiple of the even split instead -- for id streams that are uniform by construction (bench.py's
        synthetic graphs), where it saves wire bytes; an overflow is then detected on the device and raised at the epoch's end."""
        import os

        worst = int(batch_per_rank) * (2 + (int(eta) if negatives == "global" else 0))
        if n_ents is not None:
            worst = min(worst, -(-int(n_ents) // int(world)))
        if cap_factor is None and "AMDKGE_SHARD_CAP_FACTOR" in os.environ:
            cap_factor = float(os.environ["AMDKGE_SHARD_CAP_FACTOR"])
        if world <= 2 or cap_factor is None:
            return max(1, worst)
        return max(1, min(worst, int(float(cap_factor) * worst / world) + 64))

    @staticmethod
    def rows_needed(batch_per_rank, eta, negatives, world=1, n_ents=None, cap_factor=None):
        """Scratch rows one step needs behind the shard: `world` peer lists of peer_capacity rows."""
        return int(world) * ShardedStepLoop.peer_capacity(batch_per_rank, eta, negatives, world, n_ents, cap_factor)

    def step(self, global_batch, rng_step, focus=None):
        """focus: None or (w fp32 device tensor [Bg], beta, non-linearity name) -- FocusE, as trainer.StepLoop.step."""
        eng, sp, W, cap = self.engine, self.spec, self.world, self.cap_peer
        hook = self.kernel_hook
        bg = int(global_batch.shape[0])
        lo, hi = shard_bounds(bg, self.world, self.rank)
        if focus is not None:
            fw = focus[0][lo:hi].contiguous()   # kept alive until the launches below are enqueued
            self.loss_ffi.focus_nonlinearity = _ffi.FOCUS_NONLINEARITY[focus[2]]
            self.loss_ffi.focus_beta = float(focus[1])
            self.loss_ffi.d_focus_w = fw.data_ptr()
        else:
            self.loss_ffi.focus_nonlinearity = 0
            self.loss_ffi.d_focus_w = None
        xb = global_batch[lo:hi]
        b = int(xb.shape[0])
        if hook is not None:
            hook(0)
        negs = None
        if self.negatives == "global" and b > 0:
            # the very corruptions one GPU would draw (global Philox rows, ids over all N entities)
            negs = eng.sample_corruptions(xb, self.eta, self.seed, rng_step, sample_base=0,
                                          sample_range=sp.n_ents, row_offset=lo, b_global=bg)
        # ---- 1. route remote ids on the device, fetch their rows behind the shard ----------------------------
        # one scratch row per DISTINCT remote id: equal ids must stay equal after re-indexing (the kernels tell a kept
        # from a replaced side by comparing ids) and their gradients must meet in one row
        xl, nl, send_ids, counts = eng.shard_route(sp, xb, negs, cap)
        self._route_counts = counts
        recv_ids = eng._buf("route_recv", (W * cap,), send_ids.dtype)
        self.dist.all_to_all_single(recv_ids, send_ids)                 # equal splits: `cap` request slots per peer
        rows = eng.gather_rows(eng.ent, recv_ids, "rows_out")           # owner side: requested rows (-1 -> zero row)
        scratch = eng.ent[sp.n_local:sp.n_local + W * cap]
        self.dist.all_to_all_single(scratch.reshape(-1), rows.reshape(-1))   # fetched copies land behind the shard
        if hook is not None:
            hook(1)
        # ---- 2. fused train step on the local index space (gradient only) --------------------------------
        self.optimizer.iterations += 1
        lam = self.reg
        opt_ffi = self.optimizer.to_ffi(self.optimizer.iterations, 2)
        g_scratch = eng.g_ent[sp.n_local:sp.n_local + W * cap]
        if b > 0:
            kw = dict(row_offset=lo, b_global=bg, neg_override=nl)
            if nl is None:   # shard-local negatives: replacement rows are local rows [0, n_local)
                kw.update(sample_base=0, sample_range=sp.n_local)
            tiled = (self.use_tiled or self.deterministic) and eng.tiled_supported(b, self.eta)
            if self.deterministic and not tiled:   # as trainer.StepLoop: never a silent fall-back to the atomic path
                raise ValueError("deterministic mode needs the owner-computes train path (k <= 2048)")
            if tiled:
                if self.deterministic:
                    kw["deterministic"] = True
                eng.train_step_tiled(xl, self.eta, self.loss_ffi, opt_ffi, self.seed, rng_step, grad_only=True, **kw)
            else:
                eng.train_fwdbwd(xl, self.eta, self.loss_ffi, self.seed, rng_step, **kw)
        if hook is not None:
            hook(2)
        # ---- 3. gradients of fetched copies go home; relation gradient is summed over ranks ----------------
        back = eng._buf("grads_back", tuple(g_scratch.shape), g_scratch.dtype)
        self.dist.all_to_all_single(back.reshape(-1), g_scratch.reshape(-1))
        if self.deterministic:   # one peer's list holds distinct rows: peer after peer, every row's additions in rank order
            for q in range(W):
                eng.scatter_add_rows(eng.g_ent, recv_ids[q * cap:(q + 1) * cap], back[q * cap:(q + 1) * cap])
        else:
            eng.scatter_add_rows(eng.g_ent, recv_ids, back)
        eng.zero_(g_scratch)   # (the atomic-scatter train path accumulates into these rows)
        self.dist.all_reduce(eng.g_rel)
        if hook is not None:
            hook(3)
        # ---- 4. every rank sweeps its rows and the replicated relation table -------------------------------
        eng.opt_step(opt_ffi, lam, lam if isinstance(self.reg_rel, str) else self.reg_rel, rows_e=sp.n_local, reg_slots=(1, 2))
        if hook is not None:
            hook(4)
        self.n_steps += 1

    def reset_loss(self):
        self.engine.loss_acc.zero_()
        self.n_steps = 0
        if self._route_counts is not None:
            self.engine.zero_route_overflow()   # the sticky device flag: a new epoch starts clean

    def mean_batch_loss(self):
        """Data loss and the entity-shard regulariser parts are summed over ranks; the relation-table
        regulariser part is identical on every rank and counted once."""
        if self._route_counts is not None and int(self._route_counts[self.world].item()) != 0:
            self.engine.zero_route_overflow()   # sticky since the step that overflowed; reported once
            raise RuntimeError(f"row-sharded step: a peer's request list overflowed its {self.cap_peer} slots (skewed ids?); "
                               "results of this epoch are invalid -- raise AMDKGE_SHARD_CAP_FACTOR or the scratch capacity")
        acc = self.engine.loss_acc.clone()
        part = acc[0:2].clone()
        self.dist.all_reduce(part)
        return (float(part[0].item()) + float(part[1].item()) + float(acc[2].item())) / max(1, self.n_steps)

    # ------------------------------------------------------------------------------------------ tables
    def gather_entity_table(self):
        """Full (N, K) entity table on every rank (checkpointing / tests): all_gather of the padded shards."""
        sp, eng = self.spec, self.engine
        K = eng.ent.shape[1]
        mine = torch.zeros(sp.rows_per, K, dtype=eng.ent.dtype, device=eng.ent.device)
        mine[:sp.n_local] = eng.ent[:sp.n_local]
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return torch.cat(parts)[:sp.n_ents]


def sharded_rank_counts(engine, spec, dist, triples, side, flt=None, subset=None):
    """evaluate() on a row-sharded table: every rank scores ALL queries against ITS rows and the partial
    (greater, equal) counts and filter subtractions are summed over ranks -- the reference's own loop over entity
    partitions (ScoringBasedEmbeddingModel.py:1431-1452) with the partitions living on different GPUs.

    triples: (n,3) int64/int32 GLOBAL ids (same on every rank); flt: None or (lo, hi, ids) with GLOBAL ids;
    subset: None or ShardSpec.local_subset(entities_subset) -- candidates are then the subset's local rows, and filter
    ids outside the subset are dropped (the mapping-table lookup of AbstractScoringLayer.py:266-275).
    Returns (counts (n,2) int32, sub (n,) int32 or None), identical on every rank."""
    sp = spec
    x = triples.to(torch.int64)
    ids = torch.cat([x[:, 0], x[:, 2]])
    remote = (ids < sp.lo) | (ids >= sp.hi)
    rid, rinv = torch.unique(ids[remote], return_inverse=True)
    ex = RowExchange(sp, dist, rid)
    if ex.n > int(engine.ent.shape[0]) - sp.n_local:
        raise RuntimeError("not enough scratch rows behind the shard for this evaluation batch")
    ex.fetch(engine.ent, sp.n_local)
    local_idx = ids - sp.lo
    local_idx[remote] = sp.n_local + ex.slots()[rinv]
    n = int(x.shape[0])
    xl = torch.stack([local_idx[:n], x[:, 1], local_idx[n:]], 1).to(torch.int32).contiguous()
    lflt = None
    if flt is not None:
        lo, hi, fid = flt
        # ids outside [0, n_local) after the shift fail the kernel's range check: exactly the partition rule
        # of AbstractScoringLayer.py:280-288
        lid = fid.to(torch.int64) - sp.lo
        if subset is not None:
            ok = (lid >= 0) & (lid < sp.n_local)
            ok &= subset[1][lid.clamp(0, max(0, sp.n_local - 1))]
            lid = torch.where(ok, lid, torch.full_like(lid, -1))
        lflt = (lo, hi, lid.clamp(min=-1, max=2**31 - 1).to(torch.int32))
    if subset is None:
        _, counts, sub = engine.rank_side(xl, side, "worst", lflt, ent_lo=0, ent_hi=sp.n_local)
    else:
        _, counts, sub = engine.rank_side(xl, side, "worst", lflt, ent_ids=subset[0], ent_lo=0,
                                          ent_hi=int(subset[0].shape[0]), flt_range=(0, sp.n_local))
    dist.all_reduce(counts)
    if sub is not None:
        dist.all_reduce(sub)
    return counts, sub
