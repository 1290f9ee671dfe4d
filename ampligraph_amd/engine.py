"""Thin Python layer over the C ABI: torch tensors supply device memory and the HIP stream
(plumbing only); every computation is a libamdkge entry point.

`KgeEngine` owns the HBM-resident state of one model replica on one GPU: the two embedding tables,
their dense gradient buffers and optimizer slots, and the scratch used by evaluation.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _ffi
from ._ffi import check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def reg_fields(reg, default_p):
    """Regulariser argument of the engine's step / sweep calls -> (p1, lambda1, p2, lambda2) for the amdkge_opt descriptor.
    `reg`: None / a number (lambda of ONE LP term whose p the descriptor already carries: the form the kernel tests and
    older callers use) or an object with `.terms` = [(p, lambda), ...] (latent_features.regularizers.LPRegularizer: one
    term, or two for Keras' l1_l2)."""
    if reg is None:
        return int(default_p), 0.0, int(default_p), 0.0
    if isinstance(reg, (int, float)):
        return int(default_p), float(reg), int(default_p), 0.0
    t = list(reg.terms)[:2]
    while len(t) < 2:
        t.append((int(default_p), 0.0))
    return int(t[0][0]), float(t[0][1]), int(t[1][0]), float(t[1][1])


def _same_reg(a, b, default_p):
    return reg_fields(a, default_p) == reg_fields(b, default_p)


def _set_reg(opt_desc, reg, default_p):
    opt_desc.reg_p, opt_desc.reg_lambda, opt_desc.reg2_p, opt_desc.reg2_lambda = reg_fields(reg, default_p)


class KgeEngine:
    def __init__(self, scoring_type, k, n_ents, n_rels, max_rel_size=None, device=None, pad=True, k_full=None):
        """k_full: this engine holds a COLUMN SLICE -- k of the k_full units of every row -- of a k_full-unit model (column-sharded
        tables, amdkge_cols_*; HolE's scale and RotatE's phase normaliser are then those of the whole model).
        pad=True (the product's setting): tables are STORED with each half padded to a multiple of 4 units
        (include/amdkge.h "STORED row layout"), which gives every k the 16-byte kernels; `ent`, `rel`, the gradient
        buffers and the optimizer slots then have `Ks` floats per row, of which the dense `K` are live.  pack() /
        unpack() convert; set_tables() / get_tables() speak the dense layout.  pad=False keeps dense rows (tests)."""
        if scoring_type not in _ffi.SCORING_TYPES:
            raise ValueError(f"unknown scoring_type {scoring_type!r}")
        self.lib = _ffi.lib()  # raises when the HIP library is absent: no fallback path exists
        if not torch.cuda.is_available():
            raise RuntimeError("ampligraph_amd needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.scoring_type = scoring_type
        self.k = int(k)
        self.n_ents = int(n_ents)
        self.n_rels = int(n_rels)
        self.K = int(self.lib.amdkge_internal_k(_ffi.SCORING_TYPES[scoring_type], self.k))
        self.ks = int(self.lib.amdkge_padded_k(self.k)) if pad else self.k
        self.model = _ffi.Model(_ffi.SCORING_TYPES[scoring_type], self.k, self.n_ents, self.n_rels,
                                int(max_rel_size) if max_rel_size else 0, self.ks, int(k_full) if k_full else 0, 0)
        self.k_full = int(k_full) if k_full else self.k
        self.Ks = int(self.lib.amdkge_row_floats(C.byref(self.model)))
        # Both tables live in ONE flat allocation (entity rows first, relation rows on a 256-byte boundary behind them),
        # and so do their gradients and every optimizer slot: the multi-GPU step can then treat "all parameters" as one
        # vector (single all-reduce, or slice-wise reduce-scatter / sharded sweep / all-gather).
        ne, nr = self.n_ents * self.Ks, self.n_rels * self.Ks
        self._ne, self._nr = ne, nr
        self._off = (ne + 63) // 64 * 64
        self.p_flat = self._flat()
        self.ent = self.p_flat[:ne].view(self.n_ents, self.Ks)
        self.rel = self.p_flat[self._off:self._off + nr].view(self.n_rels, self.Ks)
        self.g_flat = None
        self.slot_flat = {}
        self.g_ent = None
        self.g_rel = None
        self.slots = {}
        self.opt_kind = None
        # [data loss, regulariser loss, second regulariser slot (row-sharded mode: relation-table part)]
        self.loss_acc = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._work = None
        self._twork = None
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        """Reusable scratch tensor (grown on demand, never shrunk): the step loops allocate nothing per step."""
        n = 1
        for d_ in shape:
            n *= int(d_)
        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:n].view(*shape)

    def _flat(self, pad_to=64 * 16, fill=0.0):
        """Flat fp32 buffer [entity part | pad | relation part | pad]; total length a multiple of `pad_to` floats so that
        it splits evenly over up to 16 ranks in 256-byte-aligned slices."""
        n = (self._off + self._nr + pad_to - 1) // pad_to * pad_to
        return torch.full((n,), float(fill), dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ tables
    def pack(self, dense, out=None):
        """dense rows [n, K] (device tensor or anything np.asarray takes) -> stored rows [n, Ks] (amdkge_pack_rows)."""
        if not torch.is_tensor(dense):
            dense = torch.as_tensor(np.ascontiguousarray(dense, dtype=np.float32))
        dense = dense.to(self.device, torch.float32).contiguous()
        if dense.ndim != 2 or int(dense.shape[1]) != self.K:
            raise ValueError(f"rows must have {self.K} floats, got shape {tuple(dense.shape)}")
        n = int(dense.shape[0])
        if out is None:
            out = torch.empty(n, self.Ks, dtype=torch.float32, device=self.device)
        if tuple(out.shape) != (n, self.Ks) or not out.is_contiguous():
            raise ValueError("pack: output must be a contiguous [n, Ks] tensor")
        check(self.lib.amdkge_pack_rows(C.byref(self.model), _ptr(dense), n, _ptr(out), _stream()))
        return out

    def unpack(self, stored):
        """stored rows [n, Ks] (device tensor) -> dense rows [n, K] (amdkge_unpack_rows)."""
        stored = stored.contiguous()
        n = int(stored.shape[0])
        out = torch.empty(n, self.K, dtype=torch.float32, device=self.device)
        check(self.lib.amdkge_unpack_rows(C.byref(self.model), _ptr(stored), n, _ptr(out), _stream()))
        return out

    def set_tables(self, ent, rel):
        """Dense [n_ents, K] / [n_rels, K] tables (the reference's layout) -> HBM."""
        ent, rel = np.asarray(ent), np.asarray(rel)
        if tuple(ent.shape) != (self.n_ents, self.K) or tuple(rel.shape) != (self.n_rels, self.K):
            raise ValueError(f"table shapes must be {(self.n_ents, self.K)} and {(self.n_rels, self.K)}")
        self.pack(ent, out=self.ent)
        self.pack(rel, out=self.rel)

    def get_tables(self):
        return self.unpack(self.ent).cpu().numpy(), self.unpack(self.rel).cpu().numpy()

    # ------------------------------------------------------------------ training
    def prepare_training(self, optimizer="adam"):
        if optimizer not in _ffi.OPTIMIZERS:
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.opt_kind = optimizer
        ne, nr, off = self._ne, self._nr, self._off
        self.g_flat = self._flat()
        self.g_ent = self.g_flat[:ne].view_as(self.ent)
        self.g_rel = self.g_flat[off:off + nr].view_as(self.rel)
        self.slots, self.slot_flat = {}, {}
        for nme in _ffi.OPT_SLOTS[optimizer]:   # Keras legacy Adagrad initial_accumulator_value = 0.1
            fl = self._flat(fill=0.1 if nme == "a" else 0.0)
            self.slot_flat[nme] = fl
            self.slots[nme + "_e"] = fl[:ne].view_as(self.ent)
            self.slots[nme + "_r"] = fl[off:off + nr].view_as(self.rel)

    def grad_tensors(self):
        return [self.g_flat]

    def train_fwdbwd(self, triples, eta, loss, seed, step, sample_base=0, sample_range=None,
                     row_offset=0, b_global=0, neg_override=None, pos_scores=None, neg_scores=None):
        """triples: int32 cuda tensor (B,3).  Accumulates into g_ent/g_rel and loss_acc[0]."""
        B = int(triples.shape[0])
        if sample_range is None:
            sample_range = self.n_ents
        check(self.lib.amdkge_train_fwdbwd(
            C.byref(self.model), C.byref(loss), _ptr(self.ent), _ptr(self.rel), _ptr(triples), B, int(eta),
            int(sample_base), int(sample_range), int(seed), int(step), int(row_offset), int(b_global),
            _ptr(neg_override), _ptr(self.g_ent), _ptr(self.g_rel), C.c_void_p(self.loss_acc.data_ptr()),
            _ptr(pos_scores), _ptr(neg_scores), _stream()))

    def _slots_of(self, table):
        """(slot0, slot1) tensors of table "e" | "r" in the ABI's slot order (None where the optimizer has none)."""
        names = _ffi.OPT_SLOTS[self.opt_kind]
        got = [self.slots[f"{nme}_{table}"] for nme in names]
        return (got + [None, None])[:2]

    def tiled_supported(self, B, eta):
        """True when the owner-computes step (amdkge_train_step_tiled) covers this shape."""
        return int(self.lib.amdkge_train_tiled_workspace_bytes(C.byref(self.model), int(B), int(eta))) > 0

    def train_step_tiled(self, triples, eta, loss, opt_desc, seed, step, reg_e=0.0, reg_r=0.0, sample_base=0,
                         sample_range=None, row_offset=0, b_global=0, neg_override=None, pos_scores=None,
                         neg_scores=None, grad_only=False, pos_atomic=False, deterministic=False, given=None, det_wide=False):
        """Owner-computes step (kge_train_tiled.hip).  given: the coefficient buffer of the column-sharded step (cols_loss left
        dL/dscore in it: [B] positives then [eta][B] corruptions) -- AMDKGE_TILED_GIVEN_COEFFS, phase C of that step.  grad_only=False: the COMPLETE step -- entity table
        from the LDS tiles, relation table by the fused sweep; g_ent / g_rel (zero on entry) are left zero.
        grad_only=True (data-parallel): g_ent / g_rel (zero on entry) receive the gradients; nothing is updated.
        pos_atomic: skewed graphs, see AMDKGE_TILED_POS_ATOMIC in include/amdkge.h; deterministic: AMDKGE_TILED_DETERMINISTIC
        (bitwise reproducible tables: sorted tile accumulation, staged relation gradient; excludes pos_atomic)."""
        B = int(triples.shape[0])
        hot = getattr(self, "_hot_ids", None) is not None and not pos_atomic and not deterministic
        flags = (1 if pos_atomic else 0) | (2 if deterministic else 0) | (4 if hot else 0) | (16 if deterministic and det_wide else 0)
        if given is not None:
            if pos_atomic or deterministic or int(given.numel()) != B * (1 + int(eta)) or given.dtype != torch.float32:
                raise ValueError("given: one float32 buffer of B (1 + eta) coefficients; excludes pos_atomic / deterministic")
            flags = 8
            pos_scores, neg_scores = given, given.view(-1)[B:]
        if getattr(self, "_twork_flags", flags) & 2 != flags & 2:
            self._twork = None   # the two modes lay out the bookkeeping differently: a workspace serves one of them
        self._twork_flags = flags
        self._last_tiled = (B, int(eta), flags)
        need = int(self.lib.amdkge_train_tiled_workspace_bytes(C.byref(self.model), B, int(eta)))
        if need <= 0:
            raise ValueError("shape not supported by the owner-computes path")
        if self._twork is None or self._twork.numel() < need:
            # zero-filled once; the library keeps its bookkeeping region zero between steps
            self._twork = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self._hot_applied = False
        if hot and not getattr(self, "_hot_applied", False):
            check(self.lib.amdkge_train_tiled_set_hot_rows(C.byref(self.model), _ptr(self._twork), _ptr(self._hot_ids),
                                                           int(self._hot_ids.shape[0]), _stream()))
            self._hot_applied = True
        if sample_range is None:
            sample_range = self.n_ents
        s0, s1 = self._slots_of("e")
        r0, r1 = self._slots_of("r")
        p0 = int(opt_desc.reg_p)   # the descriptor's own p: what a bare lambda refers to
        _set_reg(opt_desc, reg_e, p0)
        rp1, rl1, rp2, rl2 = reg_fields(reg_r, p0)
        if rl1 == 0.0 and rl2 != 0.0:
            rp1, rl1, rl2 = rp2, rl2, 0.0
        opt_desc.rel_reg_p, opt_desc.rel_reg2_p, opt_desc.rel_reg2_lambda = rp1, rp2, rl2
        opt_desc.row_floats = self.Ks
        try:
            check(self.lib.amdkge_train_step_tiled(
                C.byref(self.model), C.byref(loss), C.byref(opt_desc), _ptr(self.ent), _ptr(self.rel), _ptr(s0), _ptr(s1),
                _ptr(r0), _ptr(r1), float(rl1), _ptr(triples), B, int(eta), int(sample_base), int(sample_range),
                int(seed), int(step), int(row_offset), int(b_global), _ptr(neg_override),
                _ptr(self.g_ent), _ptr(self.g_rel), 0 if grad_only else 1, flags,
                C.c_void_p(self.loss_acc.data_ptr()), C.c_void_p(self.loss_acc.data_ptr() + 8),
                _ptr(pos_scores), _ptr(neg_scores), _ptr(self._twork), _stream()))
        except Exception:
            self._twork = None   # bookkeeping may be dirty after a failed launch: start from a fresh zeroed buffer
            raise
        finally:
            opt_desc.reg_p = p0

    # ------------------------------------------------------------------ column-sharded step (amdkge_cols_*, kge_train_cols.h)
    def cols_partial_scores(self, triples, eta, seed, step, sample_base=0, sample_range=None, row_offset=0, b_global=0,
                            neg_override=None, out=None):
        """Phase A: this slice's partial score sums of B positives and their eta corruptions -> float32 [B (1 + eta)] (positives,
        then corruptions at j * B + i); the caller all-reduces it over the ranks."""
        B = int(triples.shape[0])
        if out is None:
            out = self._buf("cols_scores", (B * (1 + int(eta)),), torch.float32)
        check(self.lib.amdkge_cols_partial_scores(C.byref(self.model), _ptr(self.ent), _ptr(self.rel), _ptr(triples), B, int(eta), int(sample_base),
                                                  int(self.n_ents if sample_range is None else sample_range), int(seed), int(step), int(row_offset),
                                                  int(b_global), _ptr(neg_override), _ptr(out), _stream()))
        return out

    def cols_loss(self, loss, scores, B, eta):
        """Phase B: Loss.__call__ on the complete (summed) scores: loss_acc[0] += the data loss, scores <- dL/dscore in place."""
        check(self.lib.amdkge_cols_loss(C.byref(self.model), C.byref(loss), _ptr(scores), int(B), int(eta), C.c_void_p(self.loss_acc.data_ptr()), _stream()))
        return scores

    def set_hot_rows(self, ids):
        """Declare up to 64 hot entity rows (AMDKGE_TILED_HOT_ROWS: skewed graphs); None / empty switches the feature off."""
        if ids is None or len(ids) == 0:
            self._hot_ids = None
        else:
            self._hot_ids = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int32)[:64]).to(self.device)
        self._hot_applied = False
        if self._hot_ids is None and self._twork is not None:   # clear a map an earlier fit left in the workspace
            check(self.lib.amdkge_train_tiled_set_hot_rows(C.byref(self.model), _ptr(self._twork), None, 0, _stream()))

    def tiled_status(self):
        """Sticky status of the owner-computes steps since the last query: 1 = a DETERMINISTIC step fell back to unsorted
        accumulation in some tile.  Only a deterministic step can set it, so only then is the stream synchronised to read it (the
        row-direct pass of long rows handles an overflowing LDS list by rescanning the spill: complete sums, no status)."""
        if self._twork is None or not hasattr(self, "_last_tiled") or not (self._last_tiled[2] & 2):
            return 0
        st = C.c_int32(0)
        B, eta, flags = self._last_tiled
        check(self.lib.amdkge_train_tiled_status(C.byref(self.model), B, eta, flags, _ptr(self._twork), C.byref(st), _stream()))
        return int(st.value)

    def opt_step(self, opt_desc, reg_e=0.0, reg_r=0.0, rows_e=None, reg_slots=(1, 1)):
        """Dense sweep over both tables (optimizer + regulariser + gradient reset).  rows_e limits the entity
        sweep to the first rows_e rows (row-sharded mode: the rows behind them are fetched copies of remote
        rows); reg_slots = loss_acc slots receiving the entity / relation regulariser values."""
        n_e = self.ent.numel() if rows_e is None else int(rows_e) * self.Ks
        opt_desc.row_floats = self.Ks
        p0 = int(opt_desc.reg_p)
        if rows_e is None and _same_reg(reg_e, reg_r, p0) and reg_slots[0] == reg_slots[1] and not opt_desc.lazy:
            # both tables in ONE launch: they live in one flat allocation (the padding between / behind them holds zero
            # parameters and zero gradients, which every rule maps to zero) -- one launch less on launch-bound shapes (C1)
            _set_reg(opt_desc, reg_e, p0)
            names = _ffi.OPT_SLOTS[self.opt_kind]
            sl = [self.slot_flat[n] for n in names] + [None, None]
            n_all = self._off + self._nr
            check(self.lib.amdkge_opt_step(C.byref(opt_desc), _ptr(self.p_flat), _ptr(self.g_flat), _ptr(sl[0]), _ptr(sl[1]), n_all,
                                           C.c_void_p(self.loss_acc.data_ptr() + 8 * int(reg_slots[0])), _stream()))
            opt_desc.reg_p = p0
            return
        for x, g, table, lam, n_el, slot in ((self.ent, self.g_ent, "e", reg_e, n_e, reg_slots[0]),
                                             (self.rel, self.g_rel, "r", reg_r, self.rel.numel(), reg_slots[1])):
            reg_ptr = C.c_void_p(self.loss_acc.data_ptr() + 8 * int(slot))
            _set_reg(opt_desc, lam, p0)
            s0, s1 = self._slots_of(table)
            check(self.lib.amdkge_opt_step(C.byref(opt_desc), _ptr(x), _ptr(g), _ptr(s0), _ptr(s1),
                                           n_el, reg_ptr, _stream()))
        opt_desc.reg_p = p0

    def opt_step_flat(self, opt_desc, lo, hi, reg_e=0.0, reg_r=0.0, reg_slot=1):
        """Dense sweep over elements [lo, hi) of the flat parameter vector (sharded-optimizer data parallelism: a rank
        sweeps only its slice).  The slice may straddle the entity / relation boundary; the regulariser lambda follows."""
        segs = []
        a, b = max(lo, 0), min(hi, self._ne)
        if b > a:
            segs.append((a, b, reg_e))
        # (the padding between / behind the tables holds zero parameters and zero gradients: nothing to sweep)
        a, b = max(lo, self._off), min(hi, self._off + self._nr)
        if b > a:
            segs.append((a, b, reg_r))
        reg_ptr = C.c_void_p(self.loss_acc.data_ptr() + 8 * int(reg_slot))
        names = _ffi.OPT_SLOTS[self.opt_kind]
        if opt_desc.lazy:
            raise ValueError("the touched-rows optimizer mode sweeps whole rows: use the all-reduce gradient merge")
        p0 = int(opt_desc.reg_p)
        for a, b, lam in segs:
            _set_reg(opt_desc, lam, p0)
            sl = [self.slot_flat[n][a:b] for n in names] + [None, None]
            check(self.lib.amdkge_opt_step(C.byref(opt_desc), _ptr(self.p_flat[a:b]), _ptr(self.g_flat[a:b]), _ptr(sl[0]),
                                           _ptr(sl[1]), b - a, reg_ptr, _stream()))
        opt_desc.reg_p = p0

    # ------------------------------------------------------------------ multi-GPU data path (kge_shard.hip)
    def shard_route(self, spec, triples, negs, cap):
        """amdkge_shard_route: int32 device triples [b,3] (+ corruptions [nneg,3] or None) with GLOBAL ids -> the same in this
        rank's local index space, plus the per-peer request lists.  Returns (xl, nl, send_ids [world*cap], counts [world+1]);
        counts[world] is a sticky overflow flag (zeroed by zero_route_overflow)."""
        b = int(triples.shape[0])
        nneg = 0 if negs is None else int(negs.shape[0])
        xl = self._buf("route_xl", (b, 3), torch.int32)
        nl = self._buf("route_nl", (nneg, 3), torch.int32) if negs is not None else None
        send_ids = self._buf("route_send", (spec.world * int(cap),), torch.int32)
        counts = self._bufs.get("route_counts")
        if counts is None or counts.numel() != spec.world + 1:
            counts = self._bufs["route_counts"] = torch.zeros(spec.world + 1, dtype=torch.int32, device=self.device)
        need = int(self.lib.amdkge_shard_route_workspace_bytes(b, nneg))
        if need < 0:
            raise ValueError("batch too large for one shard_route call")
        work = self._buf("route_work", (need,), torch.uint8)
        check(self.lib.amdkge_shard_route(spec.n_ents, spec.world, spec.rank, _ptr(triples), b, _ptr(negs), nneg, int(cap),
                                          _ptr(xl), _ptr(nl), _ptr(send_ids), _ptr(counts), _ptr(work), _stream()))
        return xl, nl, send_ids, counts

    def zero_route_overflow(self):
        """Clear the sticky overflow flag of shard_route (counts[world]); stream-ordered."""
        counts = self._bufs.get("route_counts")
        if counts is not None:
            self.zero_(counts[-1:])

    def gather_rows(self, table, idx, name="gathered"):
        """rows table[idx] (idx < 0: zero rows) into a reusable [n, Ks] buffer (amdkge_gather_rows)."""
        n = int(idx.shape[0])
        out = self._buf(name, (n, int(table.shape[1])), torch.float32)
        check(self.lib.amdkge_gather_rows(_ptr(table), int(table.shape[1]), _ptr(idx), n, _ptr(out), _stream()))
        return out

    def scatter_add_rows(self, table, idx, src):
        """table[idx[j]] += src[j] (idx < 0 skipped) (amdkge_scatter_add_rows)."""
        check(self.lib.amdkge_scatter_add_rows(_ptr(table), int(table.shape[1]), _ptr(idx), int(idx.shape[0]), _ptr(src), _stream()))

    def zero_(self, t):
        """hipMemsetAsync on the launch stream (no torch op in the step loops)."""
        if t.numel():
            check(self.lib.amdkge_dev_memset(_ptr(t), 0, t.numel() * t.element_size(), _stream()))

    def opt_step_merged(self, opt_desc, lo, hi, parts, n_parts, part_stride, reg_e=0.0, reg_r=0.0, reg_slot=1):
        """Sharded-optimizer data parallelism: sweep elements [lo, hi) of the flat parameter vector with the gradient
        sum_q parts[q * part_stride + (i - lo)] (amdkge_opt_step_merged: the W partial slices are summed inside the sweep)."""
        segs = []
        a, b = max(lo, 0), min(hi, self._ne)
        if b > a:
            segs.append((a, b, reg_e))
        a, b = max(lo, self._off), min(hi, self._off + self._nr)
        if b > a:
            segs.append((a, b, reg_r))
        reg_ptr = C.c_void_p(self.loss_acc.data_ptr() + 8 * int(reg_slot))
        names = _ffi.OPT_SLOTS[self.opt_kind]
        p0 = int(opt_desc.reg_p)
        for a, b, lam in segs:
            _set_reg(opt_desc, lam, p0)
            sl = [self.slot_flat[n][a:b] for n in names] + [None, None]
            check(self.lib.amdkge_opt_step_merged(C.byref(opt_desc), _ptr(self.p_flat[a:b]), _ptr(parts[a - lo:]), int(n_parts),
                                                  int(part_stride), _ptr(sl[0]), _ptr(sl[1]), b - a, reg_ptr, _stream()))
        opt_desc.reg_p = p0

    def synth_triples(self, seed, first_row, n, n_ents, n_rels, out=None):
        """int32 [n,3] triples number first_row .. of the counter-based synthetic stream (amdkge_synth_triples)."""
        if out is None:
            out = torch.empty(int(n), 3, dtype=torch.int32, device=self.device)
        check(self.lib.amdkge_synth_triples(int(seed), int(first_row), int(n), int(n_ents), int(n_rels), _ptr(out), _stream()))
        return out

    def sample_corruptions(self, triples, eta, seed, step, sample_base=0, sample_range=None,
                           row_offset=0, b_global=0):
        B = int(triples.shape[0])
        out = torch.empty(B * eta, 3, dtype=torch.int32, device=self.device)
        check(self.lib.amdkge_sample_corruptions(
            _ptr(triples), B, int(eta), int(sample_base), int(sample_range or self.n_ents), int(seed),
            int(step), int(row_offset), int(b_global), _ptr(out), _stream()))
        return out

    def filter_build(self, triples, side, n_ents, n_rels):
        """amdkge_filter_build: the CSR filter index of one side from the concatenated id triples (int32 [m,3] device tensor).
        -> (keys int64 [n_groups], start int64 [n_groups + 1], ids int32 [n_unique]) device tensors.  One stream
        synchronisation (the two counts come back to size the views); an index is built once per evaluate() and cached."""
        m = int(triples.shape[0])
        keys = torch.empty(max(m, 1), dtype=torch.int64, device=self.device)
        start = torch.empty(m + 1, dtype=torch.int64, device=self.device)
        ids = torch.empty(max(m, 1), dtype=torch.int32, device=self.device)
        counts = torch.zeros(2, dtype=torch.int64, device=self.device)
        need = int(self.lib.amdkge_filter_build_workspace_bytes(m, int(n_ents), int(n_rels)))
        if need < 0:
            raise ValueError("filter_build: bad sizes")
        work = self._buf("filter_build_work", (need,), torch.uint8)
        check(self.lib.amdkge_filter_build(_ptr(triples), m, 1 if side == "s" else 2, int(n_ents), int(n_rels), _ptr(keys), _ptr(start),
                                           _ptr(ids), _ptr(counts), _ptr(work), _stream()))
        ng, nu = (int(c) for c in counts.tolist())
        return keys[:ng], start[:ng + 1], ids[:nu]

    def filter_ranges(self, keys, start, triples, side, n_ents, n_rels):
        """(lo, hi) int64 device tensors: each triple's range in a FilterIndex id array (amdkge_filter_ranges)."""
        n = int(triples.shape[0])
        lo = torch.empty(n, dtype=torch.int64, device=self.device)
        hi = torch.empty(n, dtype=torch.int64, device=self.device)
        check(self.lib.amdkge_filter_ranges(_ptr(keys), _ptr(start), int(keys.shape[0]), _ptr(triples), n, int(side),
                                            int(n_ents), int(n_rels), _ptr(lo), _ptr(hi), _stream()))
        return lo, hi

    def compose_ranks(self, counts, sub, strategy, out=None, out_stride=1):
        """(greater, equal) counts [+ filter subtraction] -> 1-based ranks (amdkge_rank_compose)."""
        n = int(counts.shape[0])
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
            out_stride = 1
        check(self.lib.amdkge_rank_compose(_ptr(counts), _ptr(sub), n, _ffi.RANK_STRATEGY[strategy], _ptr(out),
                                           int(out_stride), _stream()))
        return out

    # ------------------------------------------------------------------ predict
    def score(self, triples):
        n = int(triples.shape[0])
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.amdkge_score(C.byref(self.model), _ptr(self.ent), _ptr(self.rel), _ptr(triples), n,
                                    _ptr(out), _stream()))
        return out

    # ------------------------------------------------------------------ discovery (kge_discovery.hip, kge_rank.hip)
    SCORE_CHUNK_BYTES = 256 << 20   # bound of the transient (queries x candidates) score block

    def topk_rows(self, vals, k, largest=True, col_scale=None, col_bias=None, payload=None):
        """(idx int32 [n,k], val fp32 [n,k]) of the k best entries per row of vals [n,m] (amdkge_topk_rows); with `payload`
        (int32, same shape and row stride as vals) idx holds the payload entries of the selected columns."""
        n, m = int(vals.shape[0]), int(vals.shape[1])
        if int(k) > 1024:
            # beyond the streaming selection's k: a full stable sort of the (scaled) row on the device, which is what the
            # reference does for every top_n (argsort over all candidates, discovery.py:1150-1160); same order rule as the
            # kernel (best first, equal values by increasing column), missing entries (m < k) as index -1 / -inf
            v = vals.to(torch.float32)
            if col_scale is not None:
                v = v * col_scale[None, :m]
            if col_bias is not None:
                v = v + col_bias[None, :m]
            v = torch.nan_to_num(v, nan=float("-inf") if largest else float("inf"), posinf=float("inf"), neginf=float("-inf"))   # NaN ranks last, as in the kernel
            sv, si = torch.sort(v, dim=1, descending=bool(largest), stable=True)
            kk = min(int(k), m)
            idx = torch.full((n, int(k)), -1, dtype=torch.int32, device=self.device)
            val = torch.full((n, int(k)), float("-inf") if largest else float("inf"), dtype=torch.float32, device=self.device)
            sel = si[:, :kk]
            idx[:, :kk] = (sel if payload is None else torch.gather(payload[:, :m].to(torch.int64), 1, sel)).to(torch.int32)
            val[:, :kk] = sv[:, :kk]
            return idx, val
        idx = torch.empty(n, int(k), dtype=torch.int32, device=self.device)
        val = torch.empty(n, int(k), dtype=torch.float32, device=self.device)
        check(self.lib.amdkge_topk_rows(_ptr(vals), n, m, int(vals.stride(0)), _ptr(col_scale), _ptr(col_bias), _ptr(payload), int(k), 1 if largest else 0,
                                        _ptr(idx), _ptr(val), _stream()))
        return idx, val

    def corruption_topk(self, triples, side, k, ent_ids=None, ent_lo=0, ent_hi=None):
        """Top-k scoring corruptions of one side for every query triple: (positions int32 [n,k] into the candidate list /
        row range, scores fp32 [n,k]).  Queries go through amdkge_corruption_scores in chunks whose score block stays under
        SCORE_CHUNK_BYTES, each chunk straight into amdkge_topk_rows."""
        n = int(triples.shape[0])
        if ent_hi is None:
            ent_hi = self.n_ents if ent_ids is None else int(ent_ids.shape[0])
        m = int(ent_hi) - int(ent_lo)
        rows = max(1, min(n, self.SCORE_CHUNK_BYTES // max(4 * m, 1)))
        out_i = torch.empty(n, int(k), dtype=torch.int32, device=self.device)
        out_v = torch.empty(n, int(k), dtype=torch.float32, device=self.device)
        for c0 in range(0, n, rows):
            c1 = min(n, c0 + rows)
            blk = self._buf("disc_scores", (c1 - c0, m), torch.float32)
            work = self._workspace(c1 - c0)
            check(self.lib.amdkge_corruption_scores(C.byref(self.model), _ptr(self.ent), _ptr(self.rel), _ptr(triples[c0:c1]), c1 - c0, int(side),
                                                    _ptr(ent_ids), int(ent_lo), int(ent_hi), _ptr(blk), m, _ptr(work), _stream()))
            i_, v_ = self.topk_rows(blk, k)
            out_i[c0:c1], out_v[c0:c1] = i_, v_
        return out_i, out_v

    def nearest_rows(self, q_rows, k, metric="euclidean", ent_ids=None, ent_lo=0, ent_hi=None, table=None):
        """k nearest table rows of every query row (stored layout, [n, Ks]), nearest first: (positions int32 [n,k],
        distances fp32 [n,k]).  Selection in GEMM form on the tile kernel (amdkge_row_dots) with the norms folded into
        amdkge_topk_rows' column scale / bias (euclidean: v = <q,e> - |e|^2 / 2; cosine: v = <q,e> / |e|) -- the distance matrix
        never exists; the k survivors are then re-measured exactly (amdkge_pair_distances: the GEMM form cancels for near
        neighbours) and put in their final order."""
        table = self.ent if table is None else table
        n = int(q_rows.shape[0])
        if ent_hi is None:
            ent_hi = int(table.shape[0]) if ent_ids is None else int(ent_ids.shape[0])
        m = int(ent_hi) - int(ent_lo)
        Kf = int(table.shape[1])
        q_rows = q_rows.contiguous()
        col = torch.empty(m, dtype=torch.float32, device=self.device)
        cosine = metric == "cosine"
        check(self.lib.amdkge_row_sqnorms(_ptr(table), Kf, _ptr(ent_ids), int(ent_lo), m, -0.5, 1 if cosine else 0, _ptr(col), _stream()))
        rows = max(1, min(n, self.SCORE_CHUNK_BYTES // max(4 * m, 1)))
        out_i = torch.empty(n, int(k), dtype=torch.int32, device=self.device)
        out_v = torch.empty(n, int(k), dtype=torch.float32, device=self.device)
        for c0 in range(0, n, rows):
            c1 = min(n, c0 + rows)
            blk = self._buf("disc_scores", (c1 - c0, m), torch.float32)
            check(self.lib.amdkge_row_dots(_ptr(q_rows[c0:c1]), c1 - c0, _ptr(table), Kf, _ptr(ent_ids), int(ent_lo), int(ent_hi), _ptr(blk), m,
                                           _stream()))
            i_, v_ = self.topk_rows(blk, k, True, col if cosine else None, None if cosine else col)
            out_i[c0:c1], out_v[c0:c1] = i_, v_
        dist = torch.empty(n, int(k), dtype=torch.float32, device=self.device)
        check(self.lib.amdkge_pair_distances(_ptr(q_rows), n, _ptr(table), Kf, _ptr(ent_ids), int(ent_lo), _ptr(out_i), int(k), 1 if cosine else 0,
                                             _ptr(dist), _stream()))
        return self.topk_rows(dist, k, largest=False, payload=out_i)   # final order by the exact distances

    def platt_step(self, scores_pos, scores_neg, w, b, label_pos, label_neg, weight_pos, weight_neg):
        """(loss, dloss/dw, dloss/db) of the Platt-scaling objective for one batch (amdkge_platt_step)."""
        out = torch.zeros(3, dtype=torch.float64, device=self.device)
        check(self.lib.amdkge_platt_step(_ptr(scores_pos), int(scores_pos.shape[0]), _ptr(scores_neg),
                                         int(scores_neg.shape[0]), float(w), float(b), float(label_pos), float(label_neg),
                                         float(weight_pos), float(weight_neg), _ptr(out), _stream()))
        return [float(v) for v in out.tolist()]

    # ------------------------------------------------------------------ evaluate
    def _workspace(self, n, lane=0):
        need = int(self.lib.amdkge_rank_workspace_bytes(C.byref(self.model), n))
        if lane:
            return self._buf(f"rank_work_lane{lane}", (need,), torch.uint8)
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._work

    SCREEN_MAX_BYTES = 4 << 30

    def _side_stream(self, lane=0):
        """Extra streams of this engine's device: 2 * lane for the filter pass of rank_side (beside the count pass), 2 * lane + 1
        for the lane itself when rank_sides runs several corruption sides at once."""
        st = getattr(self, "_fstreams", None)
        if st is None:
            st = self._fstreams = {}
        if lane not in st:
            st[lane] = torch.cuda.Stream(device=self.device)
        return st[lane]

    def screen_stats(self):
        """Of the LAST rank_side call (with rank_sides: its last lane): None when no screening / early-exit workspace was handed to
        the count pass (huge candidate ranges, or the library sees nothing to gain), else (pairs the exact chain re-checked,
        fell back to the exact kernel?) -- contraction models: of the int8 screening pass; TransE / RotatE: of the exact early exit
        ((0, False) when its probe picked the plain kernel; likewise problems too small for either pass).  Synchronises."""
        s = getattr(self, "_last_screen", None)
        if s is None:
            return None
        v = s[:8].view(torch.int32).cpu().numpy()
        return int(v[0]), bool(v[1])

    def rank_sides(self, triples, jobs, strategy="worst", ent_ids=None, subset_pos=None):
        """Several corruption sides of the same triples AT ONCE: jobs = [(side, flt, out, out_stride), ...], each on a stream and
        with workspaces of its own, joined into the current stream.  A side is one long kernel (the screening / count pass) between
        a dozen short latency-bound ones (query vectors, limbs, thresholds, merge, compose: ~13 % of a side at the C2 shape); with
        two sides in flight the short kernels of one run under the long kernel of the other."""
        if len(jobs) == 1:
            side, flt, out, stride = jobs[0]
            return [self.rank_side(triples, side, strategy, flt, ent_ids, subset_pos, out=out, out_stride=stride)]
        if os.environ.get("AMDKGE_EVAL_LANES") == "1":   # (A/B measurements: the sides one after the other)
            return [self.rank_side(triples, side, strategy, flt, ent_ids, subset_pos, out=out, out_stride=stride) for side, flt, out, stride in jobs]
        # Every lane keeps workspaces of its own (query vectors, the screening pass's fixed-point rows and recheck list -- up to
        # SCREEN_MAX_BYTES --, the filter pass's copy), so two sides in flight about double evaluate()'s peak memory.
        # Beside resident training state on a large table that can be what runs the device out of memory: the sides then run one
        # after the other on the first lane's workspaces.
        n = int(triples.shape[0])
        m = (self.n_ents if ent_ids is None else int(ent_ids.shape[0]))
        per_lane = 2 * int(self.lib.amdkge_rank_workspace_bytes(C.byref(self.model), n)) + \
            min(max(int(self.lib.amdkge_rank_screen_workspace_bytes(C.byref(self.model), n, m)), 0), self.SCREEN_MAX_BYTES)
        extra = per_lane * (len(jobs) - 1)
        have = sum(int(t.numel()) * int(t.element_size()) for k_, t in self._bufs.items() if "_lane" in k_)   # BYTES earlier calls already hold
        # (the driver query costs ~10 us and is only made when the lanes' workspaces still have to grow)
        if extra > have and extra - have > torch.cuda.mem_get_info(self.device)[0] // 4:
            return [self.rank_side(triples, side, strategy, flt, ent_ids, subset_pos, out=out, out_stride=stride) for side, flt, out, stride in jobs]
        main = torch.cuda.current_stream()
        res, lanes = [], []
        for i, (side, flt, out, stride) in enumerate(jobs):
            st = self._side_stream(2 * i + 1)
            st.wait_stream(main)
            lanes.append(st)
            with torch.cuda.stream(st):
                res.append(self.rank_side(triples, side, strategy, flt, ent_ids, subset_pos, out=out, out_stride=stride, lane=i))
        for st in lanes:
            main.wait_stream(st)
        return res

    def rank_side(self, triples, side, strategy="worst", flt=None, ent_ids=None, subset_pos=None,
                  ent_lo=0, ent_hi=None, out=None, out_stride=1, flt_range=None, lane=0):
        """Ranks (1-based, reference semantics) of `triples` for one corruption side.

        flt: None or (lo int64[n], hi int64[n], ids int32[*]) cuda tensors; flt_range: id range the filter ids are
        checked against when it differs from the candidate positions [ent_lo, ent_hi) (row-sharded subsets); lane: which set of
        workspaces / side stream to use (rank_sides)."""
        n = int(triples.shape[0])
        if ent_hi is None:
            ent_hi = self.n_ents if ent_ids is None else int(ent_ids.shape[0])
        work = self._workspace(n, lane)
        sfx = f"_lane{lane}" if lane else ""
        counts = torch.zeros(n, 2, dtype=torch.int32, device=self.device)
        # contraction models: the int8 screening pass + exact recheck (kge_rank_screen.h) -- the same counts, bit for bit, at a
        # multiple of the fp32 matrix rate; its workspace (fixed-point copies of the query vectors and the candidate rows, the
        # recheck list) is kept between calls.  Beyond SCREEN_MAX_BYTES (huge candidate ranges) the exact kernel runs alone.
        screen, sbytes = None, 0
        need = int(self.lib.amdkge_rank_screen_workspace_bytes(C.byref(self.model), n, int(ent_hi) - int(ent_lo))) if n > 0 else 0
        if 0 < need <= self.SCREEN_MAX_BYTES:
            screen, sbytes = self._buf("rank_screen" + sfx, (need,), torch.uint8), need
            screen[:256].zero_()   # the statistics words are written only by the passes that run (stale bytes otherwise)
        self._last_screen = screen
        # The filter pass (a latency-bound walk over each triple's known positives) is independent of the count pass: it runs
        # beside it on a second stream with a workspace of its own and is joined before the two are composed.
        sub, main, fstream = None, torch.cuda.current_stream(), None
        if flt is not None:
            lo, hi, ids = flt
            sub = torch.zeros(n, dtype=torch.int32, device=self.device)
            f_lo, f_hi = (ent_lo, ent_hi) if flt_range is None else flt_range
            fwork = self._buf("rank_work_filter" + sfx, (work.numel(),), torch.uint8)
            fstream = self._side_stream(2 * lane)
            fstream.wait_stream(main)
            with torch.cuda.stream(fstream):
                check(self.lib.amdkge_rank_filter(C.byref(self.model), _ptr(self.ent), _ptr(self.rel), _ptr(triples),
                                                  n, side, _ptr(lo), _ptr(hi), _ptr(ids), _ptr(subset_pos),
                                                  int(f_lo), int(f_hi), _ptr(sub), _ptr(fwork), _stream()))
        check(self.lib.amdkge_rank_counts_screened(C.byref(self.model), _ptr(self.ent), _ptr(self.rel), _ptr(triples), n,
                                                   side, _ptr(ent_ids), int(ent_lo), int(ent_hi), _ptr(counts),
                                                   _ptr(work), _ptr(screen), sbytes, _stream()))
        if fstream is not None:
            main.wait_stream(fstream)
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
            out_stride = 1
        check(self.lib.amdkge_rank_compose(_ptr(counts), _ptr(sub), n, _ffi.RANK_STRATEGY[strategy], _ptr(out),
                                           int(out_stride), _stream()))
        return out, counts, sub
