"""Op-for-op PyTorch-CPU restatement of the reference's TF op graph for one training step and one
evaluation batch (TEST / BASELINE INFRASTRUCTURE ONLY -- timed by bench.py's `cpu_baseline` leg
as kind="port"; validated against oracle/kge_oracle.py in tests/test_ref_cpu.py).

It deliberately keeps the reference's structure, because that structure is what the reference's CPU
path pays for: materialised gathers of 3*B*(1+eta) rows (EmbeddingLookupLayer.py:332-334 called at
ScoringBasedEmbeddingModel.py:252,263), un-fused elementwise scoring (ComplEx.py:53-62), autograd,
duplicate-row summation into a dense gradient, dense Keras-legacy Adam over whole tables
(optimizers.py:166-168), and for evaluation the per-batch 1-vs-all scores + compare-count + per-triple
filter subtraction (AbstractScoringLayer.py:201-307).  The 1-vs-all scores use a matmul rather than
the reference's (n,m,K) broadcast, i.e. this baseline is *faster* than the real TF graph there.
"""
import math

import torch


def _split(x):
    h = x.shape[-1] // 2
    return x[..., :h], x[..., h:]


def scores(model, s, p, o, max_rel_size=None):
    if model == "TransE":
        return -(s + p - o).abs().sum(-1)
    if model == "DistMult":
        return (s * p * o).sum(-1)
    sr, si = _split(s)
    orr, oi = _split(o)
    pr, pi = _split(p)
    if model in ("ComplEx", "HolE"):
        sc = (sr * (pr * orr + pi * oi) + si * (pr * oi - pi * orr)).sum(-1)
        return sc * (2.0 / sr.shape[-1]) if model == "HolE" else sc
    k = sr.shape[-1]
    div = math.sqrt(6.0 / (2 * k * (max_rel_size or 1))) / math.pi
    c, sn = torch.cos(pr / div), torch.sin(pr / div)
    re = sr * c - si * sn - orr
    im = sr * sn + si * c - oi
    return -torch.sqrt(re * re + im * im).sum(-1)


def loss_fn(name, P, N, eta, margin=None, alpha=0.5):
    N = N.reshape(eta, -1)
    if name == "pairwise":
        return torch.clamp((1.0 if margin is None else margin) - P + N, min=0).sum(0).sum()
    if name == "nll":
        Pc, Nc = P.clamp(-75, 75), N.clamp(-75, 75)
        sc = torch.cat([-Pc.repeat(eta).reshape(eta, -1), Nc], 0)
        return torch.log(1 + torch.exp(sc)).sum(0).sum()
    if name == "absolute_margin":
        return (torch.clamp((1.0 if margin is None else margin) + N, min=0) - P).sum(0).sum()
    if name == "self_adversarial":
        g = 3.0 if margin is None else margin
        w = torch.softmax(alpha * N, 0)
        return (-torch.nn.functional.logsigmoid(g + P)
                - (w * torch.nn.functional.logsigmoid(-N - g)).sum(0)).sum()
    if name == "multiclass_nll":
        Pc, Nc = P.clamp(-75, 75), N.clamp(-75, 75)
        return (-torch.log(torch.exp(Pc) / (torch.exp(Nc).sum(0) + torch.exp(Pc)))).sum()
    raise ValueError(name)


def corruptions(pos, n_ents, eta, gen):
    """CorruptionGenerationLayerTrain.call op graph with torch's RNG (tile, coin flip, uniform id)."""
    data = pos.repeat(eta, 1)
    keep = torch.randint(0, 2, (data.shape[0],), generator=gen, dtype=torch.int64)
    repl = torch.randint(0, n_ents, (data.shape[0],), generator=gen, dtype=torch.int64)
    subj = keep * data[:, 0] + (1 - keep) * repl
    obj = (1 - keep) * data[:, 2] + keep * repl
    return torch.stack([subj, data[:, 1], obj], 1)


class RefCpuTrainer:
    def __init__(self, model, ent, rel, eta, loss="self_adversarial", lr=1e-3, max_rel_size=None, seed=0):
        self.model, self.eta, self.loss, self.lr, self.mrs = model, eta, loss, lr, max_rel_size
        self.ent = torch.tensor(ent, dtype=torch.float32, requires_grad=True)
        self.rel = torch.tensor(rel, dtype=torch.float32, requires_grad=True)
        self.m = [torch.zeros_like(self.ent), torch.zeros_like(self.rel)]
        self.v = [torch.zeros_like(self.ent), torch.zeros_like(self.rel)]
        self.t = 0
        self.gen = torch.Generator().manual_seed(seed)

    def forward_backward(self, pos, negs=None):
        pos = torch.as_tensor(pos, dtype=torch.int64)
        if negs is None:
            negs = corruptions(pos, self.ent.shape[0], self.eta, self.gen)
        else:
            negs = torch.as_tensor(negs, dtype=torch.int64)
        for x in (self.ent, self.rel):
            x.grad = None
        sp = scores(self.model, self.ent[pos[:, 0]], self.rel[pos[:, 1]], self.ent[pos[:, 2]], self.mrs)
        sn = scores(self.model, self.ent[negs[:, 0]], self.rel[negs[:, 1]], self.ent[negs[:, 2]], self.mrs)
        L = loss_fn(self.loss, sp, sn, self.eta)
        L.backward()   # index-select backward sums duplicate rows into a dense gradient
        return L.detach()

    def adam(self, b1=0.9, b2=0.999, eps=1e-7):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        with torch.no_grad():
            for x, m, v in ((self.ent, self.m[0], self.v[0]), (self.rel, self.m[1], self.v[1])):
                g = x.grad
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                x.sub_(lr_t * m / (v.sqrt() + eps))

    def step(self, pos, negs=None):
        L = self.forward_backward(pos, negs)
        self.adam()
        return float(L)


def rank_batch(model, ent, rel, triples, fs, fo, max_rel_size=None):
    """One evaluate() batch, both sides, filtered, 'worst' ties: matmul 1-vs-all for the trilinear
    models (ent/rel are torch fp32 tensors; fs/fo lists of int64 tensors).  Returns (n,2) int64."""
    t = torch.as_tensor(triples, dtype=torch.int64)
    s, p, o = ent[t[:, 0]], rel[t[:, 1]], ent[t[:, 2]]
    pos = (scores(model, s, p, o, max_rel_size) * 1000).to(torch.int32)
    out = []
    for side, flt in (("s", fs), ("o", fo)):
        if model == "DistMult":
            q = p * o if side == "s" else s * p
            corr = q @ ent.T
        elif model in ("ComplEx", "HolE"):
            sr, si = _split(s)
            pr, pi = _split(p)
            orr, oi = _split(o)
            if side == "s":
                q = torch.cat([pr * orr + pi * oi, pr * oi - pi * orr], 1)
            else:
                q = torch.cat([sr * pr - si * pi, si * pr + sr * pi], 1)
            corr = q @ ent.T
            if model == "HolE":
                corr = corr * (2.0 / sr.shape[-1])
        elif model == "TransE":
            q = (p - o) if side == "s" else (s + p)
            corr = -torch.cdist(q if side == "o" else -q, ent, p=1)
        else:
            raise NotImplementedError(model)
        cq = (corr * 1000).to(torch.int32)
        rank = (pos[:, None] <= cq).sum(1)
        for i in range(t.shape[0]):  # the reference's per-triple while_loop
            rank[i] -= (pos[i] <= cq[i, flt[i]]).sum()
        out.append(rank)
    return torch.stack(out, 1) + 1
