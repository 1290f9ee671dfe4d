"""The reference never writes a gradient (TF autodiff, optimizers.py:166) and no reference
test pins one ("parity unpinned"), so the oracle's hand-derived gradients are cross-checked
here against torch.autograd in fp64 on a torch restatement of the reference *forward* code."""
import math

import numpy as np
import pytest
import torch

from oracle import kge_oracle as O


def torch_score(model, s, p, o, max_rel_size):
    if model == "TransE":
        return -(s + p - o).abs().sum(1)
    if model == "DistMult":
        return (s * p * o).sum(1)
    h = s.shape[1] // 2
    sr, si, pr, pi, orr, oi = s[:, :h], s[:, h:], p[:, :h], p[:, h:], o[:, :h], o[:, h:]
    if model in ("ComplEx", "HolE"):
        sc = (sr * (pr * orr + pi * oi) + si * (pr * oi - pi * orr)).sum(1)
        return sc * float(np.float32(2 / h)) if model == "HolE" else sc
    div = float(O.rotate_phase_divisor(h, max_rel_size))
    phi = pr / div
    c, sn = torch.cos(phi), torch.sin(phi)
    re = sr * c - si * sn - orr
    im = sr * sn + si * c - oi
    return -torch.sqrt(re * re + im * im).sum(1)


def torch_loss(name, P, N, eta, prm, red):
    N = N.reshape(eta, -1)
    rs = (lambda x: x.sum(0)) if red == "sum" else (lambda x: x.mean(0))
    if name == "pairwise":
        return rs(torch.clamp(prm["margin"] - P + N, min=0)).sum()
    if name == "nll":
        Pc, Nc = P.clamp(-75, 75), N.clamp(-75, 75)
        sc = torch.cat([-Pc.repeat(eta).reshape(eta, -1), Nc], 0)
        return rs(torch.log(1 + torch.exp(sc))).sum()
    if name == "absolute_margin":
        return rs(torch.clamp(prm["margin"] + N, min=0) - P).sum()
    if name == "self_adversarial":
        w = torch.softmax(prm["alpha"] * N, 0)
        return (-torch.nn.functional.logsigmoid(prm["margin"] + P)
                - rs(w * torch.nn.functional.logsigmoid(-N - prm["margin"]))).sum()
    if name == "multiclass_nll":
        Pc, Nc = P.clamp(-75, 75), N.clamp(-75, 75)
        return (-torch.log(torch.exp(Pc) / (rs(torch.exp(Nc)) + torch.exp(Pc)))).sum()
    raise ValueError(name)


@pytest.mark.parametrize("model", O.MODELS)
@pytest.mark.parametrize("loss", list(O.LOSS_DEFAULTS))
@pytest.mark.parametrize("red", ["sum", "mean"])
def test_dense_gradients_vs_autograd(model, loss, red):
    rng = np.random.default_rng(1)
    N, R, k, B, eta = 23, 3, 6, 9, 4
    K = O.internal_k(model, k)
    ent = (rng.normal(size=(N, K)) * 0.7).astype(np.float32)
    rel = (rng.normal(size=(R, K)) * 0.7).astype(np.float32)
    pos = np.stack([rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)], 1).astype(np.int32)
    negs = O.generate_corruptions(pos, N, eta, seed=5, step=2)
    reg = {"p": 3, "lam_e": 1e-2, "lam_r": 2e-2}
    total, Ge, Gr, _ = O.dense_gradients(model, ent, rel, pos, negs, eta, loss, None, red, R, reg)

    te = torch.tensor(ent, dtype=torch.float64, requires_grad=True)
    tr = torch.tensor(rel, dtype=torch.float64, requires_grad=True)
    tp, tn = torch.tensor(pos, dtype=torch.long), torch.tensor(negs, dtype=torch.long)
    sp = torch_score(model, te[tp[:, 0]], tr[tp[:, 1]], te[tp[:, 2]], R)
    sn = torch_score(model, te[tn[:, 0]], tr[tn[:, 1]], te[tn[:, 2]], R)
    L = torch_loss(loss, sp, sn, eta, O.LOSS_DEFAULTS[loss], red)
    L = L + reg["lam_e"] * te.abs().pow(3).sum() + reg["lam_r"] * tr.abs().pow(3).sum()
    L.backward()
    # scores feeding the oracle's loss are rounded to fp32, so compare at fp32-level tolerance
    assert abs(float(total) - float(L.detach())) <= 2e-5 * max(1.0, abs(float(L.detach())))
    for G, T in ((Ge, te.grad.numpy()), (Gr, tr.grad.numpy())):
        scale = max(1.0, np.abs(T).max())
        assert np.abs(G - T).max() <= 2e-5 * scale, (model, loss, np.abs(G - T).max())


def test_adam_matches_published_rule():
    rng = np.random.default_rng(0)
    st = O.TrainState(rng.normal(size=(5, 4)), rng.normal(size=(2, 4)), "adam", 1e-3)
    x0 = st.ent.astype(np.float64).copy()
    g = rng.normal(size=(5, 4))
    m = v = 0
    x = x0
    for t in range(1, 4):
        O.apply_optimizer(st, g, np.zeros((2, 4)))
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        x = x - 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        assert np.abs(st.ent - x).max() < 1e-6
    # non-lazy: rows with zero gradient still move once m != 0 (here rel had g = 0 always)
    assert st.iterations == 3
