"""Touched-rows ("lazy") optimizer mode -- the engine's explicit opt-in that deviates from the reference's dense optimizer
(optimizers.py:136-168; amdkge_opt.lazy in include/amdkge.h) -- against the oracle's restatement of the same semantics
(oracle.apply_optimizer_lazy): rows without a gradient keep their bits, touched rows follow the ordinary rule."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from test_gpu_kernels import dense, dev, loss_desc, make_engine, make_optimizer, rand_triples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,hp", [("adam", {}), ("adagrad", {}), ("sgd", {"momentum": 0.9}), ("rmsprop", {"momentum": 0.5})])
@pytest.mark.parametrize("reg", [None, (3, 1e-2)])
def test_lazy_dense_sweep_parity(gpu_lib, name, hp, reg):
    """amdkge_opt_step with lazy: zero gradient rows are skipped bit for bit, the others == oracle."""
    N, R, k = 300, 5, 10            # stored rows of 12 floats
    eng, ent, rel = make_engine("DistMult", k, N, R, scale=0.5)
    w, mk = make_optimizer(name, hp)
    w.lazy = True
    eng.prepare_training(w.name)
    st = mk(ent, rel)
    rng = np.random.default_rng(5)
    oreg = None if reg is None else dict(p=reg[0], lam_e=reg[1], lam_r=reg[1])
    lam = reg[1] if reg else 0.0
    for t in range(1, 5):
        Ge = (rng.normal(size=ent.shape) * (rng.random(size=(N, 1)) < 0.3)).astype(np.float32)   # 70 % of the rows untouched
        Gr = (rng.normal(size=rel.shape) * (np.arange(R)[:, None] != 2)).astype(np.float32)
        eng.pack(Ge, out=eng.g_ent)
        eng.pack(Gr, out=eng.g_rel)
        eng.loss_acc.zero_()
        before_e, before_slots = eng.ent.clone(), {n_: s_.clone() for n_, s_ in eng.slots.items()}
        eng.opt_step(w.to_ffi(t, reg[0] if reg else 2), lam, lam)
        want_reg = O.apply_optimizer_lazy(st, Ge, Gr, oreg)
        torch.cuda.synchronize()
        e, r = eng.get_tables()
        assert np.abs(e - st.ent).max() <= 2e-6 * max(1.0, np.abs(st.ent).max()) and np.abs(r - st.rel).max() <= 2e-6 * max(1.0, np.abs(st.rel).max())
        untouched = torch.as_tensor(~np.any(Ge != 0, axis=1)).cuda()
        assert torch.equal(eng.ent[untouched], before_e[untouched])
        for n_, s_ in eng.slots.items():
            if n_.endswith("_e"):
                assert torch.equal(s_[untouched], before_slots[n_][untouched]), n_
            assert np.allclose(dense(eng, s_), st.slots[n_], rtol=2e-5, atol=2e-7 * max(1e-30, np.abs(st.slots[n_]).max())), n_
        assert float(eng.g_ent.abs().max()) == 0.0 and float(eng.g_rel.abs().max()) == 0.0
        if reg:
            assert abs(float(eng.loss_acc[1]) - want_reg) <= 1e-5 * want_reg


@pytest.mark.parametrize("pos_atomic", [False, True])
@pytest.mark.parametrize("model,k,reg", [("ComplEx", 12, None), ("DistMult", 10, (2, 1e-3)), ("RotatE", 9, (3, 1e-2)), ("TransE", 16, None),
                                          ("RotatE", 600, None)])   # 600: rows shared by a group of waves in the tile kernel
@pytest.mark.parametrize("opt", ["adam", "adagrad"])
def test_lazy_tiled_step_parity(gpu_lib, model, k, reg, opt, pos_atomic):
    """The complete owner-computes step in touched-rows mode (tile flush skips rows without entries; pos_atomic: rows marked by
    the forward kernel's atomics) == oracle train_step(lazy=True), 3 steps; untouched rows keep their bits."""
    N, R, B, eta = (1500, 4, 120, 4) if k < 100 else (300, 3, 24, 3)
    eng, ent, rel = make_engine(model, k, N, R, scale=0.5 if k < 100 else 0.08)
    w, mk = make_optimizer(opt, {})
    w.lazy = True
    eng.prepare_training(w.name)
    st = mk(ent, rel)
    rng = np.random.default_rng(6)
    oreg = None if reg is None else dict(p=reg[0], lam_e=reg[1], lam_r=reg[1])
    lam = reg[1] if reg else 0.0
    for t in range(1, 4):
        X = rand_triples(rng, B, N, R)
        negs = O.generate_corruptions(X, N, eta, 77, t)
        touched = np.zeros(N, dtype=bool)
        touched[X[:, 0]] = True; touched[X[:, 2]] = True; touched[negs[:, 0]] = True; touched[negs[:, 2]] = True
        before = eng.ent.clone()
        ora_before = st.ent.copy()
        eng.loss_acc.zero_()
        eng.train_step_tiled(dev(X), eta, loss_desc("self_adversarial"), w.to_ffi(t, reg[0] if reg else 2), 77, t, reg_e=lam, reg_r=lam,
                             pos_atomic=pos_atomic)
        ref_loss = float(O.train_step(st, model, X, eta, "self_adversarial", 77, t, max_rel_size=R, reg=oreg, lazy=True))
        torch.cuda.synchronize()
        assert float(eng.g_rel.abs().max()) == 0.0 and float(eng.g_ent.abs().max()) == 0.0
        got_loss = float(eng.loss_acc[0].item()) + float(eng.loss_acc[1].item())
        assert abs(got_loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (t, got_loss, ref_loss)
        un = torch.as_tensor(~touched).cuda()
        assert int(un.sum()) > N // 4 and torch.equal(eng.ent[un], before[un])        # untouched rows: same bits
        e, r = eng.get_tables()
        ce = np.abs(e - st.ent) <= 1e-5 + 1e-4 * np.abs(st.ent)
        cr = np.abs(r - st.rel) <= 1e-5 + 1e-4 * np.abs(st.rel)
        assert ce.mean() > 0.995 and cr.mean() > 0.99, (t, ce.mean(), cr.mean())
        assert np.abs(e - st.ent).max() < 2.5e-2
        assert np.array_equal(st.ent[~touched], ora_before[~touched])   # ... in the oracle's restatement as well
        for nme in st.slots:
            # elements that cancel to ~0 carry the absolute noise of their terms: floor relative to the tensor's magnitude
            ok = np.isclose(dense(eng, eng.slots[nme]), st.slots[nme], rtol=1e-3, atol=1e-6 + 2e-5 * np.abs(st.slots[nme]).max())
            # (RotatE's relation gradient sums hardware sqrt / rcp results over every triple of the batch: bulk comparison)
            assert ok.mean() > (0.998 if nme.endswith("_r") else (0.9995 if k >= 100 else 0.9999)), (nme, t, ok.mean())


def test_lazy_fit_matches_oracle_replay(gpu_lib):
    """compile(optimizer_mode="lazy") through the drop-in class == the oracle replaying the same schedule lazily; and it
    differs from the dense run (the mode is not a no-op)."""
    from test_gpu_model import toy_graph

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers, regularizers
    from ampligraph_amd.latent_features.initializers import initialise

    X = toy_graph(3, n=400, N=200, R=4)
    k, eta, bs, epochs, lr = 10, 3, 100, 3, 1e-2
    hists = {}
    for mode in ("lazy", "dense"):
        m = ScoringBasedEmbeddingModel(eta=eta, k=k, scoring_type="ComplEx", seed=5)
        m.compile(optimizer=optimizers.get("adam", {"learning_rate": lr}), loss="nll",
                  entity_relation_regularizer=regularizers.get("LP", {"p": 2, "lambda": 1e-3}), optimizer_mode=mode)
        hists[mode] = m.fit(X, batch_size=bs, epochs=epochs, verbose=False).history["loss"]
        if mode == "lazy":
            lazy_tables = m._engine.get_tables()
    ents, rels = O.first_seen_index(X)
    Xi = O.to_indexes(X, ents, rels)
    N, R, K = len(ents), len(rels), 2 * k
    rng = np.random.Generator(np.random.PCG64(5))
    st = O.TrainState(initialise("glorot_uniform", (N, K), rng), initialise("glorot_uniform", (R, K), rng), "adam", lr)
    steps = (len(Xi) + bs - 1) // bs
    hist = []
    for ep in range(epochs):
        tot = 0.0
        for s in range(steps):
            tot += float(O.train_step(st, "ComplEx", Xi[s * bs:(s + 1) * bs], eta, "nll", 5, ep * steps + s, max_rel_size=R,
                                      reg=dict(p=2, lam_e=1e-3, lam_r=1e-3), lazy=True))
        hist.append(tot / steps)
    assert np.allclose(hists["lazy"], hist, rtol=2e-4), (hists["lazy"], hist)
    close = np.abs(lazy_tables[0] - st.ent) <= 1e-4 + 1e-3 * np.abs(st.ent)
    assert close.mean() > 0.995
    assert not np.allclose(hists["lazy"], hists["dense"], rtol=1e-6)


def test_lazy_pos_atomic_batch_size_changes(gpu_lib):
    """One workspace, batches of different sizes (the last batch of an epoch is short): the row marks the forward kernel's
    atomics leave for the tile pass live at a place that does not move with the batch size, so no step sees stale bytes."""
    N, R, k, eta = 1500, 4, 12, 4
    eng, ent, rel = make_engine("ComplEx", k, N, R, scale=0.5)
    w, mk = make_optimizer("adam", {})
    w.lazy = True
    eng.prepare_training(w.name)
    st = mk(ent, rel)
    rng = np.random.default_rng(8)
    for t, B in enumerate([160, 40, 160, 7, 90], start=1):
        X = rand_triples(rng, B, N, R)
        negs = O.generate_corruptions(X, N, eta, 9, t)
        touched = np.zeros(N, dtype=bool)
        touched[X[:, 0]] = True; touched[X[:, 2]] = True; touched[negs[:, 0]] = True; touched[negs[:, 2]] = True
        before = {n_: s_.clone() for n_, s_ in eng.slots.items() if n_.endswith("_e")}
        before_e = eng.ent.clone()
        eng.train_step_tiled(dev(X), eta, loss_desc("nll"), w.to_ffi(t, 2), 9, t, pos_atomic=True)
        O.train_step(st, "ComplEx", X, eta, "nll", 9, t, max_rel_size=R, lazy=True)
        torch.cuda.synchronize()
        un = torch.as_tensor(~touched).cuda()
        assert torch.equal(eng.ent[un], before_e[un]), t
        for n_, s_ in before.items():
            assert torch.equal(eng.slots[n_][un], s_[un]), (t, n_)
        e, _ = eng.get_tables()
        assert np.abs(e - st.ent).max() < 1e-4, t
