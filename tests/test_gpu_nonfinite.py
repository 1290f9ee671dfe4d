"""Non-finite scores: the loss kernels follow the reference's NaN semantics.

clip_before_exp (/root/reference/ampligraph/latent_features/loss_functions.py:60-66) is tf.clip_by_value = maximum(minimum(x, 75), -75)
with TensorFlow's NaN-propagating minimum / maximum: a NaN score is a NaN loss VALUE, while its gradient through the clip is the exact
zero of the minimum / maximum masks -- which TensorFlow then multiplies by the score's Jacobian (0 * NaN = NaN reaches every row the
NaN score was computed from).  tf.maximum(h, 0) of the margin losses (:302-308, :458-464) likewise.  The oracle states the same in numpy
(np.clip / np.maximum propagate NaN, dN * score_grads).  C's fminf / fmaxf return the other operand: until round 6 a NaN positive cost a
finite eta * log(1 + e^75).  Here: a NaN / inf entity row (or relation row) is planted and loss, scores and the set of rows that receive
a non-finite gradient -- after a whole step: the rows that ARE non-finite -- are compared with the oracle's on both train paths and on
the column-sharded path."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from test_gpu_kernels import LOSSES, MODELS, dense, dev, loss_desc, make_engine, make_optimizer, rand_triples, run_fwdbwd, run_tiled_grads

pytestmark = pytest.mark.gpu

PLANTS = ["nan_row", "nan_unit", "inf_unit", "nan_rel_unit"]


def plant(ent, rel, X, kind):
    """Spoil one row that is the subject of positive 0 (and, with 1 500 draws over 300 rows, the replacement row of several corruptions)."""
    ent, rel = ent.copy(), rel.copy()
    row = int(X[0, 0])
    if kind == "nan_row":
        ent[row, :] = np.nan
    elif kind == "nan_unit":
        ent[row, 3] = np.nan
    elif kind == "inf_unit":
        ent[row, 5] = np.inf
    else:
        rel[int(X[0, 1]), 2] = np.nan
    return ent, rel


def same_or_nan(a, b, rtol, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((np.isnan(a) & np.isnan(b)) | np.isclose(a, b, rtol=rtol, atol=atol)))


def compare_grads(G, T, what):
    """Rows that hold a non-finite gradient element: the same set on both sides; rows finite on both sides: close (the bars of
    test_gpu_kernels.assert_grads_close, against the finite rows' scale)."""
    T = np.asarray(T, dtype=np.float64)
    bad_g, bad_t = ~np.isfinite(G).all(1), ~np.isfinite(T).all(1)
    assert np.array_equal(bad_g, bad_t), (what, "rows with a non-finite gradient differ", np.nonzero(bad_g != bad_t)[0][:10].tolist(),
                                          int(bad_g.sum()), int(bad_t.sum()))
    ok = ~bad_t
    if ok.any():
        scale = np.maximum(np.abs(T[ok]).max(axis=1, keepdims=True), 1e-6 * max(np.abs(T[ok]).max(), 1e-30))
        err = np.abs(G[ok] - T[ok]) / scale
        assert err.max() < 5e-5, (what, float(err.max()))
    return int(bad_t.sum())


@pytest.mark.parametrize("kind", PLANTS)
@pytest.mark.parametrize("path", ["atomic", "tiled", "tiled_pos_atomic"])
@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("loss", LOSSES)
def test_nonfinite_rows_loss_and_gradients(gpu_lib, model, loss, path, kind):
    N, R, k, B, eta = 300, 5, 32, 257, 6
    eng, ent0, rel0 = make_engine(model, k, N, R, scale=0.6)
    rng = np.random.default_rng(3)
    X = rand_triples(rng, B, N, R)
    ent, rel = plant(ent0, rel0, X, kind)
    eng.set_tables(ent, rel)
    if path == "atomic":
        L, Ge, Gr, ps, ns = run_fwdbwd(eng, X, eta, loss, "sum", seed=9, step=4)
    else:
        L, Ge, Gr, ps, ns = run_tiled_grads(eng, X, eta, loss, "sum", seed=9, step=4, pos_atomic=(path == "tiled_pos_atomic"))
    negs = O.generate_corruptions(X, N, eta, 9, 4)
    with np.errstate(all="ignore"):
        total, Te, Tr, (sp, sn, per) = O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", R)
        ref_loss = float(per.astype(np.float64).sum())
    assert not np.isfinite(sp).all() or not np.isfinite(sn).all()           # the case is what it claims to be
    fin = np.isfinite(sp)
    # scores: non-finite exactly where the oracle's are, equal elsewhere
    assert np.array_equal(np.isfinite(ps), np.isfinite(sp)) and np.array_equal(np.isfinite(ns), np.isfinite(sn))
    assert same_or_nan(ps[fin], sp[fin], 1e-5, 1e-5 * np.abs(sp[fin]).max())
    if kind == "inf_unit":
        # An INF in the tables: which non-finite value a score takes depends on the algebraic form -- the single-pass kernels score a
        # corruption as a dot product with the side row (inf e - inf e' = NaN where the reference's own order gives +-inf; the reference's
        # evaluation uses that same query-vector form), and a score clipped at +-75 makes -log(e^P / Z) underflow in fp32 where the fp64
        # oracle does not.  Held here: the scores' finite / non-finite pattern (above), a loss that is the oracle's or not finite, and
        # that every row the oracle leaves without a finite gradient has none here either.
        assert (not np.isfinite(L)) or abs(L - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (L, ref_loss)
        for G, T in ((Ge, Te), (Gr, Tr)):
            bad_g, bad_t = ~np.isfinite(G).all(1), ~np.isfinite(np.asarray(T, dtype=np.float64)).all(1)
            assert not (bad_t & ~bad_g).any(), (int(bad_t.sum()), int(bad_g.sum()), np.nonzero(bad_t & ~bad_g)[0][:10].tolist())
        return
    # NaN plants, exactly: NaN scores where the oracle's are NaN ...
    assert np.array_equal(np.isnan(ps), np.isnan(sp)) and np.array_equal(np.isnan(ns), np.isnan(sn))
    # ... the loss VALUE NaN where the reference's is NaN, else the oracle's number ...
    assert np.isnan(L) == np.isnan(ref_loss), (L, ref_loss)
    if np.isfinite(ref_loss):
        assert abs(L - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (L, ref_loss)
    # ... and a non-finite gradient in exactly the oracle's rows
    n_bad = compare_grads(Ge, Te, "entity") + compare_grads(Gr, Tr, "relation")
    assert n_bad >= 1


@pytest.mark.parametrize("kind", ["nan_row", "nan_unit"])
@pytest.mark.parametrize("path", ["atomic", "tiled"])
@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("loss", ["nll", "multiclass_nll", "self_adversarial"])
def test_nonfinite_rows_after_whole_steps(gpu_lib, model, loss, path, kind):
    """Two complete steps (dense Adam): the loss of each step and the set of table rows that hold a non-finite value afterwards equal the
    oracle's -- the NaN spreads from the planted row to every row it was scored with, as TensorFlow's 0 * NaN does, and no further."""
    N, R, k, B, eta = 300, 5, 32, 200, 6
    eng, ent0, rel0 = make_engine(model, k, N, R, scale=0.6)
    rng = np.random.default_rng(5)
    Xs = [rand_triples(rng, B, N, R) for _ in range(2)]
    ent, rel = plant(ent0, rel0, Xs[0], kind)
    eng.set_tables(ent, rel)
    w, mk = make_optimizer("adam", {})
    eng.prepare_training(w.name)
    st = mk(ent, rel)
    spread = []
    for t, X in enumerate(Xs, start=1):
        eng.loss_acc.zero_()
        d = w.to_ffi(t, 2)
        if path == "tiled":
            eng.train_step_tiled(dev(X), eta, loss_desc(loss), d, 77, t)
        else:
            eng.train_fwdbwd(dev(X), eta, loss_desc(loss), 77, t)
            eng.opt_step(d, 0.0, 0.0)
        torch.cuda.synchronize()
        with np.errstate(all="ignore"):
            ref_loss = float(O.train_step(st, model, X, eta, loss, 77, t, max_rel_size=R))
        got_loss = float(eng.loss_acc[0].item())
        assert np.isnan(got_loss) == np.isnan(ref_loss), (t, got_loss, ref_loss)
        e, r = eng.get_tables()
        bad_e, bad_o = ~np.isfinite(e).all(1), ~np.isfinite(st.ent).all(1)
        assert np.array_equal(bad_e, bad_o), (t, int(bad_e.sum()), int(bad_o.sum()), np.nonzero(bad_e != bad_o)[0][:10].tolist())
        assert np.array_equal(~np.isfinite(r).all(1), ~np.isfinite(st.rel).all(1)), t
        ok = ~bad_o
        spread.append(int(bad_o.sum()))
        if ok.any():   # (the second step can leave no finite row: five relation rows link everything)
            # TransE: where a unit's gradient cancels to ~0 Adam turns rounding noise into a step of either sign (the bulk criterion of
            # test_gpu_kernels, with the headroom two steps on a NaN-thinned table need: measured 0.971)
            assert np.mean(np.abs(e[ok] - st.ent[ok]) <= 1e-5 + 1e-4 * np.abs(st.ent[ok])) > (0.95 if model == "TransE" else 0.99)
    assert 1 < spread[0] < N      # after ONE step: it spread, and not to everything


@pytest.mark.parametrize("kind", ["nan_row", "inf_unit"])
@pytest.mark.parametrize("loss", ["nll", "multiclass_nll", "self_adversarial", "pairwise"])
@pytest.mark.parametrize("model,k,W", [("ComplEx", 64, 4), ("TransE", 64, 2), ("RotatE", 48, 2), ("DistMult", 96, 2)])
def test_nonfinite_rows_column_sharded(gpu_lib, model, k, W, loss, kind):
    """The column-sharded step (kge_train_cols.h): cols_loss_kernel on the summed partial scores gives the oracle's loss (NaN where it
    is NaN), and every slice's gradient holds non-finite rows exactly where the column slice of the oracle's gradient does."""
    from test_gpu_cols import col_slice

    from ampligraph_amd import _ffi
    from ampligraph_amd.engine import KgeEngine

    N, R, B, eta, seed, step = 300, 5, 203, 7, 9, 3
    rng = np.random.default_rng(0)
    K = O.internal_k(model, k)
    ent0 = (rng.normal(size=(N, K)) * 0.3).astype(np.float32)
    rel0 = (rng.normal(size=(R, K)) * 0.3).astype(np.float32)
    X = rand_triples(np.random.default_rng(1), B, N, R)
    ent, rel = plant(ent0, rel0, X, kind)
    engs = []
    for r in range(W):
        e = KgeEngine(model, k // W, N, R, max_rel_size=R, k_full=k)
        e.set_tables(col_slice(ent, model, k, W, r), col_slice(rel, model, k, W, r))
        engs.append(e)
    negs = O.generate_corruptions(X, N, eta, seed, step)
    with np.errstate(all="ignore"):
        total, Te, Tr, (sp, sn, per) = O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", R)
        ref_loss = float(per.astype(np.float64).sum())
    full = torch.stack([e.cols_partial_scores(dev(X), eta, seed, step).clone() for e in engs]).sum(0)
    ld = loss_desc(loss, "sum")
    for r, e in enumerate(engs):
        e.prepare_training("adam")
        e.loss_acc.zero_()
        sc = full.clone()
        e.cols_loss(ld, sc, B, eta)
        torch.cuda.synchronize()
        lv = float(e.loss_acc[0].item())
        if kind == "inf_unit":   # (see test_nonfinite_rows_loss_and_gradients: the oracle's number, or not finite)
            assert (not np.isfinite(lv)) or abs(lv - ref_loss) <= 3e-5 * max(1.0, abs(ref_loss)), (lv, ref_loss)
        else:
            assert np.isnan(lv) == np.isnan(ref_loss), (lv, ref_loss)
            if np.isfinite(ref_loss):
                assert abs(lv - ref_loss) <= 3e-5 * max(1.0, abs(ref_loss))
        e.g_ent.fill_(123.0)
        e.g_rel.zero_()
        e.train_step_tiled(dev(X), eta, ld, _ffi.Opt(_ffi.OPTIMIZERS["adam"], 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1), seed, step, grad_only=True, given=sc)
        torch.cuda.synchronize()
        Ge, Gr = dense(e, e.g_ent), dense(e, e.g_rel)
        te, tr = col_slice(Te, model, k, W, r), col_slice(Tr, model, k, W, r)
        # a slice sees the NaN of another slice's columns only through the coefficients: rows, not elements
        bad_g, bad_t = ~np.isfinite(Ge).all(1), ~np.isfinite(te).all(1)
        bad_gr, bad_tr = ~np.isfinite(Gr).all(1), ~np.isfinite(tr).all(1)
        if kind == "inf_unit":
            assert not (bad_t & ~bad_g).any() and not (bad_tr & ~bad_gr).any(), (r, int(bad_g.sum()), int(bad_t.sum()))
        else:
            assert np.array_equal(bad_g, bad_t), (r, int(bad_g.sum()), int(bad_t.sum()), np.nonzero(bad_g != bad_t)[0][:10].tolist())
            assert np.array_equal(bad_gr, bad_tr), r


def test_nonfinite_loss_is_reported_by_fit(gpu_lib):
    """The drop-in class: a NaN in the initial entity table is a NaN loss in History from the first epoch on (the reference's
    Keras History would show the same), not a large finite number."""
    from test_gpu_model import toy_graph

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel

    X = toy_graph(5, n=3000, N=40, R=3)
    for model, loss in (("RotatE", "nll"), ("ComplEx", "multiclass_nll"), ("TransE", "pairwise")):
        # (RotatE insists on Glorot-initialised relations, ScoringBasedEmbeddingModel.py:1312-1315: only its entity table is given)
        rng = np.random.default_rng(0)
        K = O.internal_k(model, 10)
        E0 = (rng.normal(size=(40, K)) * 0.3).astype(np.float32)
        R0 = (rng.normal(size=(3, K)) * 0.3).astype(np.float32)
        E0[7, 1] = np.nan
        m = ScoringBasedEmbeddingModel(eta=5, k=10, scoring_type=model, seed=1)
        m.compile(optimizer="adam", loss=loss, entity_relation_initializer=[E0, "glorot_uniform" if model == "RotatE" else R0])
        h = m.fit(X, batch_size=1000, epochs=2, verbose=False).history["loss"]
        assert np.isnan(h).all(), (model, loss, h)
