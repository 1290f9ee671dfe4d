"""The COLUMN-SHARDED train step (amdkge_cols_*, AMDKGE_TILED_GIVEN_COEFFS, amdkge_session_group_create_cols; kge_train_cols.h):
every slice holds k / W units of every row and processes the whole batch; the partial score sums of the slices add up to the
reference's scores (TransE.py:51-53, DistMult.py:48, ComplEx.py:58-62, HolE.py:45, RotatE.py:100-104 are sums over units), the loss
on the complete scores is Loss.__call__ (loss_functions.py:185-225), and the backward / optimizer on a slice is the column slice of the
reference's step (ScoringBasedEmbeddingModel.py:370-429).  Against the oracle, phase by phase and as whole steps."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from test_gpu_kernels import assert_grads_close, dense, dev, loss_desc, make_optimizer, rand_triples

pytestmark = pytest.mark.gpu

MODELS = ["TransE", "DistMult", "ComplEx", "HolE", "RotatE"]


def col_slice(a, model, k, W, r):
    kp = k // W
    if model in ("ComplEx", "HolE", "RotatE"):
        return np.ascontiguousarray(np.concatenate([a[:, r * kp:(r + 1) * kp], a[:, k + r * kp:k + (r + 1) * kp]], 1))
    return np.ascontiguousarray(a[:, r * kp:(r + 1) * kp])


def col_merge(parts, model, k):
    if model in ("ComplEx", "HolE", "RotatE"):
        kp = parts[0].shape[1] // 2
        return np.concatenate([p[:, :kp] for p in parts] + [p[:, kp:] for p in parts], 1)
    return np.concatenate(parts, 1)


def make_slices(model, k, W, N, R, scale, seed=0):
    from ampligraph_amd.engine import KgeEngine

    rng = np.random.default_rng(seed)
    K = O.internal_k(model, k)
    ent = (rng.normal(size=(N, K)) * scale).astype(np.float32)
    rel = (rng.normal(size=(R, K)) * scale).astype(np.float32)
    engs = []
    for r in range(W):
        e = KgeEngine(model, k // W, N, R, max_rel_size=R, k_full=k)
        e.set_tables(col_slice(ent, model, k, W, r), col_slice(rel, model, k, W, r))
        engs.append(e)
    return engs, ent, rel


@pytest.mark.parametrize("loss", ["self_adversarial", "nll", "pairwise", "multiclass_nll", "absolute_margin"])
@pytest.mark.parametrize("model,k,W", [("ComplEx", 200, 8), ("ComplEx", 200, 4), ("DistMult", 96, 2), ("TransE", 64, 4), ("HolE", 40, 2), ("RotatE", 48, 4),
                                       ("RotatE", 200, 1), ("TransE", 200, 8), ("DistMult", 512, 2)])
def test_cols_phases_against_oracle(gpu_lib, model, k, W, loss):
    """A: the slices' partial sums add up to the oracle's scores; B: loss value and dL/dscore on the complete sums; C: the slice's
    gradient (gradient-only form of the tile pass) is the column slice of the oracle's dense gradient.  Group widths 16 / 32 / 64
    (13, 25, 50 ... quads per half), ragged last block, eta not a multiple of the rows in flight."""
    N, R, B, eta, seed, step = 300, 5, 203, 7, 9, 3
    engs, ent, rel = make_slices(model, k, W, N, R, 0.3 if k < 100 else 0.1)
    rng = np.random.default_rng(1)
    X = rand_triples(rng, B, N, R)
    X[:5, 2] = X[:5, 0]
    negs = O.generate_corruptions(X, N, eta, seed, step)
    total, Te, Tr, (sp, sn, per) = O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", R)
    co = {}
    O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", R, coeffs=co)
    sgn = -1.0 if model in ("TransE", "RotatE") else (2.0 / k if model == "HolE" else 1.0)
    parts = [e.cols_partial_scores(dev(X), eta, seed, step).clone() for e in engs]
    full = torch.stack(parts).sum(0)
    got = full.cpu().numpy() * np.float32(sgn)
    ref = np.concatenate([sp, sn])
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max()), np.abs(got - ref).max()
    ld = loss_desc(loss, "sum")
    for r, e in enumerate(engs):
        e.prepare_training("adam")
        e.loss_acc.zero_()
        sc = full.clone()
        e.cols_loss(ld, sc, B, eta)
        torch.cuda.synchronize()
        lv = float(e.loss_acc[0].item())
        assert abs(lv - float(total)) <= 3e-5 * max(1.0, abs(float(total))), (lv, float(total))
        coef = sc.cpu().numpy()
        scale = max(np.abs(co["dN"]).max(), np.abs(co["dP"]).max(), 1e-30)
        assert np.allclose(coef[:B], co["dP"], rtol=2e-4, atol=2e-5 * scale) and np.allclose(coef[B:], co["dN"], rtol=2e-4, atol=2e-5 * scale)
        e.g_ent.fill_(123.0)
        e.g_rel.zero_()
        from ampligraph_amd import _ffi

        e.train_step_tiled(dev(X), eta, ld, _ffi.Opt(_ffi.OPTIMIZERS["adam"], 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1), seed, step, grad_only=True, given=sc)
        torch.cuda.synchronize()
        assert_grads_close(dense(e, e.g_ent), col_slice(Te, model, k, W, r), tol=1e-4)
        assert_grads_close(dense(e, e.g_rel), col_slice(Tr, model, k, W, r), tol=1e-4)


@pytest.mark.parametrize("model,k,W,opt", [("ComplEx", 200, 8, "adam"), ("DistMult", 64, 2, "adagrad"), ("TransE", 48, 4, "sgd"), ("RotatE", 100, 2, "adam"),
                                           ("HolE", 96, 4, "rmsprop")])
def test_cols_whole_steps_match_oracle(gpu_lib, model, k, W, opt):
    """Three complete steps on W slices (the score exchange = a sum of the W buffers) == the oracle's dense steps on the whole tables,
    column for column, optimizer slots included; regulariser terms add up over the slices."""
    N, R, B, eta, seed = 200, 4, 151, 5, 4
    engs, ent, rel = make_slices(model, k, W, N, R, 0.3 if k < 100 else 0.1)
    w, mk = make_optimizer(opt, {})
    st = mk(ent, rel)
    for e in engs:
        e.prepare_training(w.name)
    rng = np.random.default_rng(5)
    ld = loss_desc("self_adversarial", "sum")
    oreg = dict(p=2, lam_e=1e-3, lam_r=1e-3)
    for t in range(1, 4):
        X = rand_triples(rng, B, N, R)
        ref = float(O.train_step(st, model, X, eta, "self_adversarial", seed, t, max_rel_size=R, reg=oreg))
        full = torch.stack([e.cols_partial_scores(dev(X), eta, seed, t).clone() for e in engs]).sum(0)
        tot = 0.0
        for r, e in enumerate(engs):
            e.loss_acc.zero_()
            sc = full.clone()
            e.cols_loss(ld, sc, B, eta)
            e.train_step_tiled(dev(X), eta, ld, w.to_ffi(t, 2), seed, t, reg_e=1e-3, reg_r=1e-3, given=sc)
            torch.cuda.synchronize()
            tot += float(e.loss_acc[1].item()) + (float(e.loss_acc[0].item()) if r == 0 else 0.0)
        assert abs(tot - ref) <= 3e-5 * abs(ref), (t, tot, ref)
    E = col_merge([e.get_tables()[0] for e in engs], model, k)
    Rl = col_merge([e.get_tables()[1] for e in engs], model, k)
    assert np.mean(np.abs(E - st.ent) <= 1e-5 + 1e-3 * np.abs(st.ent)) > 0.995 and np.abs(E - st.ent).max() < 2.5e-2
    assert np.mean(np.abs(Rl - st.rel) <= 1e-5 + 1e-3 * np.abs(st.rel)) > 0.99
    for nme in st.slots:
        if nme.endswith("_e"):
            S = col_merge([dense(e, e.slots[nme]) for e in engs], model, k)
            ok = np.isclose(S, st.slots[nme], rtol=2e-3, atol=1e-6 + 2e-5 * np.abs(st.slots[nme]).max())
            assert ok.mean() > 0.99, (nme, ok.mean())


@pytest.mark.parametrize("model,k,W,opt,force_rccl", [("ComplEx", 200, 8, "adam", False), ("DistMult", 64, 2, "adagrad", False), ("TransE", 48, 4, "sgd", False),
                                                      ("RotatE", 100, 2, "adam", False), ("HolE", 24, 3, "adam", False), ("ComplEx", 100, 1, "adam", True)])
def test_session_group_cols_matches_single_session(gpu_lib, model, k, W, opt, force_rccl):
    """amdkge_session_group_create_cols, numpy only: W column slices on device 0 (the score all-reduce = the library's local sum; W = 1
    forced through RCCL: a one-rank ncclAllReduce of the score buffer) == ONE session holding whole rows == the oracle; set_rows /
    get_rows speak whole rows."""
    from ampligraph_amd import _ffi
    from ampligraph_amd.latent_features import loss_functions, optimizers, regularizers
    from ampligraph_amd.session import Session, SessionGroup

    rng = np.random.default_rng(8)
    N, R, B, eta, seed = 140, 4, 257, 4, 6
    K = O.internal_k(model, k)
    sc = 0.3 if k < 100 else 0.1
    ent = (rng.normal(size=(N, K)) * sc).astype(np.float32)
    rel = (rng.normal(size=(R, K)) * sc).astype(np.float32)
    X = np.stack([rng.integers(0, N, 3 * B), rng.integers(0, R, 3 * B), rng.integers(0, N, 3 * B)], 1).astype(np.int32)
    reg = regularizers.get("LP", {"p": 2, "lambda": 1e-3})
    mk = lambda: (loss_functions.get("self_adversarial"), optimizers.get(opt, {"learning_rate": 1e-2}))   # noqa: E731
    single = Session(model, k, N, R, eta, *mk(), reg, seed=seed)
    group = SessionGroup([0] * W, model, k, N, R, eta, *mk(), reg, seed=seed, cols=True, force_rccl=force_rccl)
    assert group.size == W and group.info()[0] == force_rccl
    for s in (single, group):
        s.set_rows("ent", ent)
        s.set_rows("rel", rel)
    assert np.array_equal(group.get_rows("ent"), ent) and np.array_equal(group.get_rows("rel", ids=[3, 0, 3]), rel[[3, 0, 3]])
    st = O.TrainState(ent, rel, opt, 1e-2)
    for t in range(3):
        xb = X[t * B:(t + 1) * B]
        l1, lg = single.train_step(xb), group.train_step(xb)
        ref = float(O.train_step(st, model, xb, eta, "self_adversarial", seed, t, max_rel_size=R, reg=dict(p=2, lam_e=1e-3, lam_r=1e-3)))
        assert abs(lg - ref) <= 3e-5 * abs(ref) and abs(lg - l1) <= 3e-5 * abs(l1), (t, l1, lg, ref)
    eg, es = group.get_rows("ent"), single.get_rows("ent")
    assert np.mean(np.abs(eg - es) <= 1e-5 + 1e-3 * np.abs(es)) > 0.99 and np.abs(eg - es).max() < 2.5e-2
    assert np.mean(np.abs(eg - st.ent) <= 1e-5 + 1e-3 * np.abs(st.ent)) > 0.99
    assert np.mean(np.abs(group.get_rows("rel") - st.rel) <= 1e-5 + 1e-3 * np.abs(st.rel)) > 0.99
    if opt == "adam":
        mg = group.get_rows("ent_slot0")
        assert np.mean(np.abs(mg - st.slots["m_e"]) <= 1e-6 + 2e-3 * np.abs(st.slots["m_e"])) > 0.99
    with pytest.raises(_ffi.AmdKgeError):
        group.rank(X[:4])
    for s in (single, group):
        s.close()


def test_session_group_cols_argument_checks(gpu_lib):
    from ampligraph_amd import _ffi
    from ampligraph_amd.latent_features import loss_functions, optimizers
    from ampligraph_amd.session import SessionGroup

    mk = lambda: (loss_functions.get("nll"), optimizers.get("adam"))   # noqa: E731
    with pytest.raises(_ffi.AmdKgeError):   # k not a multiple of the replica count
        SessionGroup([0] * 3, "DistMult", 8, 50, 2, 2, *mk(), cols=True)
    with pytest.raises(_ffi.AmdKgeError):   # a slice wider than 256 stored units
        SessionGroup([0], "DistMult", 600, 50, 2, 2, *mk(), cols=True)
    with pytest.raises(_ffi.AmdKgeError):
        SessionGroup([0] * 2, "DistMult", 8, 50, 2, 2, *mk(), cols=True, deterministic=True)
    g = SessionGroup([0] * 2, "DistMult", 8, 50, 2, 2, *mk(), cols=True)
    bad = np.array([[0, 0, 50]], dtype=np.int32)
    with pytest.raises(_ffi.AmdKgeError):
        g.train_step(bad)
    g.close()


def test_drop_in_class_column_sharded_fit_equals_single_gpu(gpu_lib, tmp_path):
    """compile(entity_sharding="columns") through the drop-in class on TWO ranks (threads of one process, one GPU, tests/threaded_dist.py):
    ColumnStepLoop over real slice engines == the single-GPU fit (loss history, embeddings, filtered ranks, predictions); the whole
    tables every rank keeps are refreshed before validation and after fit; a checkpoint written by the column-sharded run resumes on
    ONE GPU, and the other way round, like the uninterrupted runs."""
    from test_gpu_model import toy_graph
    from threaded_dist import ThreadedWorld

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers

    X = toy_graph(n=900, N=70, R=4)
    Xt = X[:80]
    ck = str(tmp_path / "ck_cols")

    def new(dist=None, **kw):
        m = ScoringBasedEmbeddingModel(eta=4, k=24, scoring_type="ComplEx", seed=3)
        m._dist_override = dist
        m.compile(optimizer=optimizers.get("adam", {"learning_rate": 1e-2}), loss="self_adversarial", entity_relation_regularizer="l2", **kw)
        return m

    def body(dist):
        m = new(dist, entity_sharding="columns")
        h = m.fit(X, batch_size=256, epochs=2, verbose=False, validation_data=Xt, validation_freq=1, validation_filter={"train": X})
        assert m._col_engine is not None and m._loop.world == 2
        m.save_weights(ck)
        h2 = m.fit(X, batch_size=256, epochs=4, initial_epoch=2, verbose=False)
        ents = np.array([f"e{i}" for i in range(70)])
        return (h.history["loss"] + h2.history["loss"], h.history["val_mrr"], m.get_embeddings(ents), m.predict(Xt),
                m.evaluate(Xt, use_filter={"train": X}, corrupt_side="s,o", verbose=False))

    res = ThreadedWorld(2).run(body)
    m1 = new()
    h1 = m1.fit(X, batch_size=256, epochs=2, verbose=False, validation_data=Xt, validation_freq=1, validation_filter={"train": X})
    h1b = m1.fit(X, batch_size=256, epochs=4, initial_epoch=2, verbose=False)
    ents = np.array([f"e{i}" for i in range(70)])
    e1, p1 = m1.get_embeddings(ents), m1.predict(Xt)
    r1 = m1.evaluate(Xt, use_filter={"train": X}, corrupt_side="s,o", verbose=False)
    for hist, vmrr, emb, pred, ranks in res:
        assert np.allclose(hist, h1.history["loss"] + h1b.history["loss"], rtol=2e-4)
        assert np.allclose(vmrr, h1.history["val_mrr"], atol=0.02)
        assert (np.abs(emb - e1) <= 1e-5 + 1e-3 * np.abs(e1)).mean() > 0.995
        assert np.allclose(pred, p1, rtol=1e-3, atol=1e-4)
        assert (np.abs(ranks - r1) <= 1).mean() > 0.97
    assert np.array_equal(res[0][2], res[1][2])   # the whole tables of the two ranks: the same bits
    # the 2-rank checkpoint (epoch 2) resumed by ONE model, optimizer state included
    m2 = new()
    m2.load_weights(ck)
    h3 = m2.fit(X, batch_size=256, epochs=4, initial_epoch=2, verbose=False)
    assert np.allclose(h3.history["loss"], h1b.history["loss"], rtol=2e-4)


def test_column_sharded_fit_keeps_what_early_stopping_restored(gpu_lib):
    """EarlyStopping(restore_best_weights=True) puts the best tables back into the model's whole-table engine just
    before it stops the run; the column-sharded fit() used to push its slices over them afterwards (and callbacks on non-validation
    epochs saw stale whole tables).  Now the whole tables are refreshed BEFORE the callbacks of every epoch and nothing is pushed after a
    callback: the tables a stopped run ends with are the best epoch's -- on two ranks as on one GPU."""
    from test_gpu_model import toy_graph
    from threaded_dist import ThreadedWorld

    from ampligraph_amd.callbacks import EarlyStopping
    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers

    X = toy_graph(n=900, N=70, R=4)
    ents = np.array([f"e{i}" for i in range(70)])

    class Snapshots:   # what a callback SEES at the end of every epoch (whole tables, read through the public surface)
        def __init__(self):
            self.seen = []

        def set_model(self, m):
            self.m = m

        def on_epoch_end(self, epoch, logs=None):
            self.m.is_fitted = True
            self.seen.append(self.m.get_embeddings(ents).copy())
            logs["scripted"] = [5.0, 4.0, 3.0, 6.0, 7.0, 8.0, 9.0][epoch]   # best at epoch 2; patience 2 stops the run at epoch 4

    def run(dist=None, **kw):
        m = ScoringBasedEmbeddingModel(eta=4, k=24, scoring_type="ComplEx", seed=3)
        m._dist_override = dist
        m.compile(optimizer=optimizers.get("adam", {"learning_rate": 1e-2}), loss="self_adversarial", **kw)
        snap, es = Snapshots(), EarlyStopping(monitor="scripted", mode="min", patience=2, restore_best_weights=True)
        h = m.fit(X, batch_size=256, epochs=7, verbose=False, callbacks=[snap, es])
        return h.history["loss"], es.best_epoch, es.stopped_epoch, snap.seen, m.get_embeddings(ents)

    one = run()
    res = ThreadedWorld(2).run(lambda dist: run(dist, entity_sharding="columns"))
    hist1, best1, stop1, seen1, final1 = one
    assert (best1, stop1) == (2, 4) and len(hist1) == 5, (hist1, best1, stop1)   # the run stopped early, two epochs after its best
    assert np.array_equal(final1, seen1[best1])                              # single GPU: the restored tables are the best epoch's
    for hist, best, stop, seen, final in res:
        assert (best, stop) == (best1, stop1), (hist, hist1)
        assert np.array_equal(final, seen[best])                             # columns: what EarlyStopping restored survives fit()
        assert not np.array_equal(final, seen[stop])
        for a, b in zip(seen, seen1):                                        # every epoch's callback saw THAT epoch's tables
            assert (np.abs(a - b) <= 1e-4 + 2e-2 * np.abs(b)).mean() > 0.97
    assert np.array_equal(res[0][4], res[1][4])
