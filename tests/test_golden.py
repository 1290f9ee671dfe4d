"""Committed golden fixtures (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them bit for bit (freezes the checker).  GPU: the HIP path against the fixtures."""
import os

import numpy as np
import pytest

from oracle import kge_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))
MODELS = ["TransE", "DistMult", "ComplEx", "HolE", "RotatE"]
LOSSES = ["pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll"]
N, R, K_UNITS, B, ETA, SEED, STEP = 64, 5, 8, 48, 4, 1234, 7


def test_oracle_reproduces_golden_bitwise():
    import sys

    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden

    fresh = make_golden.build()
    assert set(fresh) == set(G.files)
    for k in G.files:
        assert np.array_equal(np.asarray(fresh[k]), G[k]), k


@pytest.mark.parametrize("model,loss,seed", [("TransE", "nll", 5), ("TransE", "pairwise", 511), ("RotatE", "self_adversarial", 340),
                                             ("RotatE", "nll", 2047), ("RotatE", "nll", 2048 + 1777)])
def test_oracle_reproduces_learning_golden(model, loss, seed):
    """tests/golden/learning_mrr_v1.npz (the oracle's side of test_gpu_learning's many-seed MRR test): a sample re-derived."""
    import sys

    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_learning_golden

    L = np.load(os.path.join(HERE, "golden", "learning_mrr_v1.npz"))
    want = {f"{m}/{ls}": n for m, ls, n in make_learning_golden.CASES}
    want.update({f"{m}/{ls}/ext": n for m, ls, _, n in make_learning_golden.EXTENSIONS})
    assert want == {k: len(L[k]) for k in L.files}
    got = make_learning_golden.one((model, loss, seed))
    n1 = want[f"{model}/{loss}"]
    ref = L[f"{model}/{loss}"][seed] if seed < n1 else L[f"{model}/{loss}/ext"][seed - n1]   # (second-stage seeds follow the first stage's)
    assert np.array_equal(np.asarray(got, dtype=np.float64), ref), (got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_hip_path_against_golden(gpu_lib, model):
    import torch

    from ampligraph_amd import _ffi
    from ampligraph_amd.engine import KgeEngine

    X = G["triples"]
    Xd = torch.as_tensor(X).cuda()
    eng = KgeEngine(model, K_UNITS, N, R, max_rel_size=R)
    eng.set_tables(G[f"{model}/ent"], G[f"{model}/rel"])
    ref = G[f"{model}/scores"]
    got = eng.score(Xd).cpu().numpy()
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 0.05 * np.sqrt(np.mean(ref.astype(np.float64) ** 2)))) < 1e-5
    assert np.array_equal(eng.sample_corruptions(Xd, ETA, SEED, STEP).cpu().numpy(), G["corruptions"])   # bit-exact
    for ls in LOSSES:
        prm = O.LOSS_DEFAULTS[ls]
        ld = _ffi.Loss(_ffi.LOSSES[ls], 0, float(prm.get("margin", 0.0)), float(prm.get("alpha", 0.0)))
        for path in ("atomic", "tiled"):
            eng.prepare_training("adam")
            eng.loss_acc.zero_()
            if path == "atomic":
                eng.train_fwdbwd(Xd, ETA, ld, SEED, STEP)
            else:
                eng.train_step_tiled(Xd, ETA, ld, _ffi.Opt(2, 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1), SEED, STEP, grad_only=True)
            L = float(eng.loss_acc[0])
            assert abs(L - float(G[f"{model}/{ls}/loss"])) <= 2e-5 * max(1.0, abs(L)), (ls, path)
            for got_t, key in ((eng.g_ent, "g_ent"), (eng.g_rel, "g_rel")):
                T = G[f"{model}/{ls}/{key}"]
                scale = np.maximum(np.abs(T).max(axis=1, keepdims=True), 1e-6 * max(np.abs(T).max(), 1e-30))
                assert (np.abs(eng.unpack(got_t).cpu().numpy() - T) / scale).max() < 4e-5, (ls, path, key)
    if model != "RotatE":
        eng.set_tables(G[f"{model}/dy_ent"], G[f"{model}/dy_rel"])
        fl = [np.unique(np.concatenate([X[(X[:, 1] == t[1]) & (X[:, 2] == t[2]), 0], [t[0]]])).astype(np.int32) for t in X]
        fo = [np.unique(np.concatenate([X[(X[:, 1] == t[1]) & (X[:, 0] == t[0]), 2], [t[2]]])).astype(np.int32) for t in X]

        def csr(f):
            lo = np.cumsum([0] + [len(a) for a in f[:-1]]).astype(np.int64)
            return (torch.as_tensor(lo).cuda(), torch.as_tensor(lo + np.array([len(a) for a in f], np.int64)).cuda(),
                    torch.as_tensor(np.concatenate(f)).cuda())

        for strat in ("worst", "best", "middle"):
            r = torch.stack([eng.rank_side(Xd, _ffi.SIDE_S, strat, csr(fl))[0], eng.rank_side(Xd, _ffi.SIDE_O, strat, csr(fo))[0]], 1)
            assert np.array_equal(r.cpu().numpy(), G[f"{model}/ranks/{strat}"]), strat   # bit-exact


@pytest.mark.gpu
def test_platt_against_golden(gpu_lib):
    import torch

    from ampligraph_amd.engine import KgeEngine

    eng = KgeEngine("DistMult", 4, 5, 2)
    _, _, labels, _, rate = O.platt_init(30, 50)
    got = eng.platt_step(torch.as_tensor(G["platt/sp"]).cuda(), torch.as_tensor(G["platt/sn"]).cuda(), -0.7, 0.2,
                         labels[0], labels[1], 50 / 30, (1 - rate) / rate)
    assert np.allclose(got, G["platt/loss_gw_gb"], rtol=2e-5, atol=1e-6)
