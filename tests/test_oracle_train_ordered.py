"""oracle/train_ordered.py (the train step of TransE / pairwise / Adam in the kernels' declared fp32 order) pinned on the CPU
against oracle/kge_oracle.py: where every fp32 operation is exact (dyadic tables) the two must agree bit for bit; on real-valued
tables they may differ only where a hinge term or a sign sits within rounding of its boundary; its Adam against the fp64-formed
rule of the main oracle."""
import numpy as np
import pytest

from oracle import kge_oracle as O
from oracle import train_ordered as TO


def _problem(rng, N=40, R=3, K=16, B=120, dyadic=False):
    if dyadic:
        ent = (rng.integers(-8, 9, size=(N, K)) / 8.0).astype(np.float32)
        rel = (rng.integers(-8, 9, size=(R, K)) / 8.0).astype(np.float32)
    else:
        ent = (rng.normal(size=(N, K)) * 0.4).astype(np.float32)
        rel = (rng.normal(size=(R, K)) * 0.4).astype(np.float32)
    X = np.stack([rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)], 1).astype(np.int32)
    return ent, rel, X


@pytest.mark.parametrize("layout", ["quad", "unit"])
@pytest.mark.parametrize("K,eta", [(16, 5), (52, 3), (200, 7), (256, 70), (352, 4)])
def test_ordered_step_equals_oracle_on_dyadic_tables(K, eta, layout):
    rng = np.random.default_rng(K)
    ent, rel, X = _problem(rng, K=K, dyadic=True)
    st = TO.AdamState(ent, rel, 1e-2)
    loss, Ge, Gr = TO.transe_pairwise_step(st, X, eta, 11, 3, margin=1.0, return_grads=True, layout=layout)
    negs = O.generate_corruptions(X, ent.shape[0], eta, 11, 3)
    tot, Re, Rr, _ = O.dense_gradients("TransE", ent, rel, X, negs, eta, "pairwise", None, "sum", rel.shape[0])
    assert np.array_equal(Ge, Re) and np.array_equal(Gr, Rr)
    assert loss == float(tot) or abs(loss - float(tot)) <= 1e-6 * abs(float(tot))   # (the oracle rounds its total to fp32)


def test_ordered_step_differs_from_fp64_oracle_only_at_decision_boundaries():
    rng = np.random.default_rng(1)
    ent, rel, X = _problem(rng, N=300, K=64, B=2000)
    st = TO.AdamState(ent, rel, 1e-2)
    loss, Ge, Gr = TO.transe_pairwise_step(st, X, 5, 7, 0, return_grads=True)
    negs = O.generate_corruptions(X, ent.shape[0], 5, 7, 0)
    tot, Re, Rr, _ = O.dense_gradients("TransE", ent, rel, X, negs, 5, "pairwise", None, "sum", rel.shape[0])
    assert abs(loss - float(tot)) <= 2e-6 * abs(float(tot))
    # a flipped hinge term moves K entries of three rows by one: a handful among 300 x 64, never more than a few units
    assert np.mean(Ge != Re) < 0.02 and np.abs(Ge - Re).max() <= 3 and np.abs(Gr - Rr).max() <= 3 * 64


def test_ordered_adam_is_the_oracles_rule():
    rng = np.random.default_rng(2)
    ent, rel, _ = _problem(rng)
    a = TO.AdamState(ent, rel, 2e-2)
    b = O.TrainState(ent, rel, "adam", 2e-2)
    for t in range(5):
        Ge = rng.integers(-3, 4, size=ent.shape).astype(np.float64)
        Gr = rng.integers(-3, 4, size=rel.shape).astype(np.float64)
        a.apply(Ge, Gr)
        O.apply_optimizer(b, Ge, Gr)
        # the slots are the same bits; the main oracle forms lr_t from the python-float betas, the kernels (and this module) from
        # the fp32 ones (1 - 0.999f differs from 1 - 0.999 by 1.3e-5 relative): lr x 6.4e-6 per step on x
        assert np.array_equal(a.m[0], b.slots["m_e"]) and np.array_equal(a.v[0], b.slots["v_e"])
        assert np.abs(a.ent - b.ent).max() <= (t + 1) * 2e-2 * 1e-5 + 2e-7


def test_replay_learning_runs_and_learns():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from planted import LEARNING, planted_kg

    from ampligraph_amd.latent_features.initializers import initialise

    hist, st, Xi, ti = TO.replay_learning("TransE", "pairwise", 0, LEARNING, planted_kg, initialise, epochs=6)
    assert hist[-1] < 0.8 * hist[0]
    # the fp64 oracle on the same schedule: the early epochs agree closely (before the trajectories can part)
    from planted import oracle_learning_run

    h64, _, _ = oracle_learning_run("TransE", "pairwise", 0, epochs=6)
    assert abs(hist[0] - h64[0]) <= 1e-5 * abs(h64[0]) and np.max(np.abs(hist - h64) / np.abs(h64)) < 5e-3


@pytest.mark.parametrize("kind,hp,okind,ohp", [("sgd", (None, None), "sgd", {}), ("adagrad", (None, None), "adagrad", {}),
                                               ("momentum", (0.7, 1.0), "momentum", {"momentum": 0.7, "nesterov": True}),
                                               ("rmsprop", (0.9, 0.0), "rmsprop", {}), ("rmsprop_mom", (0.9, 0.5), "rmsprop_mom", {"momentum": 0.5}),
                                               ("adadelta", (0.95, 0.0), "adadelta", {}), ("adamax", (0.9, 0.999), "adamax", {})])
def test_ordered_update_rules_are_the_oracles_rules(kind, hp, okind, ohp):
    """OptState restates kge_opt.h's opt_elem per rule in fp32; the main oracle states the same Keras-legacy rules: same values up
    to the last bits (they differ in where hyper-parameter constants are rounded)."""
    rng = np.random.default_rng(3)
    ent, rel, _ = _problem(rng)
    a = TO.OptState(ent, rel, kind, 1e-2, *hp)
    b = O.TrainState(ent, rel, okind, 1e-2, **ohp)
    for t in range(4):
        Ge = rng.integers(-3, 4, size=ent.shape).astype(np.float64)
        Gr = rng.integers(-3, 4, size=rel.shape).astype(np.float64)
        a.apply(Ge, Gr)
        O.apply_optimizer(b, Ge, Gr)
        assert np.abs(a.ent - b.ent).max() <= 2e-6 * max(1.0, np.abs(b.ent).max()), (kind, t, np.abs(a.ent - b.ent).max())


def test_ordered_absolute_margin_equals_oracle_on_dyadic_tables():
    rng = np.random.default_rng(4)
    ent, rel, X = _problem(rng, K=32, dyadic=True)
    st = TO.OptState(ent, rel, "sgd", 1e-2)
    loss, Ge, Gr = TO.transe_pairwise_step(st, X, 4, 11, 3, margin=1.0, return_grads=True, loss="absolute_margin")
    negs = O.generate_corruptions(X, ent.shape[0], 4, 11, 3)
    tot, Re, Rr, _ = O.dense_gradients("TransE", ent, rel, X, negs, 4, "absolute_margin", None, "sum", rel.shape[0])
    assert np.array_equal(Ge, Re) and np.array_equal(Gr, Rr) and abs(loss - float(tot)) <= 1e-6 * abs(float(tot))


def test_lane_layouts_against_an_independent_restatement():
    """_lane_sums: both row layouts against a scalar loop over (lane, chunk) written from the kernels' index arithmetic."""
    rng = np.random.default_rng(8)
    for K in (4, 16, 52, 128, 200, 260, 512):
        a = np.abs(rng.normal(size=(3, K))).astype(np.float32)
        for layout in ("quad", "unit"):
            ref = np.zeros((3, 64), dtype=np.float32)
            for row in range(3):
                for lane in range(64):
                    part = np.float32(0)
                    if layout == "unit":   # VEC = 1: unit index = lane + 64 c
                        for c in range((K + 63) // 64):
                            u = lane + 64 * c
                            if u < K:
                                part = np.float32(part + np.float32(np.float32(0) + a[row, u]))
                    else:                  # VEC = 4: quad index = lane + 64 c, units 4 q .. 4 q + 3 added in order
                        for c in range((K // 4 + 63) // 64):
                            q = lane + 64 * c
                            if q < K // 4:
                                acc = np.float32(0)
                                for u in range(4):
                                    acc = np.float32(acc + a[row, 4 * q + u])
                                part = np.float32(part + acc)
                    ref[row, lane] = part
            assert np.array_equal(TO._lane_sums(a, layout), ref), (K, layout)


def test_declared_transcendentals_are_accurate():
    """det_exp / det_log12 / det_sig_logsig (the deterministic mode's CPU-reproducible loss terms): within a few 1e-7 of libm."""
    x = np.linspace(-80, 80, 100001).astype(np.float32)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(TO.det_exp(x).astype(np.float64) - ref) / ref) < 3e-7
    u = np.linspace(1, 2, 50001).astype(np.float32)
    assert np.max(np.abs(TO.det_log12(u).astype(np.float64) - np.log(u.astype(np.float64)))) < 3e-7
    y = np.linspace(-75, 75, 30001).astype(np.float32)
    sg, ls = TO.det_sig_logsig(y)
    rs, rl = 1 / (1 + np.exp(-y.astype(np.float64))), -np.logaddexp(0, y.astype(np.float64))
    assert np.max(np.abs(sg - rs) / rs) < 5e-7 and np.max(np.abs(ls - rl) / np.maximum(np.abs(rl), 1e-300)) < 1e-6


@pytest.mark.parametrize("loss", ["nll", "self_adversarial", "multiclass_nll"])
@pytest.mark.parametrize("K,eta", [(16, 5), (64, 13), (200, 3)])
def test_ordered_transcendental_loss_steps_are_the_oracles_steps_up_to_rounding(K, eta, loss):
    """transe_step_det (the single-pass protocol of the forward kernel restated group by group, online softmax included) against
    the fp64 oracle's loss and dense gradients."""
    rng = np.random.default_rng(K)
    ent, rel, X = _problem(rng, N=60, K=K, B=300)
    st = TO.OptState(ent, rel, "adam", 1e-2)
    loss_v, Ge, Gr = TO.transe_step_det(st, X, eta, 5, 2, loss, return_grads=True)
    negs = O.generate_corruptions(X, ent.shape[0], eta, 5, 2)
    tot, Re, Rr, _ = O.dense_gradients("TransE", ent, rel, X, negs, eta, loss, None, "sum", rel.shape[0])
    loss = loss_v
    assert abs(loss - float(tot)) <= 2e-6 * abs(float(tot))
    # (+ one fp32 ulp of 1: with every score clipped out of range the gradient is -1 + eP / Z = 0 up to the rounding of the quotient)
    assert np.abs(Ge - Re).max() <= 2e-6 * np.abs(Re).max() + 1.3e-7 and np.abs(Gr - Rr).max() <= 2e-6 * np.abs(Rr).max() + 2e-5


def test_fmaf32_is_libm_fmaf():
    """The numpy restatement of the one-rounding fma (round-to-odd in fp64) against libm's fmaf, incl. products that cancel the
    addend almost exactly (where a double rounding would show)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    rng = np.random.default_rng(0)
    a = rng.standard_normal(20000).astype(np.float32)
    b = rng.standard_normal(20000).astype(np.float32)
    c = rng.standard_normal(20000).astype(np.float32)
    c[::3] = -(a[::3] * b[::3]).astype(np.float32)                      # near-cancellation
    c[1::7] = (c[1::7] * np.float32(1e-7)).astype(np.float32)              # addend far below the product
    # exact ties of the fp64 sum at an fp32 rounding boundary: a * b = 1 + 2^-24 (odd neighbour decides), c tiny of either sign
    a[:4] = np.float32(1.0 + 2.0 ** -12); b[:4] = np.float32(1.0 - 2.0 ** -12 + 2.0 ** -23)
    c[:4] = np.array([2.0 ** -60, -2.0 ** -60, 2.0 ** -90, -2.0 ** -90], dtype=np.float32)
    got = TO.fmaf32(a, b, c)
    ref = np.array([libm.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("loss", ["nll", "self_adversarial", "multiclass_nll"])
@pytest.mark.parametrize("model,K,eta", [("DistMult", 16, 5), ("ComplEx", 32, 5), ("ComplEx", 400, 3), ("HolE", 64, 13), ("DistMult", 200, 7),
                                         ("DistMult", 400, 7), ("ComplEx", 704, 5), ("ComplEx", 1200, 3), ("DistMult", 1200, 4)])
def test_ordered_trilinear_steps_are_the_oracles_steps_up_to_rounding(model, K, eta, loss):
    """trilinear_step_det (side rows, fmaf score chains, sum_j c_j e_j with the online softmax, grad_unit transforms, sorted tile
    sums) against the fp64 oracle's loss and dense gradients."""
    rng = np.random.default_rng(K + eta)
    ent, rel, X = _problem(rng, N=60, K=K, B=300)
    ent *= np.float32(0.5); rel *= np.float32(0.5)
    st = TO.OptState(ent, rel, "adam", 1e-2)
    loss_v, Ge, Gr = TO.trilinear_step_det(model, st, X, eta, 5, 2, loss, return_grads=True)
    negs = O.generate_corruptions(X, ent.shape[0], eta, 5, 2)
    tot, Re, Rr, _ = O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", rel.shape[0])
    assert abs(loss_v - float(tot)) <= 4e-6 * abs(float(tot))
    assert np.abs(Ge - Re).max() <= 4e-6 * np.abs(Re).max() + 1.3e-7 and np.abs(Gr - Rr).max() <= 4e-6 * np.abs(Rr).max() + 2e-5


@pytest.mark.parametrize("loss", ["nll", "self_adversarial", "multiclass_nll"])
@pytest.mark.parametrize("K,eta", [(32, 5), (400, 4), (128, 11), (200 * 2 + 8, 5), (2000, 7)])
def test_ordered_rotate_steps_are_the_oracles_steps_up_to_rounding(K, eta, loss):
    """rotate_step_det (phases, unit vectors, per-side sums in groups of three, the gradient transform, the tile entries on the
    owner's live rows) against the fp64 oracle's loss and dense gradients."""
    rng = np.random.default_rng(K + eta)
    ent, rel, X = _problem(rng, N=60, K=K, B=300)
    st = TO.OptState(ent, rel, "adam", 1e-2)
    R = rel.shape[0]
    loss_v, Ge, Gr = TO.rotate_step_det(st, X, eta, 5, 2, loss, max_rel_size=R, return_grads=True)
    negs = O.generate_corruptions(X, ent.shape[0], eta, 5, 2)
    tot, Re, Rr, _ = O.dense_gradients("RotatE", ent, rel, X, negs, eta, loss, None, "sum", R)
    assert abs(loss_v - float(tot)) <= 4e-6 * abs(float(tot))
    assert np.abs(Ge - Re).max() <= 1e-5 * np.abs(Re).max() + 1.3e-7 and np.abs(Gr - Rr).max() <= 1e-5 * np.abs(Rr).max() + 2e-5
