"""Generates tests/golden/golden_v1.npz: frozen outputs of the CPU oracle (oracle/kge_oracle.py, itself pinned by the
reference's known-answer tests in tests/test_oracle_kat.py) on small seeded inputs.  The reference cannot be imported
here (TensorFlow is not installable), so these vectors are oracle outputs, not reference outputs; they (1) freeze the
oracle against silent drift (tests/test_golden.py, CPU) and (2) give the -m gpu tests a fixture that does not depend on
the oracle's code at test time.

    python tests/golden/make_golden.py        # rewrites golden_v1.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kge_oracle as O  # noqa: E402

MODELS = ["TransE", "DistMult", "ComplEx", "HolE", "RotatE"]
LOSSES = ["pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll"]
N, R, K_UNITS, B, ETA, SEED, STEP = 64, 5, 8, 48, 4, 1234, 7


def build():
    out = {}
    rng = np.random.default_rng(20260922)
    X = np.stack([rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)], 1).astype(np.int32)
    X[:3, 2] = X[:3, 0]
    out["triples"] = X
    negs = O.generate_corruptions(X, N, ETA, SEED, STEP)
    out["corruptions"] = negs
    for m in MODELS:
        K = O.internal_k(m, K_UNITS)
        ent = (rng.normal(size=(N, K)) * 0.4).astype(np.float32)
        rel = (rng.normal(size=(R, K)) * 0.4).astype(np.float32)
        out[f"{m}/ent"], out[f"{m}/rel"] = ent, rel
        out[f"{m}/scores"] = O.compute_scores(m, *O.lookup(ent, rel, X), max_rel_size=R)
        for ls in LOSSES:
            tot, Ge, Gr, (sp, sn, per) = O.dense_gradients(m, ent, rel, X, negs, ETA, ls, None, "sum", R)
            out[f"{m}/{ls}/loss"] = np.float64(per.astype(np.float64).sum())
            out[f"{m}/{ls}/g_ent"], out[f"{m}/{ls}/g_rel"] = Ge, Gr
        st = O.TrainState(ent, rel, "adam", 1e-2)
        O.train_step(st, m, X, ETA, "self_adversarial", SEED, STEP, max_rel_size=R, reg=dict(p=2, lam_e=1e-3, lam_r=1e-3))
        out[f"{m}/adam_step/ent"], out[f"{m}/adam_step/rel"] = st.ent, st.rel
        # ranks on dyadic-rational tables (fp32 arithmetic exact in any order => ranks are bit-exact targets)
        de = (rng.integers(-4, 5, size=(N, K)) / 4.0).astype(np.float32)
        dr = (rng.integers(-4, 5, size=(R, K)) / 4.0).astype(np.float32)
        out[f"{m}/dy_ent"], out[f"{m}/dy_rel"] = de, dr
        if m != "RotatE":   # RotatE's cos/sin are not exact
            fl = [np.unique(np.concatenate([X[(X[:, 1] == t[1]) & (X[:, 2] == t[2]), 0], [t[0]]])) for t in X]
            fo = [np.unique(np.concatenate([X[(X[:, 1] == t[1]) & (X[:, 0] == t[0]), 2], [t[2]]])) for t in X]
            for strat in ("worst", "best", "middle"):
                out[f"{m}/ranks/{strat}"] = O.evaluate_ranks(m, de, dr, X, fl, fo, "s,o", strat, max_rel_size=R)
    sp, sn = rng.normal(size=30).astype(np.float32) * 2, rng.normal(size=50).astype(np.float32) * 2 - 1
    _, _, labels, _, rate = O.platt_init(30, 50)
    out["platt/sp"], out["platt/sn"] = sp, sn
    out["platt/loss_gw_gb"] = np.array(O.platt_loss_and_grads(sp, sn, -0.7, 0.2, labels, rate))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz"), **build())
    print("written")
