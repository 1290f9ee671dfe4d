"""development (GPU): why do RotatE k=1000 whole steps with sgd+momentum / rmsprop / rmsprop+momentum leave the bars of
tests/test_gpu_tile_direct.py::test_direct_step_in_place_parity?  Prints, per rule and step, where the outliers sit."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import kge_oracle as O  # noqa: E402
from test_gpu_kernels import dense, dev, loss_desc, make_engine, make_optimizer, rand_triples, run_tiled_grads  # noqa: E402

from ampligraph_amd import _ffi  # noqa: E402

lib = _ffi.lib()
model, k, reg = "RotatE", 1000, (3, 1e-2)
N, R, B, eta = 120, 4, 60, 3
out = []
LAZY = len(sys.argv) > 1 and sys.argv[1] == "lazy"
for opt in (["rmsprop", "rmsprop+momentum", "sgd+momentum", "adam"] if LAZY else ["sgd", "sgd+momentum", "rmsprop", "rmsprop+momentum", "adam"]):
    for direct in ((True,) if LAZY else (True, False)):
        lib.amdkge_set_tile_direct(1 if direct else 0)
        eng, ent, rel = make_engine(model, k, N, R, scale=0.08)
        w, mk = make_optimizer(opt.split("+")[0], {"momentum": 0.7} if "+" in opt else {})
        w.lazy = LAZY
        hist = np.zeros((3, N), dtype=bool)
        eng.prepare_training(w.name)
        st = mk(ent, rel)
        rng = np.random.default_rng(6)
        oreg = dict(p=reg[0], lam_e=reg[1], lam_r=reg[1])
        for t in range(1, 4):
            X = rand_triples(rng, B, N, R)
            negs = O.generate_corruptions(X, N, eta, 77, t)
            _, Ge, Gr, _ = O.dense_gradients(model, st.ent, st.rel, X, negs, eta, "self_adversarial", None, "sum", R, oreg)
            eng.loss_acc.zero_()
            eng.train_step_tiled(dev(X), eta, loss_desc("self_adversarial"), w.to_ffi(t, reg[0]), 77, t, reg_e=reg[1], reg_r=reg[1])
            O.train_step(st, model, X, eta, "self_adversarial", 77, t, max_rel_size=R, reg=oreg, lazy=LAZY)
            torch.cuda.synchronize()
            e, r = eng.get_tables()
            err = np.abs(e - st.ent)
            bad = err > 1e-5 + 1e-4 * np.abs(st.ent)
            touched = np.zeros(N, dtype=bool)
            touched[np.concatenate([X[:, 0], X[:, 2], negs[:, 0], negs[:, 2]])] = True
            gabs = np.abs(Ge)
            hist[t - 1] = touched
            bad_rows = bad.mean(1)
            rec = dict(lazy=LAZY, rows_with_bad=int((bad_rows > 0).sum()), bad_rows_top=[(int(i), float(bad_rows[i]), [bool(h) for h in hist[:t, i]]) for i in np.argsort(-bad_rows)[:6]],
                       bad_by_history={str(tuple(int(v) for v in key)): float(bad[np.all(hist[:t].T == np.array(key, dtype=bool), axis=1)].mean()) if np.any(np.all(hist[:t].T == np.array(key, dtype=bool), axis=1)) else None
                                       for key in ([(1, 1, 1), (0, 1, 1), (1, 0, 1), (0, 0, 1), (1, 1, 0)] if t == 3 else [])},
                       opt=opt, direct=direct, t=t, frac_inside=float(1 - bad.mean()), max_err=float(err.max()),
                       bad_in_touched_rows=float(bad[touched].mean()), bad_in_untouched_rows=float(bad[~touched].mean()) if (~touched).any() else None,
                       err_q=[float(np.quantile(err, q)) for q in (0.5, 0.9, 0.99, 0.999)],
                       g_abs_q_at_bad=[float(np.quantile(gabs[bad], q)) for q in (0.1, 0.5, 0.9)] if bad.any() else None,
                       g_abs_q_all=[float(np.quantile(gabs, q)) for q in (0.1, 0.5, 0.9)],
                       bad_first_half=float(bad[:, :k].mean()), bad_second_half=float(bad[:, k:].mean()))
            out.append(rec)
            print(json.dumps(rec), flush=True)
# the gradient itself, GPU vs oracle, on the first batch
lib.amdkge_set_tile_direct(1)
eng, ent, rel = make_engine(model, k, N, R, scale=0.08)
rng = np.random.default_rng(6)
X = rand_triples(rng, B, N, R)
negs = O.generate_corruptions(X, N, eta, 77, 1)
_, Te, Tr, _ = O.dense_gradients(model, ent, rel, X, negs, eta, "self_adversarial", None, "sum", R)
L, Ge, Gr, ps, ns = run_tiled_grads(eng, X, eta, "self_adversarial", "sum", 77, 1)
d = np.abs(Ge - Te)
print(json.dumps(dict(grad_abs_err_q=[float(np.quantile(d, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)],
                      grad_abs_q=[float(np.quantile(np.abs(Te), q)) for q in (0.5, 0.9, 0.99, 1.0)],
                      rel_err_gt_1e4=float((d > 1e-4 * np.abs(Te).max()).mean()))))
