"""development (GPU): one deterministic TransE / nll step, gradient-only, against oracle/train_ordered.transe_nll_step_det -- which
ingredient differs (scores, loss, entity gradient rows by kind of row, relation gradient)?"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import train_ordered as TO  # noqa: E402
from test_gpu_kernels import dense, dev, loss_desc, make_engine, rand_triples  # noqa: E402

from ampligraph_amd import _ffi  # noqa: E402

for (N, R, k, B, eta) in [(300, 6, 16, 1024, 5), (500, 7, 64, 777, 9)]:
    eng, ent, rel = make_engine("TransE", k, N, R, scale=0.3)
    rng = np.random.default_rng(1)
    X = rand_triples(rng, B, N, R)
    eng.prepare_training("adam")
    eng.loss_acc.zero_()
    ps = torch.empty(B, dtype=torch.float32, device="cuda")
    ns = torch.empty(B * eta, dtype=torch.float32, device="cuda")
    d = _ffi.Opt(_ffi.OPTIMIZERS["adam"], 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1)
    eng.g_ent.fill_(123.0)
    eng.train_step_tiled(dev(X), eta, loss_desc("nll", "sum"), d, 9, 4, grad_only=True, deterministic=True, pos_scores=ps, neg_scores=ns)
    torch.cuda.synchronize()
    L, Ge, Gr = float(eng.loss_acc[0].item()), dense(eng, eng.g_ent), dense(eng, eng.g_rel)
    st = TO.OptState(ent, rel, "adam", 1e-2)
    loss, Oe, Or = TO.transe_nll_step_det(st, X, eta, 9, 4, return_grads=True)
    P, _ = TO.transe_scores(ent[X[:, 0]], rel[X[:, 1]], ent[X[:, 2]], "quad")
    rowdiff = (Ge != Oe).any(1)
    rec = dict(shape=(N, R, k, B, eta), loss_gpu=L, loss_oracle=loss, loss_rel=abs(L - loss) / abs(loss), pos_scores_differing=int((ps.cpu().numpy() != P).sum()),
               ent_grad_elements_differing=int((Ge != Oe).sum()), ent_rows_differing=int(rowdiff.sum()), ent_max_abs=float(np.abs(Ge - Oe).max()),
               ent_max_ulp_like=float(np.max(np.abs(Ge - Oe) / np.maximum(np.abs(Oe), 1e-30) * (Ge != Oe))),
               rel_grad_elements_differing=int((Gr != Or).sum()), rel_max_abs=float(np.abs(Gr - Or).max()),
               first_bad_rows=[int(i) for i in np.where(rowdiff)[0][:8]])
    print(json.dumps(rec), flush=True)
