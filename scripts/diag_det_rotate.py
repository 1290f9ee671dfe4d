"""development (GPU): one deterministic RotatE step, gradient-only, against oracle/train_ordered.rotate_step_det -- which ingredient
differs (positive scores, corruption scores, loss, entity gradient rows by kind of row, relation gradient)?"""
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

for loss in ("nll", "self_adversarial"):
    for (N, R, k, B, eta) in [(300, 6, 16, 1024, 5), (500, 7, 64, 777, 9)]:
        eng, ent, rel = make_engine("RotatE", k, N, R, scale=0.3)
        rng = np.random.default_rng(1)
        X = rand_triples(rng, B, N, R)
        eng.prepare_training("adam")
        eng.loss_acc.zero_()
        ps = torch.empty(B, dtype=torch.float32, device="cuda")
        ns = torch.empty(B * eta, dtype=torch.float32, device="cuda")
        d = _ffi.Opt(_ffi.OPTIMIZERS["adam"], 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1)
        eng.g_ent.fill_(123.0)
        eng.train_step_tiled(dev(X), eta, loss_desc(loss, "sum"), d, 9, 4, grad_only=True, deterministic=True, pos_scores=ps, neg_scores=ns)
        torch.cuda.synchronize()
        L, Ge, Gr = float(eng.loss_acc[0].item()), dense(eng, eng.g_ent), dense(eng, eng.g_rel)
        st = TO.OptState(ent, rel, "adam", 1e-2)
        dbg = {}
        lo, Oe, Or = TO.rotate_step_det(st, X, eta, 9, 4, loss, max_rel_size=R, return_grads=True, debug=dbg)
        nsg = ns.cpu().numpy().reshape(eta, B).T
        own = np.zeros(N, dtype=bool); own[X[:, 0]] = True; own[X[:, 2]] = True
        negd = np.zeros(N, dtype=bool); negd[dbg["repl"].ravel()] = True
        rowdiff = (Ge != Oe).any(1)
        rec = dict(loss=loss, shape=(N, R, k, B, eta), loss_gpu=L, loss_oracle=lo, loss_rel=abs(L - lo) / abs(lo),
                   pos_scores_differing=int((ps.cpu().numpy() != dbg["P"]).sum()), neg_scores_differing=int((nsg != dbg["nsc"]).sum()),
                   neg_scores_max_abs=float(np.abs(nsg - dbg["nsc"]).max()),
                   ent_grad_elements_differing=int((Ge != Oe).sum()), ent_rows_differing=int(rowdiff.sum()),
                   rows_differing_only_own=int((rowdiff & own & ~negd).sum()), rows_differing_only_neg=int((rowdiff & negd & ~own).sum()),
                   rows_only_own=int((own & ~negd).sum()), rows_only_neg=int((negd & ~own).sum()),
                   ent_max_abs=float(np.abs(Ge - Oe).max()),
                   ent_max_rel=float(np.max(np.abs(Ge - Oe) / np.maximum(np.abs(Oe), 1e-30) * (Ge != Oe))),
                   rel_grad_elements_differing=int((Gr != Or).sum()), rel_max_abs=float(np.abs(Gr - Or).max()),
                   rel_second_half_nonzero=int((Gr[:, k:] != 0).sum()))
        print(json.dumps(rec), flush=True)
