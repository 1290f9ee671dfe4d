"""Development helper: time one training step on the C2 shape (HIP events), both paths:
  atomic : train_fwdbwd_kernel (atomic scatter) + dense optimizer sweep of both tables
  tiled  : forward+staging kernel + tile_backward_kernel (owner-computes, optimizer fused) + relation sweep"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ampligraph_amd import _ffi
from ampligraph_amd.datasets import make_synthetic_kg
from ampligraph_amd.engine import KgeEngine


def run(model="ComplEx", k=200, eta=20, B=10000, reps=30, loss_name="self_adversarial", kg="synth-fb15k237"):
    d = make_synthetic_kg(kg) if kg != "synth-fb15k237" else make_synthetic_kg()
    N, R = d["n_ents"], d["n_rels"]
    eng = KgeEngine(model, k, N, R, max_rel_size=R)
    rng = np.random.default_rng(0)
    eng.set_tables(rng.uniform(-.02, .02, (N, eng.K)).astype(np.float32), rng.uniform(-.1, .1, (R, eng.K)).astype(np.float32))
    eng.prepare_training("adam")
    loss = _ffi.Loss(_ffi.LOSSES[loss_name], 0, 3.0, 0.5)
    X = torch.as_tensor(d["train"]).cuda()
    nb = X.shape[0] // B
    out = {"model": model, "k": k, "eta": eta, "B": B, "N": N}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for path in ("atomic", "tiled"):
        if path == "tiled" and not eng.tiled_supported(B, eta):
            continue
        ta = tb = 0.0
        for s in range(-3, reps):
            xb = X[(s % nb) * B:(s % nb + 1) * B]
            o = _ffi.Opt(2, 2, 1e-3, .9, .999, 1e-7, 0.0, s + 4)
            ev[0].record()
            if path == "atomic":
                eng.train_fwdbwd(xb, eta, loss, 0, s + 3)
                ev[1].record()
                eng.opt_step(o)
            else:
                eng.train_step_tiled(xb, eta, loss, o, 0, s + 3)
                ev[1].record()
            ev[2].record()
            torch.cuda.synchronize()
            if s >= 0:
                ta += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
        out[path] = {"train_ms": ta / reps, "opt_ms": tb / reps, "step_ms": (ta + tb) / reps}
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        print(json.dumps(run("DistMult", 400, 30, 10000)))
        print(json.dumps(run("TransE", 52, 5, 10000, loss_name="pairwise")))
        print(json.dumps(run("RotatE", 200, 20, 10000)))
