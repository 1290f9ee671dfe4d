"""Development helper: time the fused train kernel alone (HIP events), C2 shape, under AMDKGE_DEBUG ablations."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ampligraph_amd import _ffi
from ampligraph_amd.engine import KgeEngine
from ampligraph_amd.datasets import make_synthetic_kg

def run(model="ComplEx", k=200, eta=20, B=10000, reps=20, opt=True):
    d = make_synthetic_kg()
    N, R = d["n_ents"], d["n_rels"]
    eng = KgeEngine(model, k, N, R, max_rel_size=R)
    rng = np.random.default_rng(0)
    eng.set_tables(rng.uniform(-.02, .02, (N, eng.K)).astype(np.float32), rng.uniform(-.1, .1, (R, eng.K)).astype(np.float32))
    eng.prepare_training("adam")
    loss = _ffi.Loss(_ffi.LOSSES["self_adversarial"], 0, 3.0, 0.5)
    X = torch.as_tensor(d["train"]).cuda()
    res = {}
    for s in range(3):
        eng.train_fwdbwd(X[s*B:(s+1)*B], eta, loss, 0, s)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = to = 0.0
    for s in range(reps):
        e0.record(); eng.train_fwdbwd(X[(s%27)*B:(s%27+1)*B], eta, loss, 0, s); e1.record()
        eng.opt_step(_ffi.Opt(2, 2, 1e-3, .9, .999, 1e-7, 0.0, s+1)); e2.record()
        torch.cuda.synchronize(); tf += e0.elapsed_time(e1); to += e1.elapsed_time(e2)
    return tf/reps, to/reps

if __name__ == "__main__":
    f, o = run()
    print(json.dumps({"dbg": os.environ.get("AMDKGE_DEBUG", "0"), "fwdbwd_ms": f, "opt_ms": o}))
