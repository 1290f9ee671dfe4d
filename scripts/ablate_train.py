"""Development helper: C2 tiled step under AMDKGE_DEBUG ablations / alternative builds (AMDKGE_LIB)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, numpy as np, torch
sys.path.insert(0, %r)
from ampligraph_amd import _ffi
from ampligraph_amd.datasets import make_synthetic_kg
from ampligraph_amd.engine import KgeEngine
d = make_synthetic_kg(popularity=os.environ.get("POP", "uniform")); N, R = d["n_ents"], d["n_rels"]
model, k, eta, B = os.environ.get("M", "ComplEx"), int(os.environ.get("K", 200)), int(os.environ.get("ETA", 20)), 10000
eng = KgeEngine(model, k, N, R, max_rel_size=R)
rng = np.random.default_rng(0)
eng.set_tables(rng.uniform(-.02, .02, (N, eng.K)).astype(np.float32), rng.uniform(-.1, .1, (R, eng.K)).astype(np.float32))
eng.prepare_training("adam")
loss = _ffi.Loss(_ffi.LOSSES["self_adversarial"], 0, 3.0, 0.5)
X = torch.as_tensor(d["train"]).cuda(); nb = X.shape[0] // B
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = 0.0; reps = 30
for s in range(-3, reps):
    xb = X[(s %% nb) * B:(s %% nb + 1) * B]
    o = _ffi.Opt(2, 2, 1e-3, .9, .999, 1e-7, 0.0, s + 4)
    e0.record(); eng.train_step_tiled(xb, eta, loss, o, 0, s + 3); e1.record(); torch.cuda.synchronize()
    if s >= 0: t += e0.elapsed_time(e1)
print(json.dumps({"cfg": os.environ.get("CFG"), "tiled_train_us": 1000 * t / reps}))
''' % ROOT
for cfg in sys.argv[1:]:
    env = dict(os.environ, CFG=cfg)
    for kv in cfg.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    subprocess.run([sys.executable, "-c", CHILD], env=env)
