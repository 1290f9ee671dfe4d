"""Development: PCIe-inclusive throughput of the session layer (host triples in, loss out, synchronous per step) at C2."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from ampligraph_amd.datasets import make_synthetic_kg  # noqa: E402
from ampligraph_amd.latent_features import loss_functions, optimizers  # noqa: E402
from ampligraph_amd.session import Session  # noqa: E402

d = make_synthetic_kg()
N, R, B = d["n_ents"], d["n_rels"], 10000
rng = np.random.default_rng(0)
s = Session("ComplEx", 200, N, R, 20, loss_functions.get("self_adversarial"), optimizers.get("adam"), seed=0)
s.set_rows("ent", rng.uniform(-0.02, 0.02, (N, 400)).astype(np.float32))
s.set_rows("rel", rng.uniform(-0.1, 0.1, (R, 400)).astype(np.float32))
X = d["train"]
nb = len(X) // B
for i in range(10):
    s.train_step(X[(i % nb) * B:(i % nb + 1) * B])
t0 = time.perf_counter()
n = 200
for i in range(n):
    s.train_step(X[(i % nb) * B:(i % nb + 1) * B])
dt = (time.perf_counter() - t0) / n
print(f"session train_step: {dt * 1e3:.4f} ms/step = {B * 21 / dt / 1e9:.3f} G triples/s (H2D of the batch + D2H of the loss + sync per step)")
T = d["test"]
t0 = time.perf_counter()
sc = s.score(T)
print(f"session score: {(time.perf_counter() - t0) * 1e3:.2f} ms for {len(T)} triples")
