import time, sys, numpy as np
sys.path.insert(0, "/root/repo")
from ampligraph_amd.datasets import make_synthetic_kg
from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel
import torch
d = make_synthetic_kg()
m = ScoringBasedEmbeddingModel(eta=20, k=200, scoring_type="ComplEx", seed=0)
m.compile(optimizer="adam", loss="self_adversarial")
X = d["train"]
m.fit(X, batch_size=10000, epochs=1, verbose=False)
torch.cuda.synchronize(); t = time.time()
m.fit(X, batch_size=10000, epochs=6, initial_epoch=1, verbose=False)
torch.cuda.synchronize(); dt = time.time() - t
steps = 5 * 28
print("fit(): %.4f ms/step, %.3f G triples/s (incl. the short last batch)" % (dt / steps * 1e3, 5 * len(X) * 21 / dt / 1e9))
t = time.time(); r = m.evaluate(d["test"], use_filter={"a": d["train"], "b": d["valid"], "c": d["test"]}, corrupt_side="s,o", verbose=False); dt = time.time() - t
print("evaluate(): %.1f ms for %d ranks (host filter index included)" % (dt * 1e3, r.size))
t = time.time(); r = m.evaluate(d["test"], use_filter={"a": d["train"], "b": d["valid"], "c": d["test"]}, corrupt_side="s,o", verbose=False); dt = time.time() - t
print("evaluate() again: %.1f ms (filter index cached)" % (dt * 1e3))
t = time.time(); m.fit(X, batch_size=10000, epochs=37, initial_epoch=7, verbose=False); torch.cuda.synchronize(); dt = time.time() - t
print("fit() 30 epochs: %.4f ms/step" % (dt / (30 * 28) * 1e3))
