"""development: where model.evaluate() spends its time (host side), after a short fit at the C2 shape"""
import cProfile, pstats, time, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from ampligraph_amd.datasets import make_synthetic_kg
from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel
d = make_synthetic_kg()
m = ScoringBasedEmbeddingModel(eta=20, k=200, scoring_type="ComplEx", seed=0)
m.compile(optimizer="adam", loss="self_adversarial")
m.fit(d["train"], batch_size=10000, epochs=2, verbose=False)
flt = {"a": d["train"], "b": d["valid"], "c": d["test"]}
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = m.evaluate(d["test"], use_filter=flt, corrupt_side="s,o", verbose=False)
    torch.cuda.synchronize(); print("evaluate() call %d: %.2f ms for %d ranks" % (i, (time.perf_counter() - t) * 1e3, r.size), flush=True)
flt2 = {"a": d["train"], "c": d["test"]}   # another filter set: the cache misses, nothing is a first use of the process any more
torch.cuda.synchronize(); t = time.perf_counter()
r = m.evaluate(d["test"], use_filter=flt2, corrupt_side="s,o", verbose=False)
torch.cuda.synchronize(); print("evaluate() call with a NEW filter set (index rebuilt): %.2f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    r = m.evaluate(d["test"], use_filter=flt, corrupt_side="s,o", verbose=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
