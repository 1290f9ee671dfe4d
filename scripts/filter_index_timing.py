"""Development: where the milliseconds of FilterIndex(..., engine=...) go (upload / device build / sync / range lookups)."""
import time

import numpy as np
import torch

from ampligraph_amd.datasets import make_synthetic_kg
from ampligraph_amd.datasets.filters import FilterIndex
from ampligraph_amd.engine import KgeEngine

d = make_synthetic_kg("synth-fb15k237", seed=0)
eng = KgeEngine("ComplEx", 200, d["n_ents"], d["n_rels"])
FilterIndex([d["test"][:8]], d["n_ents"], d["n_rels"], engine=eng)
torch.cuda.synchronize()
arrs = [d["train"], d["valid"], d["test"]]
for rep in range(3):
    t0 = time.perf_counter()
    parts = [torch.as_tensor(np.ascontiguousarray(np.asarray(a)[:, :3], dtype=np.int32)).to("cuda") for a in arrs]
    X = torch.cat(parts, 0).contiguous()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    b = {sd: eng.filter_build(X, sd, d["n_ents"], d["n_rels"]) for sd in ("s", "o")}
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fi = FilterIndex(arrs, d["n_ents"], d["n_rels"], engine=eng)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    Xd = torch.as_tensor(d["test"]).cuda()
    fi.device_filter(eng, Xd, "s"); fi.device_filter(eng, Xd, "o")
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    host = FilterIndex(arrs, d["n_ents"], d["n_rels"])
    t5 = time.perf_counter()
    print(f"rep {rep}: upload {1e3*(t1-t0):.2f} ms | 2 x filter_build {1e3*(t2-t1):.2f} | whole device FilterIndex {1e3*(t3-t2):.2f} | 2 x range lookup {1e3*(t4-t3):.2f} | host numpy build {1e3*(t5-t4):.2f}")
