"""development: evaluate()-shaped rank_side calls at the C2 shape (random tables) -- run under rocprofv3 for per-kernel times"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ampligraph_amd import _ffi
from ampligraph_amd.engine import KgeEngine
N, R, k, n = 14505, 237, 200, 20438
rng = np.random.default_rng(0)
eng = KgeEngine("ComplEx", k, N, R, max_rel_size=R)
eng.set_tables((rng.normal(size=(N, eng.K)) * 0.25).astype(np.float32), (rng.normal(size=(R, eng.K)) * 0.25).astype(np.float32))
X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1).astype(np.int32)
Xd = torch.as_tensor(X).cuda()
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for side in (_ffi.SIDE_S, _ffi.SIDE_O):
        eng.rank_side(Xd, side, "worst")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two sides %.3f ms" % (dt * 1e3), eng.screen_stats())
