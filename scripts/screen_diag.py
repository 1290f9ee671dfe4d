"""development: recheck statistics of the screening pass on the test tables of tests/test_gpu_rank_screen.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ampligraph_amd import _ffi
from ampligraph_amd.engine import KgeEngine
lib = _ffi.lib()
for model, k, N, n, tables in [("HolE", 350, 3000, 700, "gaussian"), ("ComplEx", 200, 14505, 2000, "gaussian"), ("DistMult", 400, 9000, 1500, "wild"), ("ComplEx", 200, 14505, 2000, "ties")]:
    R = 11
    rng = np.random.default_rng(k + N)
    eng = KgeEngine(model, k, N, R, max_rel_size=R)
    K = eng.K
    if tables == "gaussian":
        ent = (rng.normal(size=(N, K)) * 0.25).astype(np.float32); rel = (rng.normal(size=(R, K)) * 0.25).astype(np.float32)
    elif tables == "wild":
        ent = (rng.normal(size=(N, K)) * np.exp(rng.uniform(-14, 14, size=(N, 1))) * np.exp(rng.uniform(-7, 7, size=(N, K)))).astype(np.float32)
        ent[rng.random((N, K)) < 0.1] = 0.0; ent[5, 3] = 3e18
        rel = (rng.normal(size=(R, K)) * np.exp(rng.uniform(-3, 3, size=(R, K)))).astype(np.float32)
    else:
        ent = (rng.integers(-4, 5, size=(N, K)) / 8.0).astype(np.float32); rel = (rng.integers(-2, 3, size=(R, K)) / 4.0).astype(np.float32)
    eng.set_tables(ent, rel)
    X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1).astype(np.int32)
    Xd = torch.as_tensor(X).cuda()
    for side in (_ffi.SIDE_S, _ffi.SIDE_O):
        lib.amdkge_set_rank_kernel(3); _, ex, _ = eng.rank_side(Xd, side, "worst"); ex = ex.cpu().numpy().copy()
        lib.amdkge_set_rank_kernel(0); _, sc, _ = eng.rank_side(Xd, side, "worst"); st = eng.screen_stats()
        print(model, k, tables, "side", side, "stats", st, "of", n * N, "equal", np.array_equal(sc.cpu().numpy(), ex), flush=True)
