"""development: evaluate()-shaped rank_side calls at the C2 shape (random tables) -- run under rocprofv3 for per-kernel times"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ampligraph_amd import _ffi
from ampligraph_amd.engine import KgeEngine
N, R, k, n = 14505, 237, int(os.environ.get("ST_K", "200")), 20438   # ST_MODEL / ST_K: other row widths at the same shape
rng = np.random.default_rng(0)
eng = KgeEngine(os.environ.get("ST_MODEL", "ComplEx"), k, N, R, max_rel_size=R)
eng.set_tables((rng.normal(size=(N, eng.K)) * 0.25).astype(np.float32), (rng.normal(size=(R, eng.K)) * 0.25).astype(np.float32))
X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1).astype(np.int32)
Xd = torch.as_tensor(X).cuda()
import zlib
ts = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = [eng.rank_side(Xd, side, "worst")[0] for side in (_ffi.SIDE_S, _ffi.SIDE_O)]
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
crc = zlib.crc32(torch.stack(out).cpu().numpy().tobytes())   # (the same ranks whatever the screening kernel: compare across AMDKGE_SCREEN_KERNEL runs)
print("two sides, serial: median %.3f ms  min %.3f ms  (last 6 of 8)  screen kernel %s  recheck %s  ranks crc32 %08x" % (
    float(np.median(ts[2:])) * 1e3, min(ts[2:]) * 1e3, os.environ.get("AMDKGE_SCREEN_KERNEL", "default"), eng.screen_stats(), crc))
