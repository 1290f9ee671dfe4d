"""development: the undecided-pair list and the screening pass's own counts of ONE rank_side call (C2 shape, screen_time.py's tables), saved
for a diff between libraries (AMDKGE_LIB) -- the list's content, not only its length.  usage: screen_pairs_dump.py <out.npz>"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ampligraph_amd import _ffi
from ampligraph_amd.engine import KgeEngine
N, R, k, n = 14505, 237, 200, 20438
rng = np.random.default_rng(0)
eng = KgeEngine("ComplEx", k, N, R, max_rel_size=R)
eng.set_tables((rng.normal(size=(N, eng.K)) * 0.25).astype(np.float32), (rng.normal(size=(R, eng.K)) * 0.25).astype(np.float32))
X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1).astype(np.int32)
Xd = torch.as_tensor(X).cuda()
out = {}
for rep in range(2):
    ranks = eng.rank_side(Xd, _ffi.SIDE_S, "worst")[0]
    torch.cuda.synchronize()
    ws = eng._last_screen.cpu().numpy()
    base = (-eng._last_screen.data_ptr()) % 256   # carve_screen aligns to 256
    up = lambda x: (x + 255) & ~255
    U = 2 * k; S = (U + 31) // 32; m = N
    p = base
    counter = ws[p:p + 256].view(np.int32); p += 256
    counts = ws[p:p + n * 8].view(np.int32).reshape(n, 2).copy(); p += up(n * 8)
    p += up(((n + 31) // 32 + 4) * S * 3072) + up(n * 16) + up(n * 8) + up(((m + 31) // 32 + 4) * S * 3072) + up(m * 16) + up(((m + 63) // 64 + 4) * 16)
    cnt = int(counter[0])
    pairs = ws[p:p + cnt * 8].view(np.int32).reshape(cnt, 2).copy()
    key = pairs[:, 0].astype(np.int64) * (1 << 20) + pairs[:, 1]
    out["pairs%d" % rep] = np.sort(key); out["counts%d" % rep] = counts; out["ranks%d" % rep] = ranks.cpu().numpy()
    print("rep", rep, "count", cnt, "overflow", int(counter[1]), "unique", len(np.unique(key)), "counts sum", counts.sum(0))
np.savez_compressed(sys.argv[1], **out)
