"""Two ranks in ONE process (tests/threaded_dist.py: threads + an in-process rendezvous standing in for RCCL) driving the
product's two multi-GPU step loops with two real KgeEngines on the one dev GPU.  Run under `rocprofv3 --kernel-trace --stats`:
the kernel list shows what a training step launches -- libamdkge kernels only (kge::*), plus the device copies with which the
stand-in emulates the collectives (at::native copy / cat kernels: those are RCCL transfers on a real node)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from threaded_dist import ThreadedWorld  # noqa: E402

from ampligraph_amd.engine import KgeEngine  # noqa: E402
from ampligraph_amd.latent_features import loss_functions, optimizers  # noqa: E402
from ampligraph_amd.sharded import ShardedStepLoop, ShardSpec  # noqa: E402
from ampligraph_amd.trainer import StepLoop  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "rows"      # "rows" (row-sharded entity table) | "dp" (replicated tables)
N, R, k, eta, B, W, STEPS = 123182, 37, 200, 20, 8192, 2, 6   # BASELINE C4 shape
rng = np.random.default_rng(0)
X = np.stack([rng.integers(0, N, STEPS * B * W), rng.integers(0, R, STEPS * B * W), rng.integers(0, N, STEPS * B * W)], 1).astype(np.int32)
lim = float(np.sqrt(6.0 / (N + 2 * k)))


def body(dist):
    r = dist.get_rank()
    g = torch.Generator(device="cuda").manual_seed(1)
    if MODE == "rows":
        sp = ShardSpec(N, W, r)
        cap = ShardedStepLoop.rows_needed(B, eta, "local", W, N)
        eng = KgeEngine("ComplEx", k, sp.n_local + cap, R, max_rel_size=R)
        loop = ShardedStepLoop(eng, sp, eta, loss_functions.get("self_adversarial"), optimizers.get("adam"), None, 0, dist, negatives="local")
    else:
        eng = KgeEngine("ComplEx", k, N, R, max_rel_size=R)
        loop = StepLoop(eng, eta, loss_functions.get("self_adversarial"), optimizers.get("adam"), None, 0, dist, merge="sharded")
    eng.pack((torch.rand(eng.n_ents, eng.K, device="cuda", generator=g) * 2 - 1) * lim, out=eng.ent)
    eng.pack((torch.rand(R, eng.K, device="cuda", generator=g) * 2 - 1) * 0.1, out=eng.rel)
    Xd = torch.as_tensor(X).cuda()
    loop.reset_loss()
    for s in range(STEPS):
        loop.step(Xd[s * B * W:(s + 1) * B * W], s)
    torch.cuda.synchronize()
    return loop.mean_batch_loss()


print(MODE, ThreadedWorld(W).run(body))
