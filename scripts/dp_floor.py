"""Development probe: the data-parallel step of trainer.StepLoop with the collectives replaced by device-local copies of
the same size (an infinitely fast fabric), i.e. what one rank of W spends per step OUTSIDE xGMI.  Single GPU."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ampligraph_amd.datasets import make_synthetic_kg  # noqa: E402
from ampligraph_amd.engine import KgeEngine  # noqa: E402
from ampligraph_amd.latent_features import loss_functions, optimizers  # noqa: E402
from ampligraph_amd.trainer import StepLoop  # noqa: E402


class FakeDist:
    class ReduceOp:
        MAX = "max"

    def __init__(self, world):
        self.world = world

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return 0

    def get_backend(self):
        return "nccl"

    def barrier(self):
        pass

    def all_reduce(self, t, op=None):
        pass

    def all_to_all_single(self, out, inp):
        out.copy_(inp)

    def all_gather_into_tensor(self, out, inp):
        out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))

    def reduce_scatter_tensor(self, out, inp):
        out.copy_(inp.view(self.world, -1)[0])


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    data = make_synthetic_kg("synth-fb15k237", seed=0)
    N, R = data["n_ents"], data["n_rels"]
    rng = np.random.default_rng(0)
    B = 10000
    train = torch.as_tensor(data["train"]).cuda()
    for merge, coll in (("single", None), ("allreduce", None), ("sharded", "alltoall"), ("sharded", "alltoall+allgather"), ("sharded", "native")):
        eng = KgeEngine("ComplEx", 200, N, R, max_rel_size=R)
        eng.set_tables(rng.uniform(-0.02, 0.02, (N, 400)).astype(np.float32), rng.uniform(-0.1, 0.1, (R, 400)).astype(np.float32))
        d = None if merge == "single" else FakeDist(W)
        loop = StepLoop(eng, 20, loss_functions.get("self_adversarial"), optimizers.get("adam"), None, 0, d,
                        merge=None if merge == "single" else merge)
        if coll:
            loop.collectives = coll
        Bg = B * (1 if d is None else W)
        steps = train.shape[0] // Bg

        def batch(s):
            b0 = (s % steps) * Bg
            return train[b0:b0 + Bg]

        for s in range(20):
            loop.step(batch(s), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for s in range(20, 20 + n):
            loop.step(batch(s), s)
        torch.cuda.synchronize()
        print(f"W={W} {merge}/{coll}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)


main()
