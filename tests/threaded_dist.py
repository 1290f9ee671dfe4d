"""TEST INFRASTRUCTURE: an in-process stand-in for torch.distributed that runs W "ranks" as threads of one
process.  Lets the row-sharded plumbing (ampligraph_amd/sharded.py) drive W real KgeEngines on ONE GPU: the box
the -m gpu tests run on has a single MI355X, RCCL refuses two ranks on one device, and gloo does not move
device tensors.  Collectives are rendezvous through a barrier + shared slots; semantics follow
torch.distributed (all_to_all_single with split sizes, all_reduce sum, all_gather, barrier)."""
import threading

import torch


class ThreadedWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.errors = []

    def run(self, fn):
        """fn(dist) is called on `world` threads; returns the list of results by rank."""
        out = [None] * self.world

        def body(r):
            try:
                out[r] = fn(ThreadedDist(self, r))
            except BaseException as e:  # noqa: BLE001  (re-raised on the main thread)
                self.errors.append(e)
                self.barrier.abort()

        ths = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if self.errors:
            raise self.errors[0]
        return out


class ThreadedDist:
    def __init__(self, w, rank):
        self.w, self.rank = w, rank

    def get_world_size(self):
        return self.w.world

    def get_rank(self):
        return self.rank

    def barrier(self):
        self.w.barrier.wait()

    def _exchange(self, obj):
        if obj is not None and isinstance(obj, torch.Tensor) and obj.is_cuda:
            torch.cuda.synchronize()
        self.w.slots[self.rank] = obj
        self.w.barrier.wait()
        got = list(self.w.slots)
        self.w.barrier.wait()
        return got

    def all_reduce(self, t, op=None):
        parts = self._exchange(t.clone())
        tot = parts[0].clone()
        for p in parts[1:]:
            tot += p
        t.copy_(tot)
        self.barrier()

    def all_gather(self, out_list, t):
        parts = self._exchange(t.clone())
        for o, p in zip(out_list, parts):
            o.copy_(p)
        self.barrier()

    def all_gather_into_tensor(self, out, t):
        parts = self._exchange(t.clone())
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view_as(out))
        self.barrier()

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        W = self.w.world
        if in_splits is None:
            in_splits = [inp.numel() // W] * W
        if out_splits is None:
            out_splits = [out.numel() // W] * W
        chunks = list(torch.split(inp.clone(), in_splits))
        allc = self._exchange(chunks)
        mine = [allc[src][self.rank] for src in range(W)]
        assert [int(c.numel()) for c in mine] == [int(x) for x in out_splits], (out_splits, [c.numel() for c in mine])
        if out.numel():
            out.copy_(torch.cat(mine))
        self.barrier()
