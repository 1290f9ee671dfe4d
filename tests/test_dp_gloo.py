"""N>1 path on CPU: world_size-2 gloo runs of the product's StepLoop (ampligraph_amd/trainer.py) with an
oracle-backed engine.  Checks the sharding contract: 2 ranks at B/2 == 1 rank at B (same corruptions by
construction, same update up to fp32 summation order), losses aggregate, ragged batches work."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(0)
    N, R, k = 40, 3, 6
    ent = (rng.normal(size=(N, 2 * k)) * 0.4).astype(np.float32)
    rel = (rng.normal(size=(R, 2 * k)) * 0.4).astype(np.float32)
    X = np.stack([rng.integers(0, N, 101), rng.integers(0, R, 101), rng.integers(0, N, 101)], 1).astype(np.int32)
    return ent, rel, X, k


def _run(world, rank, port, out, tiled=False, flat=False, opt=("adam", {})):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleEngine

    from ampligraph_amd.latent_features import loss_functions, optimizers, regularizers
    from ampligraph_amd.trainer import StepLoop

    d = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        d = dist
    ent, rel, X, k = _problem()
    eng = OracleEngine("ComplEx", k, ent, rel, tiled=tiled, flat=flat)
    loop = StepLoop(eng, 3, loss_functions.get("self_adversarial"), optimizers.get(opt[0], dict(opt[1], learning_rate=1e-2)),
                    regularizers.get("LP", {"p": 2, "lambda": 1e-3}), seed=5, dist=d)
    Xt = torch.as_tensor(X)
    loop.reset_loss()
    bs = 37   # ragged: batches of 37,37,27 and odd shard sizes
    step = 0
    for ep in range(2):
        for b0 in range(0, X.shape[0], bs):
            loop.step(Xt[b0:b0 + bs], step)
            step += 1
    loss = loop.mean_batch_loss()
    if world > 1:
        assert loop.merge == ("sharded" if flat else "allreduce") and (eng.flat_sweeps > 0) == flat
    if rank == 0:
        np.savez(out, ent=eng.state.ent, rel=eng.state.rel, loss=loss, calls=np.array(eng.calls))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_bounds():
    from ampligraph_amd.trainer import shard_bounds

    for n in (0, 1, 7, 10000, 10001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("tiled,flat", [(False, False), (True, False), (True, True), (False, True)])
def test_two_ranks_equal_one_rank(tmp_path, tiled, flat):
    """tiled=True drives StepLoop through the owner-computes entry points: in-place entity update +
    relation-only sweep on one rank, gradient-only form + merge on two.  flat=True gives the engine KgeEngine's flat
    parameter buffers, which selects the sharded merge (all_to_all reduce-scatter, sharded sweep, all_gather); flat=False
    keeps the all-reduce merge."""
    single = str(tmp_path / "single.npz")
    _run(1, 0, 0, single, tiled)
    if tiled:   # the two single-rank paths agree with each other as well
        plain = str(tmp_path / "plain.npz")
        _run(1, 0, 0, plain, False)
        a, b = np.load(single), np.load(plain)
        assert np.abs(a["ent"] - b["ent"]).max() < 5e-6 and np.abs(a["rel"] - b["rel"]).max() < 5e-6
        assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))
    port = _free_port()
    multi = str(tmp_path / "multi.npz")
    mp.spawn(_run_spawn, args=(2, port, multi, tiled, flat), nprocs=2, join=True)
    a, b = np.load(single), np.load(multi)
    assert np.abs(a["ent"] - b["ent"]).max() < 5e-6
    assert np.abs(a["rel"] - b["rel"]).max() < 5e-6
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))
    # rank 0 of the 2-rank run processed the first half of every global batch with global RNG rows
    calls = b["calls"]
    assert (calls[:, 1] == 0).all() and set(calls[:, 2]) == {37, 27} and set(calls[:, 0]) == {18, 13}


def _run_forced(rank, port, out, flat, tune):
    """A process group of ONE rank with AMDKGE_FORCE_DIST=1: the multi-rank form of the step (gradient-only kernels + merge
    through the collectives) -- what `AMDKGE_BENCH_FORCE_DIST=1 bench.py` runs through RCCL on a one-GPU box."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["AMDKGE_FORCE_DIST"] = "1"
    from oracle_backend import OracleEngine

    from ampligraph_amd.latent_features import loss_functions, optimizers, regularizers
    from ampligraph_amd.trainer import StepLoop

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    ent, rel, X, k = _problem()
    eng = OracleEngine("ComplEx", k, ent, rel, tiled=True, flat=flat)
    loop = StepLoop(eng, 3, loss_functions.get("self_adversarial"), optimizers.get("adam", {"learning_rate": 1e-2}),
                    regularizers.get("LP", {"p": 2, "lambda": 1e-3}), seed=5, dist=dist)
    assert loop.multi and loop.world == 1 and loop.merge == ("sharded" if flat else "allreduce")
    Xt = torch.as_tensor(X)
    bs = 37
    if tune:   # every schedule, measured: each must carry out the same update
        used = loop.tune_merge(lambda st: Xt[(st % 3) * bs:(st % 3) * bs + bs], 0, trials=1)
        assert used == 6 and loop.merge_report is not None and all(v is not None for v in loop.merge_report.values()), loop.merge_report
    else:
        loop.reset_loss()
        step = 0
        for ep in range(2):
            for b0 in range(0, X.shape[0], bs):
                loop.step(Xt[b0:b0 + bs], step)
                step += 1
        assert (eng.flat_sweeps > 0) == flat
        np.savez(out, ent=eng.state.ent, rel=eng.state.rel, loss=loop.mean_batch_loss())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flat", [False, True])
def test_forced_single_rank_group_equals_plain_run(tmp_path, flat):
    single, forced = str(tmp_path / "single.npz"), str(tmp_path / "forced.npz")
    _run(1, 0, 0, single, True)
    mp.spawn(_run_forced, args=(_free_port(), forced, flat, False), nprocs=1, join=True)
    a, b = np.load(single), np.load(forced)
    assert np.abs(a["ent"] - b["ent"]).max() < 5e-6 and np.abs(a["rel"] - b["rel"]).max() < 5e-6
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))
    mp.spawn(_run_forced, args=(_free_port(), forced, True, True), nprocs=1, join=True)   # tune_merge over a group of one


def _run_spawn(rank, world, port, out, tiled=False, flat=False, opt=("adam", {})):
    _run(world, rank, port, out, tiled, flat, opt)


@pytest.mark.parametrize("opt", [("rmsprop", {"momentum": 0.5}), ("sgd", {"momentum": 0.9, "nesterov": True}), ("adadelta", {})])
def test_two_ranks_equal_one_rank_other_optimizers(tmp_path, opt):
    """The sharded-optimizer merge with the other update rules (one / two state tensors, hyper-parameters carried in the
    descriptor's beta fields): 2 ranks == 1 rank."""
    single, multi = str(tmp_path / "single.npz"), str(tmp_path / "multi.npz")
    _run(1, 0, 0, single, True, True, opt)
    mp.spawn(_run_spawn, args=(2, _free_port(), multi, True, True, opt), nprocs=2, join=True)
    a, b = np.load(single), np.load(multi)
    assert np.abs(a["ent"] - b["ent"]).max() < 5e-6 and np.abs(a["rel"] - b["rel"]).max() < 5e-6
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))


def test_three_ranks_fall_back_to_allreduce(tmp_path):
    """The flat buffers split evenly over 1, 2, 4, 8, 16 ranks; any other count keeps the all-reduce merge and must
    still reproduce the single-rank run."""
    single = str(tmp_path / "single.npz")
    _run(1, 0, 0, single, True, True)
    port = _free_port()
    multi = str(tmp_path / "multi.npz")
    mp.spawn(_run_spawn3, args=(3, port, multi), nprocs=3, join=True)
    a, b = np.load(single), np.load(multi)
    assert np.abs(a["ent"] - b["ent"]).max() < 5e-6 and np.abs(a["rel"] - b["rel"]).max() < 5e-6
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))


def _run_spawn3(rank, world, port, out):
    _run3(world, rank, port, out)


def _run3(world, rank, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleEngine

    from ampligraph_amd.latent_features import loss_functions, optimizers, regularizers
    from ampligraph_amd.trainer import StepLoop

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ent, rel, X, k = _problem()
    eng = OracleEngine("ComplEx", k, ent, rel, tiled=True, flat=True)
    loop = StepLoop(eng, 3, loss_functions.get("self_adversarial"), optimizers.get("adam", {"learning_rate": 1e-2}),
                    regularizers.get("LP", {"p": 2, "lambda": 1e-3}), seed=5, dist=dist)
    assert loop.merge == "allreduce"   # 1024-float padding does not split over 3 ranks
    Xt = torch.as_tensor(X)
    loop.reset_loss()
    step = 0
    for ep in range(2):
        for b0 in range(0, X.shape[0], 37):
            loop.step(Xt[b0:b0 + 37], step)
            step += 1
    loss = loop.mean_batch_loss()
    if rank == 0:
        np.savez(out, ent=eng.state.ent, rel=eng.state.rel, loss=loss)
    dist.barrier()
    dist.destroy_process_group()


def _run_tuned(rank, world, port, out, force=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleEngine

    from ampligraph_amd.latent_features import loss_functions, optimizers, regularizers
    from ampligraph_amd.trainer import StepLoop

    d = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        d = dist
        if force == "reject":   # a backend that rejects one of the collectives: that candidate is dropped, training goes on
            class Rejecting:
                def __getattr__(self, name):
                    return getattr(dist, name)

                def all_gather_into_tensor(self, *a, **kw):
                    raise RuntimeError("all_gather_into_tensor: not supported by this backend build")
            d, force = Rejecting(), None
    ent, rel, X, k = _problem()
    eng = OracleEngine("ComplEx", k, ent, rel, tiled=True, flat=True)
    loop = StepLoop(eng, 3, loss_functions.get("self_adversarial"), optimizers.get("adam", {"learning_rate": 1e-2}),
                    regularizers.get("LP", {"p": 2, "lambda": 1e-3}), seed=5, dist=d)
    Xt = torch.as_tensor(X)
    bs, nb = 37, 3

    def batch_of(step):
        b0 = (step % nb) * bs
        return Xt[b0:b0 + bs]

    pick = (lambda cands, secs: [c[0] for c in cands].index(force)) if force else None
    used = loop.tune_merge(batch_of, 0, trials=2, pick=pick) if world > 1 else 0
    if world > 1 and not isinstance(d, type(dist)):
        assert used == 6 and loop.merge_report["sharded/alltoall+allgather"] is None and loop.collectives != "alltoall+allgather"
    elif world > 1:
        assert used == 9 and set(loop.merge_report) == {"allreduce", "sharded/alltoall", "sharded/alltoall+allgather"}
        assert loop.merge in ("sharded", "allreduce") and (force is None or loop.merge == force)
    for step in range(used, 14):
        loop.step(batch_of(step), step)
    # whichever schedule won, a checkpoint needs complete optimizer slots on every rank
    loop.sync_optimizer_slots()
    np.savez(out + f".{rank}.npz", ent=eng.state.ent, rel=eng.state.rel, m=eng._slot_flat["m"], v=eng._slot_flat["v"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("force", [None, "allreduce", "reject"])
def test_tune_merge_switches_schedules_mid_training(tmp_path, force):
    """StepLoop.tune_merge: the candidates (all-reduce, all_to_all/all_to_all, all_to_all/all_gather) each take a few REAL
    training steps; whatever is picked, tables and optimizer slots after 14 steps equal the single-rank run's.  With
    force="allreduce" the choice is pinned to the all-reduce: exercises the switch back from sharded slots."""
    single = str(tmp_path / "single")
    _run_tuned(0, 1, 0, single)
    port = _free_port()
    multi = str(tmp_path / "multi")
    mp.spawn(_run_tuned, args=(2, port, multi, force), nprocs=2, join=True)
    a = np.load(single + ".0.npz")
    for r in range(2):
        b = np.load(multi + f".{r}.npz")
        for key in ("ent", "rel", "m", "v"):
            assert np.abs(a[key] - b[key]).max() < 5e-6, (r, key)
