"""The multi-GPU data-path kernels (ampligraph_amd/csrc/kge_shard.hip) through the C ABI, each against a plain numpy /
torch restatement: device-side routing with de-duplication, row gather, gradient scatter-add, the merged data-parallel
sweep, the synthetic triple stream."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from oracle import philox as PH
from test_gpu_kernels import dense, dev, make_engine, make_optimizer, rand_triples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,rank,N,b,nneg,cap", [(2, 0, 83, 64, 0, 200), (2, 1, 83, 64, 192, 400), (8, 3, 14505, 1250, 0, 700),
                                                      (8, 7, 50_000_000, 4096, 4096 * 4, 6000), (4, 2, 1000, 500, 0, 40), (3, 1, 10, 7, 21, 10)])
def test_shard_route_invariants(gpu_lib, world, rank, N, b, nneg, cap):
    from ampligraph_amd.engine import KgeEngine
    from ampligraph_amd.sharded import ShardSpec

    sp = ShardSpec(N, world, rank)
    eng = KgeEngine("DistMult", 4, 8, 3)
    rng = np.random.default_rng(world * 100 + rank)
    X = rand_triples(rng, b, N, 5)
    X[: b // 4, 0] = X[0, 0]                      # repeated ids must share one scratch row
    negs = rand_triples(rng, nneg, N, 5) if nneg else None
    xl, nl, send_ids, counts = eng.shard_route(sp, dev(X), None if negs is None else dev(negs), cap)
    xl, send_ids, counts = xl.cpu().numpy(), send_ids.cpu().numpy(), counts.cpu().numpy()
    ids = np.concatenate([X[:, 0], X[:, 2]] + ([negs[:, 0], negs[:, 2]] if nneg else [])).astype(np.int64)
    loc = np.concatenate([xl[:, 0], xl[:, 2]] + ([nl.cpu().numpy()[:, 0], nl.cpu().numpy()[:, 2]] if nneg else [])).astype(np.int64)
    assert np.array_equal(xl[:, 1], X[:, 1]) and (not nneg or np.array_equal(nl.cpu().numpy()[:, 1], negs[:, 1]))
    local = (ids >= sp.lo) & (ids < sp.hi)
    assert np.array_equal(loc[local], ids[local] - sp.lo)
    rid = np.unique(ids[~local])
    owner = rid // sp.rows_per
    want_counts = np.bincount(owner, minlength=world)
    assert np.array_equal(counts[:world], want_counts)
    overflow = bool((want_counts > cap).any())
    assert bool(counts[world]) == overflow
    if overflow:
        return
    slot = loc[~local] - sp.n_local
    assert slot.min() >= 0 and slot.max() < world * cap
    back = send_ids[slot].astype(np.int64) + (slot // cap) * sp.rows_per          # request list entry -> the global id
    assert np.array_equal(back, ids[~local])
    # equal ids <-> equal slots
    pairs = np.unique(np.stack([ids[~local], slot], 1), axis=0)
    assert len(pairs) == len(rid) == len(np.unique(slot))
    used = np.zeros(world * cap, dtype=bool)
    used[slot] = True
    assert (send_ids[~used] == -1).all()
    for q in range(world):                                                          # lists are filled from the front
        assert used[q * cap:q * cap + want_counts[q]].all() and not used[q * cap + want_counts[q]:(q + 1) * cap].any()
    # a second call with a list that is too short raises the sticky flag, a third (fitting) call keeps it
    if want_counts.max() > 1:
        _, _, _, c2 = eng.shard_route(sp, dev(X), None if negs is None else dev(negs), int(want_counts.max()) - 1)
        assert int(c2[world]) == 1
        _, _, _, c3 = eng.shard_route(sp, dev(X), None if negs is None else dev(negs), cap)
        assert int(c3[world]) == 1 and np.array_equal(c3.cpu().numpy()[:world], want_counts)


def test_gather_and_scatter_add_rows(gpu_lib):
    eng, ent, rel = make_engine("ComplEx", 50, 300, 3, scale=0.5)      # stored rows of 104 floats
    rng = np.random.default_rng(1)
    idx = rng.integers(-1, 300, size=500).astype(np.int32)
    idx[:7] = 11                                                          # repeated destination rows
    got = eng.gather_rows(eng.ent, dev(idx)).clone()
    want = eng.ent[torch.as_tensor(np.maximum(idx, 0).astype(np.int64)).cuda()].clone()
    want[torch.as_tensor(idx < 0).cuda()] = 0
    assert torch.equal(got, want)
    eng.prepare_training("adam")
    src = torch.randn(500, eng.Ks, device="cuda")
    eng.scatter_add_rows(eng.g_ent, dev(idx), src)
    ref = torch.zeros_like(eng.g_ent, dtype=torch.float64)
    keep = torch.as_tensor(idx >= 0).cuda()
    ref.index_add_(0, torch.as_tensor(idx.astype(np.int64)).cuda()[keep], src[keep].double())
    assert float((eng.g_ent.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("name,hp", [("adam", {}), ("adagrad", {}), ("sgd", {"momentum": 0.9}), ("adadelta", {})])
@pytest.mark.parametrize("reg", [None, (3, 1e-2)])
def test_opt_step_merged_equals_sum_then_sweep(gpu_lib, name, hp, reg):
    """amdkge_opt_step_merged over the W received slices == oracle sweep on their sum (tables and slots), for a slice that
    straddles the entity / relation boundary of the flat parameter vector."""
    N, R, k, W = 40, 3, 6, 4
    eng, ent, rel = make_engine("DistMult", k, N, R, scale=0.5)
    w, mk = make_optimizer(name, hp)
    eng.prepare_training(w.name)
    st = mk(ent, rel)
    n = eng.p_flat.numel()
    chunk = n // W
    rng = np.random.default_rng(3)
    lam = reg[1] if reg else 0.0
    for t in range(1, 3):
        parts = (rng.normal(size=(W, n)) * 0.1).astype(np.float32)      # rank q's partial gradient of the WHOLE vector
        total = parts.astype(np.float64).sum(0)
        Ge = total[:eng._ne].reshape(N, eng.Ks)[:, :k]
        Gr = total[eng._off:eng._off + eng._nr].reshape(R, eng.Ks)[:, :k]
        d = w.to_ffi(t, reg[0] if reg else 2)
        eng.loss_acc.zero_()
        for r in range(W):   # every "rank" sweeps its slice of the one engine: together the whole vector
            recv = np.ascontiguousarray(parts[:, r * chunk:(r + 1) * chunk]).reshape(-1)   # what all_to_all delivers to rank r
            eng.opt_step_merged(d, r * chunk, (r + 1) * chunk, dev(recv), W, chunk, lam, lam)
        want_reg = 0.0
        Ge64, Gr64 = Ge.copy(), Gr.copy()
        if reg:
            for x, G in ((st.ent, Ge64), (st.rel, Gr64)):
                xx = x.astype(np.float64)
                want_reg += lam * float((np.abs(xx) ** reg[0]).sum())
                G += lam * reg[0] * np.abs(xx) ** (reg[0] - 1) * np.sign(xx)
        O.apply_optimizer(st, Ge64, Gr64)
        torch.cuda.synchronize()
        e, r_ = eng.get_tables()
        # (the padding units received random "gradient" here: only the live units are compared)
        assert np.abs(e - st.ent).max() <= 3e-6 * max(1.0, np.abs(st.ent).max()) and np.abs(r_ - st.rel).max() <= 3e-6 * max(1.0, np.abs(st.rel).max())
        for nme, ref in st.slots.items():
            assert np.allclose(dense(eng, eng.slots[nme]), ref, rtol=3e-5, atol=3e-7 * max(np.abs(ref).max(), 1e-30)), (nme, t)
        if reg:
            assert abs(float(eng.loss_acc[1]) - want_reg) <= 2e-5 * want_reg


def test_synth_triples_stream(gpu_lib):
    eng, _, _ = make_engine("DistMult", 4, 8, 3)
    N, R, seed = 50_000_000, 1000, 12345678901
    a = eng.synth_triples(seed, 10**9 + 7, 4096, N, R).cpu().numpy()
    b = eng.synth_triples(seed, 10**9 + 7 + 1000, 100, N, R).cpu().numpy()
    assert np.array_equal(a[1000:1100], b)                                # triple i depends on i only
    rows = np.arange(10**9 + 7, 10**9 + 7 + 64, dtype=np.uint64)
    x0, x1, x2, _ = PH.philox4x32_10((rows & 0xFFFFFFFF).astype(np.uint32), (rows >> 32).astype(np.uint32), np.uint32(0x53594e54), np.uint32(0),
                                     seed & 0xFFFFFFFF, seed >> 32)
    want = np.stack([(x0.astype(np.uint64) * N) >> 32, (x1.astype(np.uint64) * R) >> 32, (x2.astype(np.uint64) * N) >> 32], 1)
    assert np.array_equal(a[:64], want.astype(np.int32))
    assert a[:, 0].max() < N and a[:, 1].max() < R and len(np.unique(a[:, 0])) > 4000
