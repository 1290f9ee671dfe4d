"""AMDKGE_TILED_DETERMINISTIC: two runs of the same steps produce bitwise equal tables and optimizer state (sorted tile
accumulation, staged relation-row gradient), and the mode computes the same step as the default one / the oracle."""
import numpy as np
import pytest
import torch

from margins import rel_gap, within
from oracle import kge_oracle as O
from test_gpu_kernels import assert_grads_close, dense, dev, loss_desc, make_engine, make_optimizer, rand_triples, run_tiled_grads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,k,N,B,eta", [("ComplEx", 200, 300, 3000, 20),   # ~220 entries per row: order matters a lot
                                               ("TransE", 50, 200, 2000, 5), ("RotatE", 33, 150, 1000, 8), ("DistMult", 350, 500, 2048, 10),
                                               ("HolE", 16, 40, 4000, 3), ("RotatE", 1000, 60, 128, 16)])
def test_two_runs_are_bitwise_equal(gpu_lib, model, k, N, B, eta):
    R = 5
    rng = np.random.default_rng(0)
    X = [rand_triples(rng, B, N, R) for _ in range(3)]
    runs = []
    for rep in range(3):
        eng, ent, rel = make_engine(model, k, N, R, scale=0.3 if k < 100 else 0.08)
        w, _ = make_optimizer("adam", {})
        eng.prepare_training(w.name)
        for t in range(1, 4):
            eng.train_step_tiled(dev(X[t - 1]), eta, loss_desc("self_adversarial"), w.to_ffi(t, 2), 7, t, reg_e=1e-3, reg_r=1e-3,
                                 deterministic=(rep < 2))
        torch.cuda.synchronize()
        assert eng.tiled_status() == 0
        runs.append((eng.ent.clone(), eng.rel.clone(), {n_: s_.clone() for n_, s_ in eng.slots.items()}))
    a, b, c = runs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n_ in a[2]:
        assert torch.equal(a[2][n_], b[2][n_]), n_
    # the default (arrival-order) run computes the same step up to fp32 summation order
    assert float((a[0] - c[0]).abs().max()) < 2.5e-2 and float(((a[0] - c[0]).abs() < 1e-5).float().mean()) > 0.98
    assert bool(torch.isfinite(a[0]).all())


@pytest.mark.parametrize("model,k", [("ComplEx", 32), ("TransE", 50), ("RotatE", 20), ("DistMult", 7), ("HolE", 12)])
def test_deterministic_gradients_match_oracle(gpu_lib, model, k):
    """The mode's gradient-only form (sorted accumulation + rel_backward_det_kernel) against the oracle's dense gradients."""
    from ampligraph_amd import _ffi

    N, R, B, eta = 300, 5, 257, 6
    eng, ent, rel = make_engine(model, k, N, R, scale=0.6)
    rng = np.random.default_rng(3)
    X = rand_triples(rng, B, N, R)
    eng.prepare_training("adam")
    for loss in ("self_adversarial", "pairwise"):
        eng.loss_acc.zero_()
        eng.g_flat.zero_()
        eng.g_ent.fill_(123.0)
        eng.g_rel.zero_()
        d = _ffi.Opt(_ffi.OPTIMIZERS["adam"], 2, 1e-2, 0.9, 0.999, 1e-7, 0.0, 1)
        eng.train_step_tiled(dev(X), eta, loss_desc(loss), d, 9, 4, grad_only=True, deterministic=True)
        torch.cuda.synchronize()
        negs = O.generate_corruptions(X, N, eta, 9, 4)
        total, Te, Tr, (sp, sn, per) = O.dense_gradients(model, ent, rel, X, negs, eta, loss, None, "sum", R)
        assert abs(float(eng.loss_acc[0]) - float(per.astype(np.float64).sum())) <= 1e-5 * max(1.0, abs(float(total)))
        assert_grads_close(dense(eng, eng.g_ent), Te)
        assert_grads_close(dense(eng, eng.g_rel), Tr)


def test_deterministic_fit_is_reproducible(gpu_lib):
    from test_gpu_model import toy_graph

    from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel

    X = toy_graph(5, n=3000, N=40, R=3)
    outs = []
    for _ in range(2):
        m = ScoringBasedEmbeddingModel(eta=5, k=10, scoring_type="ComplEx", seed=1)
        m.compile(optimizer="adam", loss="multiclass_nll", entity_relation_regularizer="l2", deterministic=True)
        h = m.fit(X, batch_size=1000, epochs=3, verbose=False)
        outs.append((m._engine.ent.clone(), m._engine.rel.clone(), h.history["loss"]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert within("det/two_fits/loss", rel_gap(outs[0][2], outs[1][2]), 1e-12)


WIDE = [("DistMult", 400, 30, "self_adversarial"),     # C3's row: 100 quads, two per lane, groups of six rows
        ("ComplEx", 352, 20, "self_adversarial"),      # the reference's published k = 350 as stored: 88 quads per component, groups of two
        ("ComplEx", 600, 8, "nll"),                    # four waves per positive, one quad per component and thread
        ("DistMult", 1200, 6, "multiclass_nll"),       # four waves, two quads per thread
        ("RotatE", 1000, 64, "self_adversarial"),      # C5's row and eta: four waves, groups of three
        ("RotatE", 352, 20, "nll"),                    # one wave, two quads per lane, groups of two
        ("TransE", 600, 8, "pairwise"),                # four waves per positive, the sign-stash form (integer gradients)
        ("TransE", 352, 20, "absolute_margin")]        # one wave, two quads per lane


@pytest.mark.gpu
@pytest.mark.parametrize("model,k,eta,loss", WIDE, ids=[f"{m}-{k}-{e}-{l}" for m, k, e, l in WIDE])
def test_deterministic_steps_bitwise_in_every_launch_geometry(gpu_lib, model, k, eta, loss):
    """Two whole deterministic steps (owner-computes pair + dense Adam) at the row widths that take the forward kernel's other
    launch geometries -- two quads per lane, four waves per positive (partial sums meeting in LDS), groups of two / three / six
    rows -- against oracle/train_ordered (trilinear_step_det / rotate_step_det: _reduce_quads, _pf_of): tables and Adam slots
    bit-identical.  Incl. BASELINE configs[2]'s row (DistMult k = 400, eta = 30) and configs[4]'s (RotatE k = 1000, eta = 64)."""
    import torch

    from oracle import train_ordered as TO

    from ampligraph_amd.engine import KgeEngine
    from ampligraph_amd.latent_features import loss_functions, optimizers
    from ampligraph_amd.trainer import StepLoop

    N, R, B = 1500, 7, 1024
    rng = np.random.default_rng(k + eta)
    K = k if model in ("DistMult", "TransE") else 2 * k
    ent = rng.uniform(-0.2, 0.2, size=(N, K)).astype(np.float32)
    rel = rng.uniform(-0.2, 0.2, size=(R, K)).astype(np.float32)
    X = np.stack([rng.integers(0, N, 2 * B), rng.integers(0, R, 2 * B), rng.integers(0, N, 2 * B)], 1).astype(np.int32)
    eng = KgeEngine(model, k, N, R, max_rel_size=R)
    eng.set_tables(ent, rel)
    loop = StepLoop(eng, eta, loss_functions.get(loss), optimizers.get("adam", {"learning_rate": 1e-2}), None, seed=6, dist=None)
    loop.deterministic = True
    st = TO.OptState(ent.copy(), rel.copy(), "adam", 1e-2)
    Xd = torch.as_tensor(X).cuda()
    loop.reset_loss()
    ref = 0.0
    for step in range(2):
        loop.step(Xd[step * B:(step + 1) * B], step)
        xb = X[step * B:(step + 1) * B]
        ref += (TO.rotate_step_det(st, xb, eta, 6, step, loss, max_rel_size=R) if model == "RotatE"
                else TO.transe_pairwise_step(st, xb, eta, 6, step, loss=loss, layout="quad") if model == "TransE"
                else TO.trilinear_step_det(model, st, xb, eta, 6, step, loss))
    torch.cuda.synchronize()
    got = loop.mean_batch_loss() * 2
    e, r = eng.get_tables()
    assert np.array_equal(e, st.ent) and np.array_equal(r, st.rel), (int((e != st.ent).sum()), int((r != st.rel).sum()), float(np.abs(e - st.ent).max()))
    assert np.array_equal(eng.unpack(eng.slots["m_e"]).cpu().numpy(), st.s0[0]) and np.array_equal(eng.unpack(eng.slots["v_e"]).cpu().numpy(), st.s1[0])
    assert within(f"det/geometry_vs_ordered_oracle/{loss}", rel_gap(got, ref), 5e-6 if loss == "multiclass_nll" else 1e-12), (got, ref)


def _det_plan_mirror(N, K, B, eta):
    """make_plan's deterministic-mode sizing (kge_train_tiled.hip) restated: -> (tile_rows, n_tiles, cap, sort_cap)."""
    entries, row_bytes, budget = B * (eta + 2), K * 4, 96 * 1024
    while True:
        fit = budget // row_bytes
        best, sel, rb = -1.0, None, 8
        while rb >= 1:
            if rb <= fit:
                blocks, mm = (N + rb - 1) // rb, 1
                while True:
                    per = (blocks + 256 * mm - 1) // (256 * mm)
                    if per * rb <= fit and per * rb <= 4096:
                        break
                    mm += 1
                nt = (blocks + per - 1) // per
                eff = N / (per * rb) / (((nt + 255) // 256) * 256)
                if eff > best + 0.03 or best < 0:
                    best, sel = eff, (per * rb, nt)
                if eff >= 0.97:
                    break
            rb >>= 1
        tile_rows, n_tiles = sel
        mean = (entries + n_tiles - 1) // n_tiles
        cap = 2 * mean + (256 if mean >= 224 else 32 + mean)
        sc = (cap + 64 + 63) & ~63
        fixed = tile_rows * K * 4 + 4096 + 16 + 1024
        if sc <= 8192 and fixed + sc * 20 <= 158 * 1024:
            while sc * 2 <= 8192 and fixed + sc * 2 * 20 <= 158 * 1024:
                sc *= 2
            return tile_rows, n_tiles, cap, sc
        budget = budget * 3 // 4


def test_deterministic_hub_row_with_a_sort_queue_stride_that_is_no_power_of_two(gpu_lib):
    """the wave-local index sort of the deterministic tile pass pads a wave's queue to a power of two; the queues
    are sort_cap / 8 indices apart and sort_cap is only a multiple of 64.  On the FB15K-237 shape with a short batch (ComplEx k = 200,
    B = 1 250, eta = 20: sort_cap 1 792, stride 224) a hub row with 129 .. 224 entries made its wave pad to 256 -- into the next wave's
    queue.  A wave's capacity is now the largest power of two inside the stride and such a tile takes the block-wide fall-back.
    Two steps against oracle/train_ordered, bit for bit, and three runs equal."""
    from oracle import train_ordered as TO

    from ampligraph_amd.engine import KgeEngine
    from ampligraph_amd.latent_features import loss_functions, optimizers
    from ampligraph_amd.trainer import StepLoop

    model, k, N, R, B, eta = "ComplEx", 200, 14505, 7, 1250, 20
    tile_rows, n_tiles, cap, sort_cap = _det_plan_mirror(N, 2 * k, B, eta)
    stride = sort_cap // 8
    assert stride < 256 and stride & (stride - 1), (sort_cap, "the shape no longer has the stride this test is about")
    rng = np.random.default_rng(11)
    ent = rng.uniform(-0.2, 0.2, size=(N, 2 * k)).astype(np.float32)
    rel = rng.uniform(-0.2, 0.2, size=(R, 2 * k)).astype(np.float32)
    X = np.stack([rng.integers(0, N, 2 * B), rng.integers(0, R, 2 * B), rng.integers(0, N, 2 * B)], 1).astype(np.int32)
    for step in range(2):   # 190 positives of each batch point at one hub: its owner wave holds > 128 and <= stride entries
        X[step * B:step * B + 190, 2] = 4321
    outs = []
    for rep in range(3):
        eng = KgeEngine(model, k, N, R, max_rel_size=R)
        eng.set_tables(ent, rel)
        loop = StepLoop(eng, eta, loss_functions.get("self_adversarial"), optimizers.get("adam", {"learning_rate": 1e-2}), None, seed=6, dist=None)
        loop.deterministic = True
        Xd = torch.as_tensor(X).cuda()
        loop.reset_loss()
        for step in range(2):
            loop.step(Xd[step * B:(step + 1) * B], step)
        torch.cuda.synchronize()
        assert eng.tiled_status() == 0
        outs.append(eng.get_tables())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[1][1])
    st = TO.OptState(ent.copy(), rel.copy(), "adam", 1e-2)
    for step in range(2):
        TO.trilinear_step_det(model, st, X[step * B:(step + 1) * B], eta, 6, step, "self_adversarial")
    e, r = outs[0]
    assert np.array_equal(e, st.ent) and np.array_equal(r, st.rel), (int((e != st.ent).sum()), float(np.abs(e - st.ent).max()))
