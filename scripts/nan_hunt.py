"""Development aid: hunt the birth of a non-finite value in the tables during fit().

Runs the fits of tests/test_gpu_learning.py::test_mean_mrr_over_seeds_matches_oracle (planted graph, k = 16, eta = 5, Adam 2e-2, 40 epochs
of 5 steps) through the drop-in class with a watch around StepLoop.step: the tables before every step are kept, isfinite of both tables
is checked after it.  At the first step that leaves a non-finite value the script stops that fit and names the cause from the PRE-step
tables on the host: the step's positives and corruptions (the kernels' own Philox draws, oracle.generate_corruptions), every unit's
z = s o r - o in the kernels' fp32 operation order (numpy float32, -ffp-contract=off semantics: one rounding per operation; cos / sin of
the fp32 phase correctly rounded as rel_phase_kernel / prep_rel_exact give them) and in fp64 -- a unit with zr == 0 and zi == 0 in fp32 is
RotatE's modulus-zero case: sqrt(0) = 0, 0 / 0 = NaN in the gradient, exactly what the reference's tf.sqrt gradient yields
(/root/reference/ampligraph/latent_features/layers/scoring/RotatE.py:102-104: no epsilon).

usage: nan_hunt.py MODEL LOSS FIRST_SEED N_FITS OUT.jsonl [mode]
  mode: default | forced (hot-row replicas for every entity that qualifies at threshold 0, AMDKGE_DEBUG_BUCKET_CAP must be set by the
        caller for the overflow list, scratch poisoned with 0xFF between fits) | det (compile(deterministic=True))
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))

import torch  # noqa: E402
from planted import planted_kg  # noqa: E402

from ampligraph_amd import trainer  # noqa: E402
from ampligraph_amd.latent_features import ScoringBasedEmbeddingModel, optimizers  # noqa: E402
from oracle import kge_oracle as O  # noqa: E402

EPOCHS, BATCH, ETA, K, LR = 40, 1024, 5, 16, 2e-2   # tests/test_gpu_learning.py
F32 = np.float32


class Watch:
    """Wraps a StepLoop: keeps the tables as they were before each step and stops checking after the first non-finite step."""

    def __init__(self, loop, eng):
        self.loop, self.eng, self.bad, self.steps = loop, eng, None, 0

    def __getattr__(self, name):
        return getattr(self.loop, name)

    def step(self, batch, rng_step, *a):
        eng = self.eng
        if self.bad is None:
            prev = (eng.ent.clone(), eng.rel.clone())
        self.loop.step(batch, rng_step, *a)
        self.steps += 1
        if self.bad is None and not (bool(torch.isfinite(eng.ent).all()) and bool(torch.isfinite(eng.rel).all())):
            self.bad = dict(rng_step=int(rng_step), batch=batch.cpu().numpy().copy(),
                            ent_before=eng.unpack(prev[0]).cpu().numpy(), rel_before=eng.unpack(prev[1]).cpu().numpy(),
                            ent_after=eng.unpack(eng.ent).cpu().numpy(), rel_after=eng.unpack(eng.rel).cpu().numpy())


def rotate_units(ent, rel, tri, k, n_rels):
    """Per-unit z of RotatE for the id triples `tri`, in the train kernels' fp32 operation order and in fp64 -> (zr32, zi32, |z|64)."""
    div = F32(O.rotate_phase_divisor(k, n_rels))
    s, p, o = ent[tri[:, 0]], rel[tri[:, 1]], ent[tri[:, 2]]
    phi = (p[:, :k] / div).astype(F32)
    c, sn = np.cos(phi.astype(np.float64)).astype(F32), np.sin(phi.astype(np.float64)).astype(F32)
    sr, si, orr, oi = s[:, :k], s[:, k:], o[:, :k], o[:, k:]
    zr = ((sr * c).astype(F32) - (si * sn).astype(F32)).astype(F32) - orr
    zi = ((sr * sn).astype(F32) + (si * c).astype(F32)).astype(F32) - oi
    c6, s6 = np.cos(phi.astype(np.float64)), np.sin(phi.astype(np.float64))
    z6 = np.hypot(sr.astype(np.float64) * c6 - si.astype(np.float64) * s6 - orr, sr.astype(np.float64) * s6 + si.astype(np.float64) * c6 - oi)
    return zr.astype(F32), zi.astype(F32), z6


def transe_units(ent, rel, tri):
    s, p, o = ent[tri[:, 0]], rel[tri[:, 1]], ent[tri[:, 2]]
    return ((s + p).astype(F32) - o).astype(F32)


def explain(model, seed, bad, n_ents, n_rels):
    X = bad["batch"]
    negs = O.generate_corruptions(X, n_ents, ETA, seed, bad["rng_step"])
    ent, rel = bad["ent_before"], bad["rel_before"]
    out = dict(rng_step=bad["rng_step"], epoch=bad["rng_step"] // 5, positives=int(len(X)),
               tables_finite_before=bool(np.isfinite(ent).all() and np.isfinite(rel).all()),
               entity_rows_nonfinite_after=[int(r) for r in np.nonzero(~np.isfinite(bad["ent_after"]).all(1))[0][:16]],
               n_entity_rows_nonfinite_after=int((~np.isfinite(bad["ent_after"]).all(1)).sum()),
               relation_rows_nonfinite_after=[int(r) for r in np.nonzero(~np.isfinite(bad["rel_after"]).all(1))[0]])
    if model == "RotatE":
        k = ent.shape[1] // 2
        events = []
        for name, tri in (("positive", X), ("corruption", negs)):
            zr, zi, z6 = rotate_units(ent, rel, tri, k, n_rels)
            zero = (zr == 0) & (zi == 0)
            out[f"{name}_units"] = int(zero.size)
            out[f"{name}_units_with_fp32_modulus_zero"] = int(zero.sum())
            out[f"{name}_smallest_fp64_modulus"] = float(z6.min())
            for t, u in zip(*np.nonzero(zero)):
                events.append(dict(kind=name, row_in_list=int(t), triple=[int(x) for x in tri[t]], unit=int(u), fp64_modulus=float(z6[t, u]),
                                   typical_fp64_modulus_of_this_triple=float(np.median(z6[t]))))
        out["modulus_zero_events"] = events[:8]
        touched = set()
        for e in events:
            touched.update((e["triple"][0], e["triple"][2]))
        out["every_nonfinite_entity_row_is_an_s_or_o_of_an_event"] = bool(set(out["entity_rows_nonfinite_after"]) <= touched) if events else False
    return out


def main():
    model, loss, first, count, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    mode = sys.argv[6] if len(sys.argv) > 6 else "default"
    if mode == "forced":
        trainer.HOT_ROW_REPLICA_THRESHOLD = 0.0   # every entity that is an s / o at all qualifies (the 64 most frequent get replica rows)
    t0 = time.time()
    found, nan_loss_fits, fits = [], 0, 0
    with open(path, "a") as f:
        for seed in range(first, first + count):
            d = planted_kg(model, seed=seed % 4096)   # (graphs repeat beyond 4 096 seeds; the model's seed -- tables and negatives -- does not)
            train = d["train"].astype(str)
            m = ScoringBasedEmbeddingModel(eta=ETA, k=K, scoring_type=model, seed=seed)
            m.compile(optimizer=optimizers.get("adam", {"learning_rate": LR}), loss=loss, deterministic=(mode == "det"))
            make = m._make_loop
            holder = {}

            def wrapped():
                holder["w"] = Watch(make(), m._engine)
                return holder["w"]

            m._make_loop = wrapped
            if mode == "forced":   # poison what the caching allocator hands out next
                junk = [torch.full((1 << 24,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(4)]
                del junk
            hist = m.fit(train, batch_size=BATCH, epochs=EPOCHS, verbose=False).history["loss"]
            fits += 1
            w = holder["w"]
            nan_loss_fits += int(not np.isfinite(hist).all())
            if w.bad is not None:
                rec = dict(model=model, loss=loss, mode=mode, seed=seed, graph_seed=seed % 4096, steps=w.steps,
                           first_nonfinite_loss_epoch=int(np.nonzero(~np.isfinite(hist))[0][0]) if not np.isfinite(hist).all() else None,
                           last_epoch_loss=float(hist[-1]), **explain(model, seed, w.bad, m._n_ents, m._n_rels))
                found.append(rec)
                f.write(json.dumps(rec) + "\n"); f.flush()
                print("NON-FINITE", json.dumps(rec), flush=True)
            elif not np.isfinite(hist).all():
                f.write(json.dumps(dict(model=model, loss=loss, mode=mode, seed=seed, nonfinite_loss_with_finite_tables=True)) + "\n"); f.flush()
            if fits % 500 == 0:
                print(f"{fits} fits, {len(found)} non-finite, {time.time() - t0:.0f} s", flush=True)
        summary = dict(summary=True, model=model, loss=loss, mode=mode, first_seed=first, fits=fits, fits_with_nonfinite_tables=len(found),
                       fits_with_nonfinite_loss=nan_loss_fits, seconds=round(time.time() - t0, 1),
                       bucket_cap_env=os.environ.get("AMDKGE_DEBUG_BUCKET_CAP"))
        f.write(json.dumps(summary) + "\n")
        print(json.dumps(summary), flush=True)


if __name__ == "__main__":
    main()
