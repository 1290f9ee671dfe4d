"""Philox4x32-10 counter-based RNG, numpy restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The HIP sampler in
ampligraph_amd/csrc/kge_common.h implements the same function; the reference's
own sampler draws from TensorFlow's Philox *stream*
(/root/reference/ampligraph/latent_features/layers/corruption_generation/
CorruptionGenerationLayerTrain.py:55-74, `tf.random.uniform(..., dtype=int32)`),
which cannot be reproduced without TF, so both sides of this repo use the
published Philox4x32-10 block function (Salmon et al., SC'11) keyed directly
by (seed, step, row).  Known-answer vectors from the Random123 distribution
pin the block function in tests/test_oracle_philox.py.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All inputs broadcastable uint32 arrays.

    Returns four uint32 arrays (x0, x1, x2, x3).
    """
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * PHILOX_M0
            p1 = c2.astype(np.uint64) * PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = (hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0)
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def sample_corruption_draws(row_index, step, seed, n_ents):
    """Draw (keep_subj, replacement) for global corruption-row indices.

    Contract shared with the HIP kernel:
      counter = (row_lo, row_hi, step_lo, step_hi), key = (seed_lo, seed_hi)
      keep_subj   = x0 & 1
      replacement = (x1 * n_ents) >> 32          (multiply-shift range reduction)
    """
    row_index = np.asarray(row_index, dtype=np.uint64)
    step = int(step)
    seed = int(seed)
    x0, x1, _, _ = philox4x32_10(
        (row_index & _MASK32).astype(np.uint32),
        (row_index >> np.uint64(32)).astype(np.uint32),
        np.uint32(step & 0xFFFFFFFF),
        np.uint32((step >> 32) & 0xFFFFFFFF),
        seed & 0xFFFFFFFF,
        (seed >> 32) & 0xFFFFFFFF,
    )
    keep_subj = (x0 & np.uint32(1)).astype(np.int32)
    repl = ((x1.astype(np.uint64) * np.uint64(n_ents)) >> np.uint64(32)).astype(np.int32)
    return keep_subj, repl
