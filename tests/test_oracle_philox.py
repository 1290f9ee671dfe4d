"""Pins oracle/philox.py against the Random123 known-answer vectors for philox4x32-10
(Random123 `kat_vectors`, Salmon et al. SC'11)."""
import numpy as np

from oracle.philox import philox4x32_10, sample_corruption_draws

KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
     (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_philox_kat():
    for ctr, key, exp in KAT:
        out = philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(x[0]) for x in out) == exp


def test_draw_contract():
    rows = np.arange(100000, dtype=np.uint64)
    keep, repl = sample_corruption_draws(rows, step=3, seed=7, n_ents=14505)
    assert set(np.unique(keep)) == {0, 1}
    assert repl.min() >= 0 and repl.max() < 14505
    assert abs(keep.mean() - 0.5) < 0.01
    # different steps / seeds give different streams, same inputs reproduce
    k2, r2 = sample_corruption_draws(rows, step=4, seed=7, n_ents=14505)
    assert (r2 != repl).mean() > 0.99
    k3, r3 = sample_corruption_draws(rows, step=3, seed=7, n_ents=14505)
    assert (r3 == repl).all() and (k3 == keep).all()
    # uniformity: chi-square-ish bound on 10 buckets
    hist = np.bincount(repl * 10 // 14505, minlength=10)
    assert hist.min() > 9000 and hist.max() < 11000
