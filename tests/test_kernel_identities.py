"""The algebraic identities the round-2 train kernels rest on, checked in fp64 against the oracle's own gradients
(oracle.score_grads restates TransE.py:51-53 / RotatE.py:96-104 and Appendix A of SURVEY.md).  CPU only: these pin the
derivations, the GPU parity tests pin the kernels.

* RotatE, forward kernel (kge_train_kernel.h, single pass): with z_j = s o r - o of corruption j and Z_side = sum_j g_j z_j / |z_j|
  over the corruptions of one side, the row gradients of the positive are LINEAR in Z:
  d/ds = -conj(r) o Z_obj, d/do = +Z_subj, d/dphase = -Im(conj(A) Z_obj) - Im(conj(o) Z_subj) with A = s o r.
* RotatE, tile pass (kge_train_tiled.hip): d|e o r - o| / de = (e - B) / |e - B| with B = o o conj(r), and
  d|A - e| / de = (e - A) / |A - e|.
* TransE: d(-|d|)/d(s, p, o) = (-sign d, -sign d, +sign d) with sign(0) = 0, i.e. a coefficient with the sign bit of d flipped in
  wherever d != 0."""
import numpy as np

from oracle import kge_oracle as O


def _cplx(x):
    h = x.shape[-1] // 2
    return x[..., :h] + 1j * x[..., h:]


def test_rotate_row_gradients_are_linear_in_the_unit_vector_sums():
    rng = np.random.default_rng(0)
    k, R, eta = 24, 5, 9
    K = 2 * k
    s, o = rng.normal(size=(1, K)).astype(np.float32), rng.normal(size=(1, K)).astype(np.float32)
    p = rng.uniform(-0.05, 0.05, size=(1, K)).astype(np.float32)
    E = rng.normal(size=(eta, K)).astype(np.float32)
    g = rng.normal(size=eta)                      # dL/dscore of each corruption
    keep = rng.integers(0, 2, eta).astype(bool)   # True: object replaced (s, p, e_j); False: subject replaced (e_j, p, o)
    div = float(O.rotate_phase_divisor(k, R))
    phi = (p[0, :k] / np.float32(div)).astype(np.float64)
    r = np.cos(phi) + 1j * np.sin(phi)
    S, Ocx, A = _cplx(s[0].astype(np.float64)), _cplx(o[0].astype(np.float64)), None
    A = S * r
    # oracle: accumulate g_j * d score_j / d(s, p, o) over the corruptions
    gs_ref, gp_ref, go_ref = np.zeros(K), np.zeros(K), np.zeros(K)
    Z_obj, Z_subj = np.zeros(k, complex), np.zeros(k, complex)
    for j in range(eta):
        if keep[j]:
            ds, dp, _ = O.score_grads("RotatE", s, p, E[j:j + 1], max_rel_size=R)
            gs_ref += g[j] * ds[0]; gp_ref += g[j] * dp[0]
            z = A - _cplx(E[j].astype(np.float64))
            Z_obj += g[j] * z / np.abs(z)
        else:
            _, dp, do = O.score_grads("RotatE", E[j:j + 1], p, o, max_rel_size=R)
            go_ref += g[j] * do[0]; gp_ref += g[j] * dp[0]
            z = _cplx(E[j].astype(np.float64)) * r - Ocx
            Z_subj += g[j] * z / np.abs(z)
    # score = -sum |z|
    gs = -np.conj(r) * Z_obj
    go = Z_subj
    gphase = -(np.conj(A) * Z_obj).imag - (np.conj(Ocx) * Z_subj).imag
    assert np.allclose(np.concatenate([gs.real, gs.imag]), gs_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(np.concatenate([go.real, go.imag]), go_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(gphase / div, gp_ref[:k], rtol=1e-10, atol=1e-12) and not gp_ref[k:].any()


def test_rotate_replaced_row_gradient_from_the_rotated_side_rows():
    rng = np.random.default_rng(1)
    k, R, n = 16, 3, 40
    K = 2 * k
    s, o, e = (rng.normal(size=(n, K)).astype(np.float32) for _ in range(3))
    p = rng.uniform(-0.1, 0.1, size=(n, K)).astype(np.float32)
    div = float(O.rotate_phase_divisor(k, R))
    r = np.exp(1j * (p[:, :k] / np.float32(div)).astype(np.float64))
    S, Ocx, Ecx = _cplx(s.astype(np.float64)), _cplx(o.astype(np.float64)), _cplx(e.astype(np.float64))
    # object replaced by e: d score/d e
    _, _, do = O.score_grads("RotatE", s, p, e, max_rel_size=R)
    A = S * r
    want = -(Ecx - A) / np.abs(Ecx - A)          # score = -|A - e|
    assert np.allclose(np.concatenate([want.real, want.imag], -1), do, rtol=1e-9, atol=1e-12)
    # subject replaced by e: d score/d e
    ds, _, _ = O.score_grads("RotatE", e, p, o, max_rel_size=R)
    B = Ocx * np.conj(r)
    want = -(Ecx - B) / np.abs(Ecx - B)
    assert np.allclose(np.concatenate([want.real, want.imag], -1), ds, rtol=1e-9, atol=1e-12)


def test_transe_gradient_is_a_sign_flipped_coefficient_with_exact_zeros():
    rng = np.random.default_rng(2)
    n, k = 50, 12
    s, p, o = (rng.integers(-2, 3, size=(n, k)).astype(np.float32) * 0.25 for _ in range(3))   # coarse grid: many exact zeros
    d = (s + p) - o
    assert (d == 0).mean() > 0.05
    c = np.float32(0.37)
    bits = c.view(np.uint32) ^ (d.view(np.uint32) & np.uint32(0x80000000))   # the coefficient with d's sign bit xor-ed in
    flipped = bits.view(np.float32)
    gs, gp, go = O.score_grads("TransE", s, p, o)
    want = np.where(d != 0, flipped, 0.0)        # sign(0) = 0: the kernels take their exact form wherever a unit is zero
    assert np.array_equal(want.astype(np.float64), c * -gs) and np.array_equal(gs, gp) and np.array_equal(go, -gs)
    # the byte handed to the tile pass (top byte of d: sign + 7 exponent bits) carries the sign, and is 0x00 / 0x80 exactly
    # for |d| < 2^-125 -- which on real inputs means d == 0
    top = (d.view(np.uint32) >> 24).astype(np.uint8)
    assert np.array_equal((top & 0x80) != 0, np.signbit(d))
    assert np.array_equal((top & 0x7F) == 0, np.abs(d) < 2.0 ** -125)
