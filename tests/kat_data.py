"""Known-answer inputs/outputs transcribed from the reference's scoring tests
(/root/reference/tests/ampligraph/latent_features/layers/scoring/test_{TransE,DistMult,ComplEx,HolE,RotatE}.py)."""
import numpy as np

f32 = np.float32


def _cplx_triples():
    return (np.array([[1, 1, 1, 2, 2, 2], [10, 10, 10, 11, 11, 11]], dtype=f32),
            np.array([[5, 5, 5, 3, 3, 3], [100, 100, 100, 101, 101, 101]], dtype=f32),
            np.array([[4, 4, 4, 6, 6, 6], [9, 9, 9, 19, 19, 19]], dtype=f32))


def _real_triples(model):
    if model == "TransE":  # test_TransE.py:18-24
        return (np.full((2, 7), 1, f32) * np.array([[1], [10]], f32),
                np.full((2, 7), 1, f32) * np.array([[13], [100]], f32),
                np.array([[4, 4, 4, 4, 4, 4, 9], [90] * 7], dtype=f32))
    return (np.full((2, 7), 1, f32) * np.array([[1], [10]], f32),   # test_DistMult.py:18-24
            np.full((2, 7), 1, f32) * np.array([[5], [100]], f32),
            np.full((2, 7), 1, f32) * np.array([[4], [9]], f32))


EXPECTED = {
    "TransE": np.array([-65., -140.], f32),
    "DistMult": np.array([140., 63000.], f32),
    "ComplEx": np.array([222., 117273.], f32),
    "HolE": 2 * np.array([222., 117273.], f32) / 3.0,
    "RotatE": np.array([-28.03, -94.19], f32),
}


