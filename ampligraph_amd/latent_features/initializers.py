"""Embedding initialisers.  The reference hands the name to Keras
(/root/reference/ampligraph/latent_features/layers/encoding/EmbeddingLookupLayer.py:105-129, default
"glorot_uniform"); formulas below are Keras' (fan_in = rows, fan_out = cols for a 2-D table), the random
stream is numpy PCG64(seed) because TF's cannot be reproduced without TF."""
import math

import numpy as np


def _fans(shape):
    return shape[0], shape[1]


def _make(name, shape, rng):
    fi, fo = _fans(shape)
    if name == "glorot_uniform":
        lim = math.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape)
    if name == "glorot_normal":
        return _trunc_normal(rng, shape, math.sqrt(2.0 / (fi + fo)) / 0.87962566103423978)
    if name == "he_uniform":
        lim = math.sqrt(6.0 / fi)
        return rng.uniform(-lim, lim, size=shape)
    if name == "he_normal":
        return _trunc_normal(rng, shape, math.sqrt(2.0 / fi) / 0.87962566103423978)
    if name == "random_uniform":
        return rng.uniform(-0.05, 0.05, size=shape)
    if name == "random_normal":
        return rng.normal(0.0, 0.05, size=shape)
    if name == "zeros":
        return np.zeros(shape)
    if name == "ones":
        return np.ones(shape)
    raise ValueError(f"unknown initializer {name!r}")


def _trunc_normal(rng, shape, std):
    x = rng.normal(0.0, std, size=shape)
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rng.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x


def initialise(identifier, shape, rng):
    """identifier: Keras-style name | callable(shape, rng) or callable(shape) | ndarray of `shape`."""
    if isinstance(identifier, str):
        out = _make(identifier.lower(), shape, rng)
    elif isinstance(identifier, np.ndarray):
        if tuple(identifier.shape) != tuple(shape):
            raise ValueError(f"initial value has shape {identifier.shape}, expected {shape}")
        out = identifier
    elif callable(identifier):
        try:
            out = identifier(shape, rng)
        except TypeError:
            out = identifier(shape)
    else:
        raise ValueError(f"Could not interpret initializer identifier: {identifier!r}")
    return np.ascontiguousarray(out, dtype=np.float32)
