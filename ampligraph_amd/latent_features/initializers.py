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


# ---------------------------------------------------------------------------------------------------------------
# Row-range access to the initial tables (row-sharded entity tables): a rank draws only ITS rows
# and gets exactly the values of the whole-table draw above.  Works for the uniform family because
# Generator.uniform consumes one PCG64 output per element, so rows [lo, hi) of a table that starts `skip` outputs
# into the stream are reached with bit_generator.advance(skip + lo * cols) -- O(log) time, no memory.
_UNIFORM_LIMITS = {"glorot_uniform": lambda fi, fo: math.sqrt(6.0 / (fi + fo)), "he_uniform": lambda fi, fo: math.sqrt(6.0 / fi),
                   "random_uniform": lambda fi, fo: 0.05}
_CONSTANTS = {"zeros": 0.0, "ones": 1.0}


def stream_cost(identifier, shape):
    """PCG64 outputs a whole-table draw of `identifier` consumes, or None when that is not a fixed number
    (normal family: rejection sampling; callables: unknown)."""
    if isinstance(identifier, np.ndarray):
        return 0
    if isinstance(identifier, str):
        name = identifier.lower()
        if name in _UNIFORM_LIMITS:
            return int(shape[0]) * int(shape[1])
        if name in _CONSTANTS:
            return 0
    return None


class RowSource:
    """rows(lo, hi) -> fp32 (hi-lo, cols) block of the table `identifier` would initialise to, given that the table
    starts `skip` outputs into PCG64(seed)'s stream.  `streams` is False when the only way is to draw the whole
    table once on the host (then kept; fine for tables that fit host RAM)."""

    def __init__(self, identifier, shape, seed, skip=0):
        self.identifier, self.shape, self.seed, self.skip = identifier, (int(shape[0]), int(shape[1])), seed, skip
        self._whole = None
        name = identifier.lower() if isinstance(identifier, str) else None
        self._lim = _UNIFORM_LIMITS[name](*self.shape) if name in _UNIFORM_LIMITS else None
        self._const = _CONSTANTS.get(name)
        self.streams = skip is not None and (self._lim is not None or self._const is not None or isinstance(identifier, np.ndarray))
        if isinstance(identifier, np.ndarray) and tuple(identifier.shape) != self.shape:
            raise ValueError(f"initial value has shape {identifier.shape}, expected {self.shape}")

    def rows(self, lo, hi):
        lo, hi = int(lo), int(hi)
        cols = self.shape[1]
        if self._lim is not None and self.skip is not None:
            bg = np.random.PCG64(self.seed)
            bg.advance(self.skip + lo * cols)
            return np.random.Generator(bg).uniform(-self._lim, self._lim, size=(hi - lo, cols)).astype(np.float32)
        if self._const is not None:
            return np.full((hi - lo, cols), self._const, dtype=np.float32)
        if isinstance(self.identifier, np.ndarray):
            return np.ascontiguousarray(self.identifier[lo:hi], dtype=np.float32)
        if self._whole is None:
            if self.skip is None:
                raise ValueError("this initialiser follows one whose stream length is unknown: draw the tables in order")
            bg = np.random.PCG64(self.seed)
            bg.advance(self.skip)
            self._whole = initialise(self.identifier, self.shape, np.random.Generator(bg))
        return self._whole[lo:hi]
