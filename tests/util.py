"""Shared helpers for the parity tests."""
from __future__ import annotations

import functools
import os

import numpy as np
import torch

from tests.golden import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def as_torch(w, dtype=torch.float32):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in w.items()}


@functools.lru_cache(maxsize=2)
def net_weights_torch(net: str):
    return as_torch(cases.net_weights(net))


def max_abs(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def grad_scales(og):
    """max|g| per tensor of the oracle's gradients, the scale a gradient error is held to.  One exception: the bias of
    a conv in front of a GroupNorm group of ONE channel (dim 8) has a gradient that is zero in exact arithmetic (the
    normalisation removes a per-channel shift), i.e. rounding noise of a cancelling sum on both sides; where the
    oracle's bias gradient is below 1e-4 of the same conv's weight gradient it is held to that weight gradient's scale."""
    out = {}
    for k, v in og.items():
        s = float(v.abs().max())
        wk = k[:-len("bias")] + "weight"
        if k.endswith(".block.0.bias") and wk in og and s < 1e-4 * float(og[wk].abs().max()):
            s = float(og[wk].abs().max())
        out[k] = max(s, 1e-12)
    return out
