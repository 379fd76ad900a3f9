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
