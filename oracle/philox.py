"""Oracle (test infrastructure — see oracle/__init__.py) for the BUILD'S OWN noise generator.

The reference draws noise with ``torch.randn`` (diffusion.py:218,241; policies.py:100,134);
its value stream is device- and version-specific, so parity tests inject noise instead.  The
throughput path uses a counter-based generator inside the posterior kernel
(``csrc/pointwise.hpp``: Philox4x32-10 + Box-Muller).  This file restates that generator in
numpy so the kernel can be checked: the integer stage is pinned by the Random123 known-answer
vectors, the float stage is compared with a tolerance (device logf/sinf/cosf are not bit-equal
to numpy's).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32 (10 rounds)."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0 = np.broadcast_to(key[..., 0], c[0].shape).astype(np.uint32).copy()
    k1 = np.broadcast_to(key[..., 1], c[0].shape).astype(np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0]
            p1 = M1 * c[2]
            n0 = (p1 >> np.uint64(32)) ^ c[1] ^ k0.astype(np.uint64)
            n1 = p1 & MASK32
            n2 = (p0 >> np.uint64(32)) ^ c[3] ^ k1.astype(np.uint64)
            n3 = p0 & MASK32
            c = [n0 & MASK32, n1, n2 & MASK32, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def normal(elem_index: np.ndarray, draw: int, seed: int) -> np.ndarray:
    """Standard normals for global element indices (uint64 array), as the kernel draws them:
    block = e >> 2 with counter (block_lo, block_hi, draw_lo, draw_hi), key = seed halves;
    pair = (e >> 1) & 1 selects (r0, r1) or (r2, r3); Box-Muller, cos for even e, sin for odd."""
    e = np.asarray(elem_index, dtype=np.uint64)
    q = e >> np.uint64(2)
    ctr = np.stack([(q & MASK32).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                    np.full(q.shape, draw & 0xFFFFFFFF, np.uint32),
                    np.full(q.shape, (draw >> 32) & 0xFFFFFFFF, np.uint32)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    r = philox4x32_10(ctr, key)
    pair = ((e >> np.uint64(1)) & np.uint64(1)).astype(np.int64)
    ra = np.take_along_axis(r, (2 * pair)[..., None], axis=-1)[..., 0]
    rb = np.take_along_axis(r, (2 * pair + 1)[..., None], axis=-1)[..., 0]
    u1 = ((ra >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    u2 = (rb >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(6.28318530717958647692) * u2).astype(np.float32)
    odd = (e & np.uint64(1)).astype(bool)
    return np.where(odd, rad * np.sin(ang), rad * np.cos(ang)).astype(np.float32)
