"""Portable synthetic tensors: integer hash -> float, identical on every machine.

No dataset or checkpoint of the reference exists offline, so weights,
conditions and injected noise for parity tests and for ``bench.py`` come from this
generator.  It only uses uint64 integer arithmetic plus exactly-rounded IEEE operations
(int -> float64 conversion, float64 add, one multiply, one cast to float32), so the GPU
box regenerates bit-identical tensors without the reference and without committing
weights.  Golden outputs under ``tests/golden`` were produced by the real reference on
exactly these tensors (see ``tests/golden/make_golden.py``).

Shapes follow the reference's ``model_state_dict`` schema (SURVEY.md Appendix C;
``/root/reference/m_diffuser/models/temporal_unet.py:135-197``).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    base = (int(seed) * 0x1000003 + h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return _splitmix64(np.array([base], dtype=np.uint64))[0]


def _u24(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n integers in [0, 2^24) for stream (seed, name), sub-stream ``lane``."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (_stream_base(seed, name) + idx * np.uint64(0xD1342543DE82EF95)
               + np.uint64(lane) * np.uint64(0xA0761D6478BD642F)) & _M64
    return (_splitmix64(key) >> np.uint64(40)).astype(np.int64)


def uniform(seed: int, name: str, shape: Sequence[int], bound: float = 1.0) -> np.ndarray:
    """float32 tensor ~ U(-bound, bound); exact in float64, one rounding to float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_u24(seed, name, n).astype(np.float64) + 0.5) / float(1 << 24)      # (0,1)
    return ((2.0 * u - 1.0) * float(bound)).astype(np.float32).reshape(shape)


def normal_like(seed: int, name: str, shape: Sequence[int]) -> np.ndarray:
    """float32 tensor, zero mean / unit variance, bell shaped (Irwin-Hall of 12 uniforms).

    Not a true Gaussian (support is +-6), but every value is a sum of exactly
    representable float64 terms, so it is bit-portable.  Used as *injected* noise in
    parity tests; throughput runs use the in-kernel Philox generator instead.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(12):
        acc += (_u24(seed, name, n, lane=lane + 1).astype(np.float64) + 0.5) / float(1 << 24)
    return (acc - 6.0).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# TemporalUnet / GaussianDiffusion parameter schema
# --------------------------------------------------------------------------------------

def unet_param_shapes(transition_dim: int, dim: int, dim_mults: Sequence[int],
                      kernel_size: int = 5, time_dim: int | None = None,
                      prefix: str = "") -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} of every learnable tensor of the denoiser.

    Key names and shapes follow the reference's module tree
    (``temporal_unet.py:155-197``), including its quirks: the
    decoder has ``len(dim_mults) - 1`` stages and every stage owns an upsampling
    transposed conv (``temporal_unet.py:184-191``).
    """
    time_dim = time_dim or dim
    k = kernel_size
    chans = [transition_dim] + [dim * m for m in dim_mults]
    pairs = list(zip(chans[:-1], chans[1:]))
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def add(name: str, *shape: int) -> None:
        out[prefix + name] = tuple(int(s) for s in shape)

    def conv_block(base: str, ci: int, co: int) -> None:
        add(base + ".block.0.weight", co, ci, k)
        add(base + ".block.0.bias", co)
        add(base + ".block.1.weight", co)
        add(base + ".block.1.bias", co)

    def res_block(base: str, ci: int, co: int) -> None:
        conv_block(base + ".blocks.0", ci, co)
        conv_block(base + ".blocks.1", co, co)
        add(base + ".time_mlp.1.weight", co, time_dim)
        add(base + ".time_mlp.1.bias", co)
        if ci != co:
            add(base + ".residual_conv.weight", co, ci, 1)
            add(base + ".residual_conv.bias", co)

    add("time_mlp.1.weight", 4 * time_dim, dim)
    add("time_mlp.1.bias", 4 * time_dim)
    add("time_mlp.3.weight", time_dim, 4 * time_dim)
    add("time_mlp.3.bias", time_dim)

    n_levels = len(pairs)
    for i, (ci, co) in enumerate(pairs):
        res_block(f"downs.{i}.0", ci, co)
        res_block(f"downs.{i}.1", co, co)
        if i < n_levels - 1:
            add(f"downs.{i}.2.conv.weight", co, co, 3)
            add(f"downs.{i}.2.conv.bias", co)
    mid = chans[-1]
    res_block("mid_block1", mid, mid)
    res_block("mid_block2", mid, mid)
    for j, (ci, co) in enumerate(reversed(pairs[1:])):
        res_block(f"ups.{j}.0", co * 2, ci)
        res_block(f"ups.{j}.1", ci, ci)
        add(f"ups.{j}.2.conv.weight", ci, ci, 4)        # ConvTranspose1d: (in, out, k)
        add(f"ups.{j}.2.conv.bias", ci)
    conv_block("final_conv.0", dim, dim)
    add("final_conv.1.weight", transition_dim, dim, 1)
    add("final_conv.1.bias", transition_dim)
    return out


def _fan_in(key: str, shape: Tuple[int, ...]) -> int:
    if len(shape) == 3:
        if ".2.conv.weight" in key and shape[2] == 4:      # transposed conv (in, out, k)
            return shape[1] * shape[2]
        return shape[1] * shape[2]
    if len(shape) == 2:
        return shape[1]
    return 1


def synth_unet_state(transition_dim: int, dim: int, dim_mults: Sequence[int],
                     seed: int = 0, kernel_size: int = 5, time_dim: int | None = None,
                     affine_jitter: float = 0.0, prefix: str = "",
                     gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic denoiser weights: conv/linear ~ U(+-gain/sqrt(fan_in)), GroupNorm
    gamma = 1 + jitter*U, beta = jitter*U (jitter 0 => PyTorch defaults 1/0)."""
    shapes = unet_param_shapes(transition_dim, dim, dim_mults, kernel_size, time_dim, prefix)
    state: "OrderedDict[str, np.ndarray]" = OrderedDict()
    fan_of_weight: Dict[str, int] = {}
    for key, shape in shapes.items():
        is_norm = ".block.1." in key
        if is_norm:
            if key.endswith("weight"):
                v = 1.0 + uniform(seed, key, shape, affine_jitter) if affine_jitter else \
                    np.ones(shape, np.float32)
            else:
                v = uniform(seed, key, shape, affine_jitter) if affine_jitter else \
                    np.zeros(shape, np.float32)
            state[key] = v.astype(np.float32)
            continue
        if key.endswith("weight"):
            fan = _fan_in(key, shape)
            fan_of_weight[key[:-len("weight")]] = fan
        else:
            fan = fan_of_weight[key[:-len("bias")]]
        state[key] = uniform(seed, key, shape, gain / float(np.sqrt(fan)))
    return state


def count_params(shapes: Dict[str, Tuple[int, ...]]) -> int:
    return int(sum(int(np.prod(s)) for s in shapes.values()))


def unet_flops_per_sample(transition_dim: int, dim: int, dim_mults: Sequence[int],
                          horizon: int, kernel_size: int = 5,
                          time_dim: int | None = None) -> int:
    """2*MACs of every Conv1d / ConvTranspose1d / Linear of one forward for ONE sample
    (GroupNorm/Mish/pointwise excluded) — the quantity SURVEY.md §8(d) calls ``f``."""
    time_dim = time_dim or dim
    k = kernel_size
    chans = [transition_dim] + [dim * m for m in dim_mults]
    pairs = list(zip(chans[:-1], chans[1:]))
    macs = dim * 4 * time_dim + 4 * time_dim * time_dim

    def res(ci: int, co: int, L: int) -> int:
        m = ci * co * k * L + co * co * k * L + time_dim * co
        if ci != co:
            m += ci * co * L
        return m

    L = horizon
    n_levels = len(pairs)
    for i, (ci, co) in enumerate(pairs):
        macs += res(ci, co, L) + res(co, co, L)
        if i < n_levels - 1:
            L //= 2
            macs += co * co * 3 * L
    mid = chans[-1]
    macs += 2 * res(mid, mid, L)
    for ci, co in reversed(pairs[1:]):
        macs += res(co * 2, ci, L) + res(ci, ci, L)
        macs += ci * ci * 4 * L                      # transposed conv: 4 taps per input col
        L *= 2
    macs += dim * dim * k * L + dim * transition_dim * L
    return 2 * macs


ARCHS = {
    # name: (observation_dim, action_dim, dim, dim_mults, T)   — SURVEY.md §8 header
    "pointmaze": (4, 2, 128, (1, 2, 4), 100),
    "halfcheetah": (17, 6, 256, (1, 4, 8), 1000),
    "door": (39, 28, 256, (1, 2, 4, 8), 1000),
    "tiny": (4, 2, 32, (1, 2, 4), 20),
}
