"""Zero-padded GroupNorm groups: how widths the conv-GEMM tiles cannot hold still run on them.

``GroupNorm(8, C)`` (temporal_unet.py:71) needs only ``C % 8 == 0`` — the reference trains ``--dim 48`` or
``--dim 96`` — while the tiles of ``csrc/conv_gemm.hpp`` want whole power-of-two groups of at least four
channels (a multiple of 32).  A level of another width is run as the next such width: every group gets zero
channels appended (``C/8 -> p = max(4, next power of two)``, channel ``c`` of group ``g`` moves to ``g * p + c``),
the weights / biases / gamma / beta of the padding are zero, so the padded net computes the same function —
its padding channels are exactly zero everywhere — provided GroupNorm's statistics count the real channels
only (``dad_model_set_group_channels``; the epilogue masks the padding in the variance).  This module builds the
padded tensors from a reference ``state_dict``; it runs on the host, once per engine build.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Sequence, Tuple

import torch


def padded_width(channels: int) -> int:
    """Width the engine runs a level of ``channels`` (a multiple of 8) at."""
    if channels < 8 or channels % 8:
        raise ValueError(f"GroupNorm(8, C) needs C % 8 == 0, got {channels}")
    cpg = channels // 8
    return 8 * max(4, 1 << (cpg - 1).bit_length())


def channel_index(channels: int) -> torch.Tensor:
    """Position of every real channel inside the padded width (group-major)."""
    cpg, p = channels // 8, padded_width(channels) // 8
    c = torch.arange(channels)
    return (c // cpg) * p + c % cpg


def _scatter(t: torch.Tensor, axis: int, index: torch.Tensor, size: int) -> torch.Tensor:
    shape = list(t.shape)
    shape[axis] = size
    out = torch.zeros(shape, dtype=t.dtype)
    out.index_copy_(axis, index, t)
    return out


def pad_unet_state(state: Mapping[str, torch.Tensor], transition_dim: int, dim: int,
                   dim_mults: Sequence[int]) -> Tuple[Dict[str, torch.Tensor], int, List[int]]:
    """(padded state_dict, padded dim, padded level widths) of a TemporalUnet state_dict (reference keys
    without ``model.``, temporal_unet.py:155-197).  Tensors of levels that need no padding pass through."""
    widths = [dim * m for m in dim_mults]
    padded = [padded_width(c) for c in widths]
    n = len(widths)
    out: Dict[str, torch.Tensor] = {}

    def one(c: int):                       # (index, padded size) of a tensor of c channels
        return channel_index(c), padded_width(c)

    def cat(cs: Sequence[int]):            # virtual concat [c0 | c1 | ...], each part padded on its own
        idx, off = [], 0
        for c in cs:
            i, p = one(c)
            idx.append(i + off)
            off += p
        return torch.cat(idx), off

    def conv(key: str, cout, cin) -> None:
        """Conv1d weight (co, ci, k) + bias; cin: None = external (the trajectory: no padding) or widths."""
        w, b = state[key + ".weight"].detach().cpu().float(), state[key + ".bias"].detach().cpu().float()
        if cout is not None:
            io, po = one(cout)
            w, b = _scatter(w, 0, io, po), _scatter(b, 0, io, po)
        if cin is not None:
            ii, pi = cat(cin)
            w = _scatter(w, 1, ii, pi)
        out[key + ".weight"], out[key + ".bias"] = w, b

    def norm(key: str, c: int) -> None:
        i, p = one(c)
        out[key + ".weight"] = _scatter(state[key + ".weight"].detach().cpu().float(), 0, i, p)
        out[key + ".bias"] = _scatter(state[key + ".bias"].detach().cpu().float(), 0, i, p)

    def res_block(base: str, cin, cout: int) -> None:
        conv(base + ".blocks.0.block.0", cout, cin)
        norm(base + ".blocks.0.block.1", cout)
        conv(base + ".blocks.1.block.0", cout, [cout])
        norm(base + ".blocks.1.block.1", cout)
        i, p = one(cout)
        out[base + ".time_mlp.1.weight"] = _scatter(state[base + ".time_mlp.1.weight"].detach().cpu().float(), 0, i, p)
        out[base + ".time_mlp.1.bias"] = _scatter(state[base + ".time_mlp.1.bias"].detach().cpu().float(), 0, i, p)
        if base + ".residual_conv.weight" in state:
            conv(base + ".residual_conv", cout, cin)

    # time MLP: the sinusoid keeps its `dim` real columns in front (HipEngine pads the table the same way)
    w1 = state["time_mlp.1.weight"].detach().cpu().float()
    out["time_mlp.1.weight"] = _scatter(w1, 1, torch.arange(dim), padded[0])
    for k in ("time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias"):
        out[k] = state[k].detach().cpu().float()

    cx = None                              # the trajectory: transition_dim channels, never padded
    for i, co in enumerate(widths):
        res_block(f"downs.{i}.0", cx if cx is None else [cx], co)
        res_block(f"downs.{i}.1", [co], co)
        if i < n - 1:
            conv(f"downs.{i}.2.conv", co, [co])
        cx = co
    res_block("mid_block1", [cx], cx)
    res_block("mid_block2", [cx], cx)
    for j in range(n - 1):
        lvl = n - 1 - j
        co = widths[lvl - 1]
        res_block(f"ups.{j}.0", [cx, widths[lvl]], co)
        res_block(f"ups.{j}.1", [co], co)
        # ConvTranspose1d weight is (in, out, k)
        key = f"ups.{j}.2.conv"
        i, p = one(co)
        out[key + ".weight"] = _scatter(_scatter(state[key + ".weight"].detach().cpu().float(), 0, i, p), 1, i, p)
        out[key + ".bias"] = _scatter(state[key + ".bias"].detach().cpu().float(), 0, i, p)
        cx = co
    conv("final_conv.0.block.0", dim, [cx])
    norm("final_conv.0.block.1", dim)
    conv("final_conv.1", None, [dim])
    missing = set(state) - set(out)
    if missing:
        raise KeyError(f"unexpected keys in the denoiser state_dict: {sorted(missing)[:4]}")
    return out, padded[0], padded
