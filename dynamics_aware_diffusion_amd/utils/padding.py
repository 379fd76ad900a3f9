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
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out.index_copy_(axis, index, t)
    return out


def padding_plan(keys, transition_dim: int, dim: int, dim_mults: Sequence[int]):
    """key -> [(axis, index of every real entry along that axis in the padded tensor, padded size)] for every
    tensor of a TemporalUnet state_dict (reference keys without ``model.``, temporal_unet.py:155-197), plus the
    padded dim and level widths.  Keys of levels that need no padding get an empty list."""
    widths = [dim * m for m in dim_mults]
    padded = [padded_width(c) for c in widths]
    n = len(widths)
    keys = set(keys)
    plan: Dict[str, list] = {}

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
        w, b = [], []
        if cout is not None:
            io, po = one(cout)
            w.append((0, io, po)); b.append((0, io, po))
        if cin is not None:
            ii, pi = cat(cin)
            w.append((1, ii, pi))
        plan[key + ".weight"], plan[key + ".bias"] = w, b

    def vec(key: str, c: int) -> None:
        i, p = one(c)
        plan[key] = [(0, i, p)]

    def res_block(base: str, cin, cout: int) -> None:
        conv(base + ".blocks.0.block.0", cout, cin)
        vec(base + ".blocks.0.block.1.weight", cout); vec(base + ".blocks.0.block.1.bias", cout)
        conv(base + ".blocks.1.block.0", cout, [cout])
        vec(base + ".blocks.1.block.1.weight", cout); vec(base + ".blocks.1.block.1.bias", cout)
        vec(base + ".time_mlp.1.weight", cout); vec(base + ".time_mlp.1.bias", cout)
        if base + ".residual_conv.weight" in keys:
            conv(base + ".residual_conv", cout, cin)

    # time MLP: the sinusoid keeps its `dim` real columns in front (HipEngine pads the table the same way)
    plan["time_mlp.1.weight"] = [(1, torch.arange(dim), padded[0])]
    for k in ("time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias"):
        plan[k] = []

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
        i, p = one(co)                     # ConvTranspose1d weight is (in, out, k)
        plan[f"ups.{j}.2.conv.weight"] = [(0, i, p), (1, i, p)]
        plan[f"ups.{j}.2.conv.bias"] = [(0, i, p)]
        cx = co
    conv("final_conv.0.block.0", dim, [cx])
    vec("final_conv.0.block.1.weight", dim); vec("final_conv.0.block.1.bias", dim)
    conv("final_conv.1", None, [dim])
    missing = keys - set(plan)
    if missing:
        raise KeyError(f"unexpected keys in the denoiser state_dict: {sorted(missing)[:4]}")
    return plan, padded[0], padded


def pad_tensor(t: torch.Tensor, steps) -> torch.Tensor:
    """The padded form of one tensor (zeros outside the real entries), on the tensor's own device."""
    t = t.detach().float()
    for axis, index, size in steps:
        t = _scatter(t, axis, index.to(t.device), size)
    return t.contiguous()


def unpad_tensor(t: torch.Tensor, steps) -> torch.Tensor:
    """The real entries of a tensor in the padded layout (the inverse of pad_tensor on them)."""
    for axis, index, _ in steps:
        t = t.index_select(axis, index.to(t.device))
    return t.contiguous()


def pad_unet_state(state: Mapping[str, torch.Tensor], transition_dim: int, dim: int,
                   dim_mults: Sequence[int], device=None) -> Tuple[Dict[str, torch.Tensor], int, List[int]]:
    """(padded state_dict, padded dim, padded level widths) of a TemporalUnet state_dict.  ``device``: where the padded
    tensors should live (default: the host — what dad_model_load_weight takes; the GPU for a device-side refresh)."""
    plan, pdim, widths = padding_plan(state.keys(), transition_dim, dim, dim_mults)
    out: Dict[str, torch.Tensor] = {}
    for key, t in state.items():
        src = t.detach().float()
        src = src.cpu() if device is None else src.to(device)
        out[key] = pad_tensor(src, plan[key])
    return out, pdim, widths


class FlatPadding:
    """Many tensors padded — or un-padded — with ONE scatter / gather over flat vectors: the real tensors flattened side
    by side on one side, the padded tensors side by side (at ``offsets``, default: packed) on the other, ``index`` the
    position of every real entry in the padded vector (gaps between tensors stay zero).  This is how a net of zero-padded widths trains: the engine
    writes padded gradients into one flat buffer, ``gather`` hands autograd real-shaped views of one ``index_select`` of
    it, and after an optimiser step ``pad_flat`` rebuilds the padded parameter copies on the device."""

    def __init__(self, keys: Sequence[str], shapes: Sequence[Sequence[int]], plan: Mapping[str, list],
                 offsets: Sequence[int] = None, total: int = None):
        self.keys = list(keys)
        self.shapes = [tuple(int(d) for d in s) for s in shapes]
        self.padded_shapes: List[Tuple[int, ...]] = []
        self.offsets: List[int] = []
        index, off = [], 0
        for i, (key, shape) in enumerate(zip(self.keys, self.shapes)):
            ps = list(shape)
            for axis, _, size in plan[key]:
                ps[axis] = size
            n = int(torch.Size(ps).numel())
            off = int(offsets[i]) if offsets is not None else (off + 3) // 4 * 4        # (packed: 16-byte aligned tensors)
            index.append(unpad_tensor(torch.arange(n).view(ps), plan[key]).reshape(-1) + off)
            self.padded_shapes.append(tuple(ps))
            self.offsets.append(off)
            off += n
        self.sizes = [int(torch.Size(s).numel()) for s in self.padded_shapes]
        self.real_sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.total = int(total) if total is not None else off
        self.index = torch.cat(index) if index else torch.zeros(0, dtype=torch.long)
        self._on: Dict[str, torch.Tensor] = {}

    def index_on(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._on:
            self._on[key] = self.index.to(device)
        return self._on[key]

    def pad_flat(self, tensors: Sequence[torch.Tensor]) -> torch.Tensor:
        """The padded tensors side by side in one new vector (tensor k at ``offsets[k]``); differentiable."""
        flat = torch.cat([t.reshape(-1) for t in tensors]).float()
        return torch.zeros(self.total, dtype=torch.float32, device=flat.device).index_copy(0, self.index_on(flat.device), flat)

    def pad(self, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """The padded tensors as views of ``pad_flat``, in the order of ``keys``."""
        wide = self.pad_flat(tensors)
        return [wide[o:o + n].view(s) for o, n, s in zip(self.offsets, self.sizes, self.padded_shapes)]

    def gather(self, wide: torch.Tensor) -> List[torch.Tensor]:
        """Real-shaped views of ONE gather of the real entries out of the padded vector ``wide``."""
        real = wide.index_select(0, self.index_on(wide.device))
        return [v.view(s) for v, s in zip(real.split_with_sizes(self.real_sizes), self.shapes)]
