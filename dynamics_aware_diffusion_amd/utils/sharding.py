"""Batch sharding of independent plans over the GPUs of one node.

Trajectories of a batch never interact (convolutions and GroupNorm are per sample; the only
shared state is the weights and the scalar timestep — SURVEY.md §8(e)), so the sampler shards
rows contiguously over ranks with NO collective inside the reverse-diffusion loop.  The single
exchange is the gather of finished plans, ``(rows, H, td)`` fp32 per rank, over RCCL
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).

Noise does not depend on the sharding: the in-kernel Philox stream is indexed by the GLOBAL
row (``row_offset`` of ``dad_sample_loop`` / ``dad_fill_normal``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous ``(start, count)`` of rank ``rank``; the first ``total % world`` ranks take one
    extra row.  ``sum(count) == total_rows`` and ranges are ordered by rank."""
    if total_rows < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad shard request: rows={total_rows} world={world_size} rank={rank}")
    base, extra = divmod(total_rows, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def gather_plans(local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """All ranks receive the full ``(total_rows, H, td)`` batch in global row order.

    Equal shards use one ``all_gather_into_tensor``; ragged shards are padded to the largest
    shard for the collective and trimmed afterwards."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != total_rows:
            raise ValueError("single process: the local shard must be the whole batch")
        return local
    world = dist.get_world_size(group)
    counts = [shard_range(total_rows, world, r)[1] for r in range(world)]
    rank = dist.get_rank(group)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {counts[rank]}")
    widest = max(counts)
    tail = tuple(local.shape[1:])
    if min(counts) == widest:
        out = torch.empty((total_rows,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((widest,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    buf = torch.empty((world * widest,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = [buf[r * widest:r * widest + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def sample_sharded(policy, total_rows: int, conditions=None, group=None,
                   gather: bool = True) -> torch.Tensor:
    """Run ``policy.sample_loop`` on this rank's shard of a ``total_rows`` batch (global-row
    Philox offsets), then gather.  Per-row conditions ``(total_rows, td)`` are sliced."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    start, count = shard_range(total_rows, world, rank)
    local_cond = None
    if conditions is not None:
        local_cond = {}
        for k, v in conditions.items():
            rows = v.reshape(-1, v.shape[-1])
            local_cond[k] = rows[start:start + count] if rows.shape[0] == total_rows else v
    plans = policy.sample_loop(batch_size=count, conditions=local_cond, row_offset=start)
    return gather_plans(plans, total_rows, group) if gather else plans
