"""Data-parallel sharding of the batched forward: one process per GPU, contiguous batch split, and ONE
collective per batch — an all-gather (RCCL over xGMI on MI355X; gloo in the CPU tests) that re-assembles the
outputs callers consume on every rank.  GSC inference has no cross-sample op (BatchNorm uses moving
statistics, ShareLayer is never called: /root/reference/model.py:221 vs :228-290), so no other exchange exists.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced split of n items over `world` ranks (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def all_gather_rows(local: torch.Tensor, counts: Sequence[int], group=None, async_op: bool = False,
                    out: Optional[torch.Tensor] = None):
    """All-gather a ragged leading dimension: rank r contributes counts[r] rows.  Shards are padded to the
    largest count so a single all_gather_into_tensor moves everything; returns (gathered [sum(counts), ...], work)."""
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    cmax = max(counts)
    if local.shape[0] < cmax:
        pad = torch.zeros((cmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    buf = out if out is not None else torch.empty((world * cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(buf, local, group=group, async_op=async_op)

    def finish() -> torch.Tensor:
        if work is not None:
            work.wait()
        if all(c == cmax for c in counts):
            return buf
        return torch.cat([buf[r * cmax:r * cmax + c] for r, c in enumerate(counts)], dim=0)
    return finish, work


class ShardedGenerator:
    """Runs `gen(inputs, uv)` on this rank's contiguous shard of a global batch and all-gathers
    `cat[con_rgb, dif]` (the two outputs the reference's test loops consume: train_test_GSC.py:871-873)."""

    def __init__(self, gen: Callable, group=None):
        self.gen = gen
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def forward_global(self, inputs: torch.Tensor, uv: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """inputs/uv: the GLOBAL batch (same on every rank).  Returns (con_rgb, dif) for the global batch."""
        n = inputs.shape[0]
        bounds = shard_bounds(n, self.world)
        lo, hi = bounds[self.rank]
        counts = [b - a for a, b in bounds]
        if hi > lo:
            _, con_rgb, _, dif = self.gen(inputs[lo:hi].contiguous(), uv[lo:hi].contiguous())
            local = torch.cat([con_rgb, dif], dim=3)
        else:
            local = torch.zeros((0,) + tuple(inputs.shape[1:3]) + (4,), dtype=inputs.dtype, device=inputs.device)
        if self.world == 1:
            full = local
        else:
            finish, _ = all_gather_rows(local, counts, self.group)
            full = finish()
        return full[..., :3], full[..., 3:4]
