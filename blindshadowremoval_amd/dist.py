"""Data-parallel sharding of the batched forward: one process per GPU, contiguous batch split, and ONE
collective per batch — an all-gather (RCCL over xGMI on MI355X; gloo in the CPU tests) that re-assembles the
outputs callers consume on every rank.  GSC inference has no cross-sample op (BatchNorm uses moving
statistics, ShareLayer is never called: /root/reference/model.py:221 vs :228-290), so no other exchange exists.

`ShardedGenerator` is the product form of what `bench.py` times at N > 1: the tail kernel writes con_rgb | dif straight into the
`[B,H,W,4]` all-gather payload (`Generator(..., packed_out=)` = bsr_forward_packed), the gather is asynchronous and double-buffered,
so batch k's gather runs beside batch k + 1's forward.  The data-parallel form of the reference's LOOPS (`FSRNet.test` /
`testFFHQ` over a sharded name list) lives in fsrnet.py and needs no data-path collective at all.
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


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size) in `group` (the default group when None); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


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


class _Ticket:
    """One submitted global batch: the gather(s) in flight and how to cut the result."""
    __slots__ = ("slot", "counts", "cmax", "work", "work2", "has2", "done")

    def __init__(self, slot, counts, cmax, work, work2, has2):
        self.slot, self.counts, self.cmax, self.work, self.work2, self.has2, self.done = slot, counts, cmax, work, work2, has2, False


class ShardedGenerator:
    """Runs `gen(inputs, uv)` on this rank's contiguous shard of a global batch and all-gathers `con_rgb | dif` (the two outputs
    the reference's test loops consume: train_test_GSC.py:871-873) — optionally `gs | mask22` too — so every rank holds the global
    result.

        sg = ShardedGenerator(gen)                       # under an initialised process group (torchrun: one process per GPU)
        con_rgb, dif = sg.forward_global(inputs, uv)     # blocking form
        t = sg.submit(inputs_k, uv_k)                    # pipelined form: the gather of batch k ...
        t2 = sg.submit(inputs_k1, uv_k1)                 # ... overlaps the forward of batch k + 1 (two payload buffers)
        con_rgb, dif = sg.result(t)

    `packed`: the generator accepts `packed_out=` and writes con_rgb | dif into it from its tail kernel (blindshadowremoval_amd.Generator
    does: bsr_forward_packed); None = detect (`Generator` instances, or objects with `accepts_packed_out = True`).  Without it the
    payload is assembled by a `torch.cat` pass."""

    def __init__(self, gen: Callable, group=None, packed: Optional[bool] = None):
        self.gen = gen
        self.group = group
        self.rank, self.world = rank_world(group)
        if packed is None:
            packed = bool(getattr(gen, "accepts_packed_out", False))
            if not packed:
                try:
                    from .model import Generator
                    packed = isinstance(gen, Generator) and not type(gen).__name__.endswith("TSM")
                except Exception:       # the HIP library is not needed to shard a stand-in generator (CPU tests)
                    packed = False
        self.packed = packed
        self._payload = [None, None]            # [cmax,H,W,4] con_rgb | dif of the local shard, two slots
        self._payload2 = [None, None]           # [cmax,H,W,4] gs | mask22 (only when asked for)
        self._gathered = [None, None]
        self._gathered2 = [None, None]
        self._pending: List[Optional[_Ticket]] = [None, None]
        self._turn = 0

    @staticmethod
    def _buf(old, shape, like):
        if old is None or tuple(old.shape) != tuple(shape) or old.device != like.device or old.dtype != like.dtype:
            return torch.empty(shape, dtype=like.dtype, device=like.device)
        return old

    def submit(self, inputs: torch.Tensor, uv: torch.Tensor, want_gs_mask22: bool = False) -> _Ticket:
        """inputs / uv: the GLOBAL batch (the same tensors on every rank).  Runs the local shard and starts the gather(s).  Every rank
        pays for the whole batch's host-to-device traffic this way: a loader that produces only its rank's rows uses `submit_shard`."""
        bounds = shard_bounds(inputs.shape[0], self.world)
        lo, hi = bounds[self.rank]
        return self._submit_rows(inputs[lo:hi], uv[lo:hi], [b - a for a, b in bounds], want_gs_mask22)

    def submit_shard(self, inputs: torch.Tensor, uv: torch.Tensor, global_n: Optional[int] = None, want_gs_mask22: bool = False) -> _Ticket:
        """inputs / uv: THIS RANK'S rows only — rows `shard_bounds(global_n, world)[rank]` of the global batch (what a per-rank loader, e.g.
        `Dataset.shard`, hands over: 1 / world of `submit`'s input traffic per rank; the reference's loop feeds one device the whole batch,
        /root/reference/train_test_GSC.py:854-858).  `global_n`: the global row count when every rank knows it (the split must then be
        `shard_bounds`'); None = the ranks exchange their row counts first (one small object all-gather; any split, empty shards included).
        The result is the same global tensors `submit` gives, in rank order."""
        if global_n is not None:
            bounds = shard_bounds(global_n, self.world)
            counts = [b - a for a, b in bounds]
            if inputs.shape[0] != counts[self.rank]:
                raise ValueError("submit_shard: rank %d holds %d rows, shard_bounds(%d, %d) gives it %d" % (self.rank, inputs.shape[0], global_n, self.world, counts[self.rank]))
        elif self.world > 1:
            counts = [None] * self.world
            dist.all_gather_object(counts, int(inputs.shape[0]), group=self.group)
        else:
            counts = [int(inputs.shape[0])]
        if uv.shape[0] != inputs.shape[0]:
            raise ValueError("submit_shard: inputs and uv differ in their row count")
        return self._submit_rows(inputs, uv, counts, want_gs_mask22)

    def _submit_rows(self, x: torch.Tensor, u: torch.Tensor, counts: List[int], want_gs_mask22: bool) -> _Ticket:
        """x / u: this rank's rows (counts[rank] of them; may be none); counts: every rank's row count."""
        rows = counts[self.rank]
        cmax = max(max(counts), 1)
        slot = self._turn
        self._turn ^= 1
        old = self._pending[slot]
        if old is not None and not old.done:            # the buffers of two submissions ago: free once their gather completed
            self._wait(old)
        H, W = x.shape[1], x.shape[2]
        pay = self._payload[slot] = self._buf(self._payload[slot], (cmax, H, W, 4), x)
        pay2 = None
        if want_gs_mask22:
            pay2 = self._payload2[slot] = self._buf(self._payload2[slot], (cmax, H, W, 4), x)
        if rows > 0:
            x, u = x.contiguous(), u.contiguous()
            if self.packed:
                gs, _, mask22, _ = self.gen(x, u, packed_out=pay[:rows])
            else:
                gs, con_rgb, mask22, dif = self.gen(x, u)
                torch.cat([con_rgb, dif], dim=3, out=pay[:rows])
            if pay2 is not None:
                torch.cat([gs, mask22], dim=3, out=pay2[:rows])
        if rows < cmax:                                 # ragged split: the pad rows travel as zeros
            pay[rows:].zero_()
            if pay2 is not None:
                pay2[rows:].zero_()
        work = work2 = None
        if self.world > 1:
            g = self._gathered[slot] = self._buf(self._gathered[slot], (self.world * cmax, H, W, 4), x)
            work = dist.all_gather_into_tensor(g, pay, group=self.group, async_op=True)
            if pay2 is not None:
                g2 = self._gathered2[slot] = self._buf(self._gathered2[slot], (self.world * cmax, H, W, 4), x)
                work2 = dist.all_gather_into_tensor(g2, pay2, group=self.group, async_op=True)
        t = _Ticket(slot, counts, cmax, work, work2, pay2 is not None)
        self._pending[slot] = t
        return t

    def _wait(self, t: _Ticket) -> None:
        for w in (t.work, t.work2):
            if w is not None:
                w.wait()
        t.done = True

    def _cut(self, full: torch.Tensor, t: _Ticket) -> torch.Tensor:
        if all(c == t.cmax for c in t.counts):
            return full
        return torch.cat([full[r * t.cmax:r * t.cmax + c] for r, c in enumerate(t.counts)], dim=0)

    def result(self, t: _Ticket):
        """Wait for ticket `t`; -> (con_rgb, dif) of the GLOBAL batch, or (con_rgb, dif, gs, mask22) when it was submitted with
        want_gs_mask22.  The tensors are views of this object's buffers: valid until the submission after next."""
        self._wait(t)
        if self.world == 1:
            n = t.counts[0]
            full, full2 = self._payload[t.slot][:n], (self._payload2[t.slot][:n] if t.has2 else None)
        else:
            full = self._cut(self._gathered[t.slot], t)
            full2 = self._cut(self._gathered2[t.slot], t) if t.has2 else None
        if full2 is None:
            return full[..., :3], full[..., 3:4]
        return full[..., :3], full[..., 3:4], full2[..., :1], full2[..., 1:4]

    def forward_global(self, inputs: torch.Tensor, uv: torch.Tensor, want_gs_mask22: bool = False):
        """Blocking form: (con_rgb, dif[, gs, mask22]) for the global batch on every rank."""
        return self.result(self.submit(inputs, uv, want_gs_mask22))
