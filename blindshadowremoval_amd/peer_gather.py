"""Full-mesh DIRECT all-gather of the output payload by peer-to-peer copies — the opt-in alternative to the RCCL all-gather of
dist.ShardedGenerator / bench.py (round 5; SURVEY §5 / §8e: "full-mesh direct all-gather: each link carries one shard").

RCCL's all-gather runs as compute kernels beside the forward's one-round 256-workgroup launches; how much those kernels take from
the trunk on a real 8-GPU node has never been measured (no multi-GPU hardware was available to the builder).  This form uses no
compute unit for the exchange: every rank WRITES its shard into slot `rank` of every peer's gather buffer with a device-to-device
copy (hipMemcpyAsync between IPC-mapped buffers: the copy engines over the 7 point-to-point xGMI links, one shard per link — no ring,
no relay), on its own stream, behind an event of the forward that produced the shard.

  * the gather buffers are exchanged ONCE as HIP IPC handles (torch's CUDA-IPC storage sharing; HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf);
  * completion is a host-side barrier on a CONTROL group (gloo) after this rank's own copies have finished: when every rank has
    passed it, every shard has landed everywhere;
  * reuse rule (two slots): `finish()` of step k is called before `push()` of step k + 1, and the data of step k - 1 must have been
    consumed by then — the barrier inside finish(k) therefore orders every rank's reads of slot (k+1) & 1 before anyone's next write.

Verified like the RCCL path (every rank checks its own shard bit for bit and the others' by checksum; bench.py --gather peer).  The
world-2 self-test (`python -m torch.distributed.run --nproc-per-node 2 -m blindshadowremoval_amd.peer_gather --device 0`) runs both
ranks on ONE GPU — the IPC mapping and the protocol are the same, only the copies do not cross xGMI.
"""
from __future__ import annotations

import os
import sys
from typing import List, Optional

import torch
import torch.distributed as dist


class PeerGather:
    def __init__(self, shard_shape, dtype=torch.float32, device: Optional[torch.device] = None, control_group=None, slots: int = 2):
        """shard_shape: the shape of ONE rank's payload (equal on all ranks).  control_group: a gloo (CPU) process group spanning the
        ranks — object exchange of the IPC handles and the completion barriers; None = the default group (must then be gloo)."""
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("PeerGather needs an initialised process group")
        self.group = control_group
        self.rank, self.world = dist.get_rank(control_group), dist.get_world_size(control_group)
        if dist.get_backend(control_group) != "gloo":
            raise ValueError("PeerGather's control group must be a gloo group (host-side barriers and object exchange; no device collective is used)")
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.shard_shape, self.dtype, self.slots = tuple(shard_shape), dtype, int(slots)
        n = self.shard_shape[0]
        # ONE allocation per rank: [slots][world * n, ...]; its IPC handle goes to every peer
        self.local = torch.empty((self.slots, self.world * n) + self.shard_shape[1:], dtype=dtype, device=self.device)
        handle = self.local.untyped_storage()._share_cuda_()
        handles: List = [None] * self.world
        dist.all_gather_object(handles, handle, group=control_group)
        self.peers: List[Optional[torch.Tensor]] = []
        self._keep = []
        for r, h in enumerate(handles):
            if r == self.rank:
                self.peers.append(self.local)
                continue
            st = torch.UntypedStorage._new_shared_cuda(*h)
            self._keep.append(st)
            t = torch.empty(0, dtype=dtype, device=st.device).set_(st, 0, self.local.shape, self.local.stride())
            self.peers.append(t)
        self.stream = torch.cuda.Stream(device=self.device)
        self._done: Optional[torch.cuda.Event] = None
        self._reads: list = []                              # events behind this rank's asynchronous reads of gathered buffers (consumed())
        dist.barrier(group=control_group)                  # every mapping exists before anyone writes

    def push(self, slot: int, payload: torch.Tensor) -> None:
        """Enqueue the copies of this rank's shard into slot `slot` of every rank's buffer (its own included), on the copy stream,
        behind everything already queued on the CURRENT stream (the forward that wrote `payload`).  Returns at once."""
        if tuple(payload.shape) != self.shard_shape or payload.dtype != self.dtype:
            raise ValueError("payload must be %s %s" % (self.shard_shape, self.dtype))
        n = self.shard_shape[0]
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            for k in range(self.world):
                p = (self.rank + 1 + k) % self.world          # start with the next rank: the links are used evenly at every moment
                self.peers[p][slot, self.rank * n:(self.rank + 1) * n].copy_(payload, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record()
        payload.record_stream(self.stream)

    def finish(self) -> None:
        """Blocks until EVERY rank's last push has landed everywhere: this rank's copies have completed, then the control barrier.  Also waits
        for this rank's recorded READS of gathered buffers (`consumed()`): after the barrier a peer may push into a slot again, and GPU work
        of this rank that still reads it would race with those writes — the barrier orders hosts, not this rank's compute stream."""
        if self._done is not None:
            self._done.synchronize()
            self._done = None
        for ev in self._reads:
            ev.synchronize()
        self._reads = []
        dist.barrier(group=self.group)

    def gathered(self, slot: int) -> torch.Tensor:
        """[world * n, ...]: valid between the finish() after its push and the finish() before the next push into this slot.  A caller
        whose GPU work reads it asynchronously calls `consumed()` after enqueueing that work (host-side reads — .cpu(), .item() — need not)."""
        return self.local[slot]

    def consumed(self) -> None:
        """Record, on the CURRENT stream, that every read of gathered buffers enqueued so far is what the next finish() must wait for."""
        ev = torch.cuda.Event()
        ev.record()
        self._reads.append(ev)

    def close(self) -> None:
        dist.barrier(group=self.group)                     # nobody unmaps while a peer may still write
        self.peers, self._keep = [], []


def _selftest(argv=None) -> int:
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=None, help="GPU of every rank (default: LOCAL_RANK): with one GPU, all ranks share it")
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = (4, 64, 64, 4)
    pg = PeerGather(shape, torch.float32, torch.device("cuda", dev))
    ok = True
    for step in range(args.steps):
        slot = step & 1
        g = torch.Generator(device="cpu").manual_seed(1000 * step + rank)
        mine = torch.rand(shape, generator=g).cuda(dev)
        pg.push(slot, mine)
        pg.finish()
        chk = pg.gathered(slot).double().sum()            # an asynchronous GPU-side reader of the gathered buffer ...
        pg.consumed()                                     # ... which the next finish() must wait for before peers may overwrite the slot
        got = pg.gathered(slot).cpu()
        ok = ok and abs(float(chk) - float(got.double().sum())) < 1e-6
        for r in range(world):
            want = torch.rand(shape, generator=torch.Generator(device="cpu").manual_seed(1000 * step + r))
            ok = ok and bool(torch.equal(got[r * shape[0]:(r + 1) * shape[0]], want))
    oks = [None] * world
    dist.all_gather_object(oks, ok)
    pg.close()
    if rank == 0:
        print(json.dumps({"peer_gather_selftest": all(oks), "world": world, "steps": args.steps, "device_of_every_rank": dev}))
    dist.destroy_process_group()
    return 0 if all(oks) else 1


if __name__ == "__main__":
    sys.exit(_selftest())
