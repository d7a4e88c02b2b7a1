"""N>1 path on CPU: world_size-2 gloo processes shard a global batch, run a stand-in per-row function in
place of the HIP generator (no GPU here) and all-gather the consumed outputs — ragged and even splits."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blindshadowremoval_amd.dist import ShardedGenerator, all_gather_rows, shard_bounds


def test_shard_bounds():
    assert shard_bounds(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)]
    for n in (0, 1, 7, 99, 100):
        b = shard_bounds(n, 8)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(7))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def fake_gen(inputs, uv, *a, **k):
    """Per-row deterministic stand-in with the generator's output shapes."""
    gs = inputs.mean(-1, keepdim=True)
    con_rgb = inputs * 2 + uv
    mask22 = uv * 0
    dif = (inputs - uv).sum(-1, keepdim=True)
    return gs, con_rgb, mask22, dif


class PackedFake:
    """fake_gen with the `packed_out=` keyword of blindshadowremoval_amd.Generator (bsr_forward_packed)."""
    accepts_packed_out = True

    def __call__(self, inputs, uv, packed_out=None):
        gs, con_rgb, mask22, dif = fake_gen(inputs, uv)
        assert packed_out is not None and tuple(packed_out.shape) == tuple(inputs.shape[:3]) + (4,) and packed_out.is_contiguous()
        packed_out[..., :3].copy_(con_rgb)
        packed_out[..., 3:].copy_(dif)
        return gs, packed_out[..., :3], mask22, packed_out[..., 3:]


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        inp = torch.rand(n, 8, 8, 3, generator=g)
        uv = torch.rand(n, 8, 8, 3, generator=g)
        con_rgb, dif = ShardedGenerator(fake_gen).forward_global(inp, uv)
        want_gs, want_rgb, want_m22, want_dif = fake_gen(inp, uv)
        ok = torch.equal(con_rgb, want_rgb) and torch.equal(dif, want_dif)
        # the pipelined form: two submissions in flight (double-buffered payloads), a third one recycles the first slot; the packed
        # stand-in writes con_rgb | dif straight into the payload like the HIP generator's tail kernel; gs | mask22 on request
        sg = ShardedGenerator(PackedFake())
        assert sg.packed
        t1 = sg.submit(inp, uv)
        t2 = sg.submit(inp * 0.5, uv, want_gs_mask22=True)
        r1 = sg.result(t1)
        ok = ok and torch.equal(r1[0], want_rgb) and torch.equal(r1[1], want_dif)
        r2 = sg.result(t2)
        g2 = fake_gen(inp * 0.5, uv)
        ok = ok and len(r2) == 4 and all(torch.equal(a, b) for a, b in zip(r2, (g2[1], g2[3], g2[0], g2[2])))
        t3 = sg.submit(inp, uv, want_gs_mask22=True)
        r3 = sg.result(t3)
        ok = ok and all(torch.equal(a, b) for a, b in zip(r3, (want_rgb, want_dif, want_gs, want_m22)))
        # the per-rank-shard form: every rank hands over ITS rows only (a per-rank loader) — the same global result; with the global row
        # count known, and with the ranks exchanging their counts (any split: here rank 0 takes all but one row)
        lo, hi = shard_bounds(n, world)[rank]
        r4 = sg.result(sg.submit_shard(inp[lo:hi], uv[lo:hi], global_n=n, want_gs_mask22=True))
        ok = ok and all(torch.equal(a, b) for a, b in zip(r4, (want_rgb, want_dif, want_gs, want_m22)))
        cut = max(n - 1, 0)
        mine = slice(0, cut) if rank == 0 else slice(cut, n)
        r5 = sg.result(sg.submit_shard(inp[mine], uv[mine]))
        ok = ok and torch.equal(r5[0], want_rgb) and torch.equal(r5[1], want_dif)
        try:
            sg.submit_shard(inp[:0] if hi > lo else inp[:1], uv[:0] if hi > lo else uv[:1], global_n=n)
            ok = False
        except ValueError:
            pass
        # async ragged gather
        counts = [hi - lo for lo, hi in shard_bounds(n, world)]
        finish, work = all_gather_rows(inp[lo:hi], counts, async_op=True)
        ok = ok and torch.equal(finish(), inp)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [4, 5, 1])
def test_world2_gloo_sharded_forward(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
