"""Data-parallel FSRNet.test / testFFHQ (round 4) on CPU: world-size-2 gloo process groups around a stand-in generator.  The
union of the ranks' results must equal the single-process loop item for item — names, losses, PNG bytes — ragged splits included,
the gathered running means must be the single-process ones, and every PNG must be written by the rank that owns the item.
(The generator here is a labelled stand-in: the product path has no CPU generator.  The GPU counterpart at world 1 over RCCL is
tests/test_fsrnet.py::test_data_parallel_loop_world1_rccl.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class StandInGenerator:
    """The call surface FSRNet drives (model.Generator), computed by a few torch-CPU ops per row — deterministic, row-independent."""
    _device = None
    dtype = "f32"
    _handle = 1

    def __call__(self, inputs, uv, reg=None, chuck=1, training=False):
        g0 = inputs.mean(dim=3, keepdim=True)
        con_rgb = inputs * 0.8 + uv * 0.1 + 0.05
        dif = (con_rgb.mean(dim=3, keepdim=True) - g0) * 3.0
        return g0 + dif, con_rgb, torch.cat([dif.clamp(min=0), dif * 0, (-dif).clamp(min=0)], 3), dif

    def close(self):
        pass


def _config(out_dir):
    from blindshadowremoval_amd.fsrnet import Config
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = out_dir
    cfg.DATA_DIR_TEST = [os.path.join(GOLDEN, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(GOLDEN, "UCB_masks")
    return cfg


def _run(out_dir, n, ucb, plain_feed=False):
    """One loop over the first n UCB fixture items in this process (under whatever process group is initialised)."""
    import contextlib
    import io
    from blindshadowremoval_amd.dataset import Dataset
    from blindshadowremoval_amd.fsrnet import FSRNet
    cfg = _config(out_dir)
    ds = Dataset(cfg, "test", ucb=True)
    ds.name_list = ds.name_list[:n]
    if plain_feed:                                  # an object with only .feed / .name_list (what the reference's dataset.py offers): no shard()
        class Plain:
            pass
        p = Plain()
        p.name_list, p.feed = ds.name_list, ds.feed
        ds = p
    fsr = FSRNet(cfg, gen=StandInGenerator())
    fsr.post_threads = 1
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = fsr.test(ds, batch=2) if ucb else fsr.testFFHQ(ds, batch=2)
    fsr.log.close()
    out = {"names": [r[0] for r in res], "losses": [dict(r[2]) if len(r) > 2 else {} for r in res], "saved": list(fsr.log.saved),
           "all_losses": [(n_, dict(l)) for n_, l in fsr.all_losses], "means": {k: v[0] / max(v[1], 1) for k, v in fsr.log.losses.items()},
           "stdout": buf.getvalue(), "timings": dict(fsr.timings)}
    return out


def _worker(rank, world, port, out_dir, n, ucb, plain_feed, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run(out_dir, n, ucb, plain_feed)))
    finally:
        dist.destroy_process_group()


def _spawn(tmp, n, ucb, plain_feed=False, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, n, ucb, plain_feed, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [got[r] for r in range(world)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n,ucb,plain", [(5, True, False), (7, False, False), (1, False, False), (3, True, True)])
def test_world2_loop_equals_the_single_process_loop(tmp_path, n, ucb, plain):
    from blindshadowremoval_amd.dist import shard_bounds
    single = _run(str(tmp_path / "single"), n, ucb)
    ranks = _spawn(str(tmp_path / "dp"), n, ucb, plain)
    bounds = shard_bounds(n, 2)
    # the union of the shards, in rank order, is the single-process list item for item
    assert ranks[0]["names"] + ranks[1]["names"] == single["names"]
    assert [len(r["names"]) for r in ranks] == [hi - lo for lo, hi in bounds]
    for a, b in zip(ranks[0]["losses"] + ranks[1]["losses"], single["losses"]):
        assert a == b                                                   # the same float values (the stand-in and the post-processing are deterministic)
    # every rank holds the gathered per-item losses and the single-process running means
    for r in ranks:
        assert r["all_losses"] == single["all_losses"]
        assert r["means"] == single["means"]
    # PNG strips: written by the rank that owns the item, byte-identical to the single-process files
    want = {os.path.basename(p): open(p, "rb").read() for p in single["saved"]}
    assert len(want) == n
    seen = {}
    for r in ranks:
        for p in r["saved"]:
            assert os.path.basename(p) not in seen
            seen[os.path.basename(p)] = open(p, "rb").read()
    assert seen == want
    # only rank 0 talks; its last progress line carries the global means
    assert ranks[1]["stdout"].strip() == ""
    if ucb:
        last = [ln for ln in ranks[0]["stdout"].replace("\n", "\r").split("\r") if "Testing" in ln][-1]
        want_last = [ln for ln in single["stdout"].replace("\n", "\r").split("\r") if "Testing" in ln][-1]
        assert last.strip() == want_last.strip() and ("Testing %d/%d" % (n, n)) in last
    assert all(r["timings"]["world"] == 2 for r in ranks)


def test_dataset_shard_keeps_the_sibling_draws(tmp_path):
    """rows > 1 draws random siblings from a seeded RNG in list order: a sharded feed must hand out the SAME elements."""
    from blindshadowremoval_amd.dataset import Dataset
    cfg = _config(str(tmp_path))
    full = Dataset(cfg, "test", ucb=True, rows=3, seed=5)
    full.name_list = full.name_list[:4]
    want = [next(full.feed) for _ in range(4)]
    part = Dataset(cfg, "test", ucb=True, rows=3, seed=5)
    part.name_list = part.name_list[:4]
    part.shard(2, 4)
    got = [next(part.feed) for _ in range(2)]
    with pytest.raises(StopIteration):
        next(part.feed)
    for a, b in zip(got, want[2:]):
        np.testing.assert_array_equal(a[0], b[0])
        assert a[2][0] == b[2][0]
    with pytest.raises(RuntimeError, match="after the first element"):
        part.shard(0, 1)
    with pytest.raises(ValueError):
        Dataset(cfg, "test", ucb=True).shard(3, 2)
