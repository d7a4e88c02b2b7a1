"""Harness counterpart of FSRNet.testFFHQ / test (train_test_GSC.py:360-422, 840-890) on the real-input fixture
built from /root/reference/sample_imgs/02165 by the reference's own input preparation (tools/make_sample_fixture.py)."""
import os

import numpy as np
import pytest
import torch

from blindshadowremoval_amd.weights import init_weights


class OneSampleDataset:
    """What dataset.py hands the loops: .name_list and .feed yielding (img[1,10,256,256,16], box[1,4], name)."""

    def __init__(self, golden_dir, copies=1):
        z = np.load(os.path.join(golden_dir, "sample_02165.npz"))
        self.row = z["row"]
        chunk = np.stack([self.row] * 10, axis=0)[None]                 # dataset.py:764-768, batch(1)
        self.name_list = ["sample_imgs/02165/02165.npy"] * copies
        self.feed = iter([(chunk, z["box"][None], np.array([b"sample_imgs/02165/02165.png"]))] * copies)


def test_fixture_layout(golden_dir):
    ds = OneSampleDataset(golden_dir)
    row = ds.row
    assert row.shape == (256, 256, 16) and row.dtype == np.float32
    np.testing.assert_array_equal(row[..., 0:3], row[..., 3:6])          # gt == img for FFHQ (dataset.py:625,638)
    assert float((row[..., 6:9] == 0).mean()) > 0.5                      # uv is 0 outside the landmark hull (warp.py:231)
    assert float(np.abs(row[..., 11]).max()) == 0 and float(np.abs(row[..., 14]).max()) == 0   # 3rd reg channels are x*0
    assert 0.0 <= row[..., 15].min() and row[..., 15].max() <= 1.0       # blurred face mask


def test_logging_strip(tmp_path):
    from blindshadowremoval_amd.fsrnet import Config, Logging
    cfg = Config()
    cfg.CHECKPOINT_DIR = str(tmp_path)
    log = Logging(cfg)
    a = torch.rand(2, 8, 8, 3) * 1.5 - 0.2
    m = torch.rand(2, 8, 8, 1)
    strip = log.get_imgs([a, m])
    assert strip.shape == (8, 16, 3) and strip.dtype == np.uint8
    np.testing.assert_array_equal(strip[:, :8], np.rint(np.clip(a[0].numpy(), 0, 1) * 255).astype(np.uint8))
    np.testing.assert_array_equal(strip[:, 8:, 0], strip[:, 8:, 2])      # grey replicated to 3 channels
    out = log.save_img([a, m], "sample_imgs/02165/02165.png")
    assert out.endswith("02165_02165-result.png") and os.path.isfile(out)


@pytest.mark.gpu
def test_testFFHQ_on_sample_matches_oracle(golden_dir, tmp_path):
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from oracle.gsc_oracle import GeneratorOracle, test_step_ffhq
    w = init_weights(1)
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    fsr = FSRNet(cfg, weights=w)
    results = fsr.testFFHQ(OneSampleDataset(golden_dir, copies=3), batch=2)     # 3 elements, batched 2 + 1
    assert len(results) == 3 and len(fsr.log.saved) == 3 and all(os.path.isfile(p) for p in fsr.log.saved)
    row = torch.from_numpy(OneSampleDataset(golden_dir).row)[None]
    ref = test_step_ffhq(GeneratorOracle(w), row)
    for name, figs in results:
        assert name == "sample_imgs/02165/02165.npy"
        for a, b in zip(figs, ref):
            assert float((a.cpu() - b).abs().max()) <= 1e-3
    # the reference's own 10-row element forward keeps row 0: identical to the row-0-only forward
    ds = OneSampleDataset(golden_dir)
    element = next(ds.feed)
    _, figs10 = fsr.test_step_FFHQ(element[0], element[1], training=False, all_rows=True)
    _, figs1 = fsr.test_step_FFHQ(element[0], element[1], training=False)
    assert figs10[1].shape[0] == 10 and figs1[1].shape[0] == 1
    for a, b in zip(figs10, figs1):
        assert torch.equal(a[:1], b)
    # UCB loop head (generator outputs)
    res = fsr.test(OneSampleDataset(golden_dir), batch=4, postprocess=False)
    assert len(res) == 1 and res[0][1][1].shape == (1, 256, 256, 1)
    with pytest.raises(FileNotFoundError, match="UCB mask folders"):        # the full UCB step needs the seven mask folders
        fsr.test(OneSampleDataset(golden_dir), batch=4)


@pytest.mark.gpu
def test_restore_from_tensor_bundle(tmp_path):
    """FSRNet restores generator weights from a TF tensor-bundle checkpoint directory (train_test_GSC.py:842-845)."""
    from blindshadowremoval_amd import tf_bundle
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    w = init_weights(2)
    tf_bundle.write_bundle(str(tmp_path / "ckpt-94"), w)
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    fsr = FSRNet(cfg)
    assert fsr._restore() == 94
    torch.manual_seed(1)
    inp, uv = torch.rand(1, 256, 256, 3), torch.rand(1, 256, 256, 3)
    from parity_util import run_and_compare
    run_and_compare(fsr.gen, w, inp, uv)
    empty = Config(0)
    empty.CHECKPOINT_DIR = str(tmp_path / "none")
    with pytest.raises(RuntimeError, match="no generator weights"):
        FSRNet(empty)._restore()


def test_roc_auc_known_values():
    from blindshadowremoval_amd.fsrnet import roc_auc_score
    assert roc_auc_score([0, 0, 1, 1], [0.1, 0.4, 0.35, 0.8]) == 0.75          # the sklearn docstring example
    assert roc_auc_score([1, 0, 1, 0], [0.9, 0.1, 0.8, 0.2]) == 1.0
    assert roc_auc_score([0, 1, 0, 1], [0.5, 0.5, 0.5, 0.5]) == 0.5             # all ties
    rng = np.random.default_rng(0)
    y = rng.integers(0, 2, 500)
    s = np.round(rng.random(500), 2)
    pos, neg = s[y == 1], s[y == 0]
    brute = ((pos[:, None] > neg[None]).sum() + 0.5 * (pos[:, None] == neg[None]).sum()) / (len(pos) * len(neg))
    assert abs(roc_auc_score(y, s) - brute) < 1e-12


@pytest.mark.gpu
def test_tsm_harness_steps(tmp_path):
    """testsfw (frame 2) and testsfw_video (frame 10) step functions against the TSM oracle."""
    from blindshadowremoval_amd.fsrnet import Config, FSRNetTSM, roc_auc_score
    from oracle.gsc_oracle import GeneratorTSMOracle
    w = init_weights(1, variant="tsm")
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    fsr = FSRNetTSM(cfg, weights=w)
    torch.manual_seed(21)

    def field(n, c):
        return torch.nn.functional.interpolate(torch.rand(n, c, 9, 9), size=(256, 256), mode="bicubic", align_corners=True).permute(0, 2, 3, 1)
    # SFW image element: [2,256,256,17] = img3, cmap3, mask1 in {0,1,2}, uv3, reg6, face1
    img, uv, face = field(2, 3).clamp(0, 1), field(2, 3).clamp(0, 1), field(2, 1).clamp(0, 1)
    reg = (field(2, 6) - 0.5) * 0.2
    mask = (field(2, 1) * 3).floor().clamp(0, 2)
    el = torch.cat([img, img, mask, uv, reg, face], dim=3)[None]
    losses, figs = fsr.test_step_sfw(el)
    ref = GeneratorTSMOracle(w)(img, uv, reg, 2, True)
    mp = ref[3] * face
    assert float((figs[1].cpu() - ref[1].clamp(0, 1)).abs().max()) <= 1e-3 and float((figs[2].cpu() - mp * 2).abs().max()) <= 2e-3
    want_auc = roc_auc_score(np.concatenate([[1, 0], (mask[0] == 2).float().numpy().reshape(-1)]), np.concatenate([[1, 0], mp[0].numpy().reshape(-1)]))
    assert abs(losses["auc"] - want_auc) < 1e-3 and 0.0 <= losses["auc"] <= 1.0
    # video element: [10,256,256,13]
    img, uv, face = field(10, 3).clamp(0, 1), field(10, 3).clamp(0, 1), field(10, 1).clamp(0, 1)
    reg = (field(10, 6) - 0.5) * 0.2
    _, figs = fsr.test_step_sfw_video(torch.cat([img, uv, reg, face], dim=3)[None])
    ref = GeneratorTSMOracle(w)(img, uv, reg, 10, True)
    assert figs[1].shape == (10, 256, 256, 3)
    assert float((figs[1].cpu() - ref[1].clamp(0, 1)).abs().max()) <= 1e-3


@pytest.mark.gpu
def test_tsm_loops_over_the_sfw_loaders(tmp_path, golden_dir):
    """The TSM script's two test loops end to end: `Dataset(config, 'test', dset='sfw' | 'sfw_video')` (pinned to the reference's
    own parsers by tests/test_dataset.py) -> `FSRNetTSM.testsfw` / `testsfw_video` (train_with_TSM.py:619-748) on the synthetic
    SFW folder, outputs against the TSM oracle on the same elements."""
    import os
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNetTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    w = init_weights(1, variant="tsm")
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "sfw_synth", "*")]
    fsr = FSRNetTSM(cfg, weights=w)
    res = fsr.testsfw(D.Dataset(cfg, "test", dset="sfw", workers=2))
    assert len(res) == 2 and all(0.0 <= r[1]["auc"] <= 1.0 and np.isfinite(r[1]["psnr"]) for r in res)
    el = torch.from_numpy(next(D.Dataset(cfg, "test", dset="sfw").feed)[0][0])                      # [2,256,256,17]
    img, _, _, uv, reg, face = torch.split(el, [3, 3, 1, 3, 6, 1], dim=3)
    ref = GeneratorTSMOracle(w)(img, uv, reg, 2, True)
    assert float((res[0][2][1].cpu() - ref[1].clamp(0, 1)).abs().max()) <= 1e-3
    vid = fsr.testsfw_video(D.Dataset(cfg, "test", dset="sfw_video", workers=2))
    assert len(vid) == 2 and vid[0][2][1].shape == (10, 256, 256, 3)
    el = torch.from_numpy(next(D.Dataset(cfg, "test", dset="sfw_video").feed)[0][0])                # [10,256,256,13]
    img, uv, reg, face = torch.split(el, [3, 3, 6, 1], dim=3)
    ref = GeneratorTSMOracle(w)(img, uv, reg, 10, True)
    assert float((vid[0][2][1].cpu() - ref[1].clamp(0, 1)).abs().max()) <= 1e-3
    assert len([f for f in os.listdir(os.path.join(str(tmp_path), "test")) if f.endswith("-result.png")]) == 2      # same names for both loops
    # several elements per forward (round 4; config[4]'s "batch = 64 frames" = 32 pairs / 6 ten-frame groups): each element is its own
    # ShareLayer group, so its outputs are those of its own forward, bit for bit
    res2 = fsr.testsfw(D.Dataset(cfg, "test", dset="sfw", workers=2), batch=2)
    vid2 = fsr.testsfw_video(D.Dataset(cfg, "test", dset="sfw_video", workers=2), batch=2)
    for one, many in ((res, res2), (vid, vid2)):
        assert [r[0] for r in one] == [r[0] for r in many] and [r[1] for r in one] == [r[1] for r in many]
        for a, b in zip(one, many):
            assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


@pytest.mark.gpu
def test_tsm_512_frames():
    """BASELINE config 5 names 512x512 frames: the kernels are size-generic (S = 64 attention over 4096 tokens)."""
    from blindshadowremoval_amd import GeneratorTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    w = init_weights(1, variant="tsm")
    gen = GeneratorTSM().load_weights(w)
    torch.manual_seed(3)
    inp, uv = torch.rand(2, 512, 512, 3), torch.rand(2, 512, 512, 3)
    reg = (torch.nn.functional.interpolate(torch.rand(2, 6, 9, 9), size=(512, 512), mode="bicubic", align_corners=True).permute(0, 2, 3, 1) - 0.5) * 0.1
    out = [t.cpu() for t in gen(inp.cuda(), uv.cuda(), reg.contiguous().cuda(), 2, True)]
    ref = GeneratorTSMOracle(w)(inp, uv, reg, 2, True, bmask_override=gen.probe("bmask").cpu())
    for a, b in zip(out, ref):
        assert float((a - b).abs().max()) <= 1e-3


def test_ucb_masks_are_indexed_strictly_by_item(golden_dir, tmp_path):
    """train_test_GSC.py:386-396 reads masks[count]: a name list longer than the mask list is an error (never a wrap-around),
    checked before any weight is needed — runs on CPU."""
    from blindshadowremoval_amd.fsrnet import Config, FSRNet

    class DS:
        name_list = ["a.npy"] * 21
        feed = iter(())
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    fsr = FSRNet(cfg)
    n_masks = len(fsr._ucb_masks())
    DS.name_list = ["a.npy"] * (n_masks + 1)
    with pytest.raises(ValueError, match="mask files"):
        fsr.test(DS())
    with pytest.raises(ValueError, match="mask files"):
        fsr.test(DS(), mask_files=fsr._ucb_masks()[:3])


def test_logging_flush_waits_for_all_writers_and_reraises(tmp_path):
    from blindshadowremoval_amd.fsrnet import Config, Logging
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    log = Logging(cfg, png_threads=2)
    img = torch.rand(1, 8, 8, 3)
    p1 = log.save_img([img], "x/one.png")
    bad = os.path.join(str(tmp_path), "test", "x_two-result.png")
    os.makedirs(bad)                                   # a directory where the PNG should go: this writer fails
    log.save_img([img], "x/two.png")
    p3 = log.save_img([img], "x/three.png")
    with pytest.raises(Exception):
        log.flush()
    assert os.path.isfile(p1) and os.path.isfile(p3)   # the good ones were still waited for
    log.close()


@pytest.mark.gpu
def test_device_prepared_loader_feeds_the_loops(golden_dir, tmp_path):
    """Dataset(device_prep=gpu): rows are prepared on the device (prep.py / csrc/prep_kernels.h) and reach the generator without a
    host round trip; FSRNet.testFFHQ on them gives the figures of the host-prepared path (inputs equal to 1e-6)."""
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    w = init_weights(1)
    outs = []
    for k, kw in enumerate((dict(workers=2), dict(workers=2, device_prep=0, device_batch=8))):
        cfg.CHECKPOINT_DIR = str(tmp_path / ("run%d" % k))
        ds = D.Dataset(cfg, "test", ucb=True, **kw)
        ds.name_list = ds.name_list[:20]
        fsr = FSRNet(cfg, weights=w)
        res = fsr.testFFHQ(ds, batch=16)
        ds.close()
        assert len(res) == 20 and all(os.path.isfile(f) for f in fsr.log.saved) and len(fsr.log.saved) == 20
        outs.append(res)
        fsr.close()
    for (n0, f0), (n1, f1) in zip(*outs):
        assert n0 == n1
        assert f1[0].is_cuda
        assert float((f0[0].cpu() - f1[0].cpu()).abs().max()) <= 1e-6          # the prepared image itself
        for a, b in zip(f0[1:], f1[1:]):
            assert float((a.cpu() - b.cpu()).abs().max()) <= 1e-3               # generator outputs on inputs that differ by <= 1e-6


@pytest.mark.gpu
def test_ucb_post_processing_in_worker_processes_equals_the_in_process_form(golden_dir, tmp_path):
    """FSRNet.test(post_workers=N): the reference's per-item post-processing runs in worker processes one batch behind the GPU;
    same code, same inputs => the same metrics and PNG strips (up to torch-CPU's thread-count-dependent rounding of the resizes)."""
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    w = init_weights(1)
    runs = []
    for k, (pw, figs) in enumerate(((0, True), (3, True), (3, False), (0, False))):      # (0, False): no pool, no figures back — the strips are still written
        cfg.CHECKPOINT_DIR = str(tmp_path / ("run%d" % k))
        ds = D.Dataset(cfg, "test", ucb=True, workers=2)
        ds.name_list = ds.name_list[:20]
        fsr = FSRNet(cfg, weights=w)
        fsr.post_device = False                                 # this test is about the HOST forms (in-process / worker pool); the device form: its own test below
        fsr.post_workers, fsr.return_figs = pw, figs
        res = fsr.test(ds, batch=8)
        ds.close()
        assert len(res) == 20 and len(fsr.log.saved) == 20
        runs.append((res, [open(f, "rb").read() for f in fsr.log.saved]))
        fsr.close()
    base_res, base_png = runs[0]
    for res, png in runs[1:]:
        assert [r[0] for r in res] == [r[0] for r in base_res]
        for r, b in zip(res, base_res):                                         # SSIM / PSNR, item by item
            assert abs(r[2]["ssim"] - b[2]["ssim"]) < 1e-5 and abs(r[2]["psnr"] - b[2]["psnr"]) < 1e-4
        if png != base_png:                                                     # the strips, byte for byte — say what differs
            from PIL import Image
            import io
            for i, (a, b) in enumerate(zip(png, base_png)):
                if a != b:
                    A, B = np.asarray(Image.open(io.BytesIO(a))).astype(int), np.asarray(Image.open(io.BytesIO(b))).astype(int)
                    cols = sorted(set((np.argwhere(A != B)[:, 1] // 256).tolist()))
                    # torch's CPU bilinear resize rounds differently with 1 thread (the workers) than with the parent's many: a handful of
                    # resized pixels land one grey level apart; anything more is a real difference
                    assert int(np.abs(A - B).max()) <= 1 and int((A != B).any(2).sum()) <= 64, (i, cols)
    assert all(r[1] is None for r in runs[2][0]) and all(r[1] is None for r in runs[3][0]) and all(len(r[1]) == 7 for r in runs[1][0])


@pytest.mark.gpu
def test_data_parallel_loop_world1_rccl(golden_dir, tmp_path):
    """`python -m blindshadowremoval_amd.run_loop` (the torchrun entry of the data-parallel loops) under a ONE-rank RCCL process group:
    FSRNet.testFFHQ shards the name list (one shard), gathers the per-item results with all_gather_object over RCCL and rank 0 reports;
    the world-2 logic is covered on CPU over gloo (tests/test_fsrnet_dp_cpu.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSR_LOOP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "blindshadowremoval_amd.run_loop", "--loop", "ffhq", "--data", os.path.join(golden_dir, "UCB", "train", "input", "*"),
           "--checkpoint-dir", str(tmp_path), "--random-weights", "1", "--batch", "8"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    n_items = len([f for d in os.listdir(os.path.join(golden_dir, "UCB", "train", "input")) for f in os.listdir(os.path.join(golden_dir, "UCB", "train", "input", d)) if f.endswith(".npy")])
    assert out["process_group"] == "nccl" and out["ranks"] == 1 and out["items"] == n_items and out["items_this_rank"] == n_items
    assert len([f for f in os.listdir(os.path.join(str(tmp_path), "test")) if f.endswith("-result.png")]) == n_items


@pytest.mark.gpu
def test_sharded_generator_packed_payload_on_the_gpu():
    """dist.ShardedGenerator over the HIP generator (world 1, no process group): the tail kernel writes con_rgb | dif into the payload
    (packed_out), the pipelined submit / result form double-buffers it, gs | mask22 on request — all equal to the plain forward."""
    from blindshadowremoval_amd import Generator
    from blindshadowremoval_amd.dist import ShardedGenerator
    gen = Generator(device=0).load_weights(init_weights(1))
    torch.manual_seed(4)
    a, ua = torch.rand(3, 256, 256, 3).cuda(), torch.rand(3, 256, 256, 3).cuda()
    b, ub = torch.rand(2, 256, 256, 3).cuda(), torch.rand(2, 256, 256, 3).cuda()
    want_a = [t.clone() for t in gen(a, ua)]
    want_b = [t.clone() for t in gen(b, ub)]
    sg = ShardedGenerator(gen)
    assert sg.packed and sg.world == 1
    ta = sg.submit(a, ua, want_gs_mask22=True)
    tb = sg.submit(b, ub)
    ra, rb = sg.result(ta), sg.result(tb)
    assert all(torch.equal(x, y) for x, y in zip(ra, (want_a[1], want_a[3], want_a[0], want_a[2])))
    assert torch.equal(rb[0], want_b[1]) and torch.equal(rb[1], want_b[3])
    con_rgb, dif = sg.forward_global(a, ua)
    assert torch.equal(con_rgb, want_a[1]) and torch.equal(dif, want_a[3])
    gen.close()


@pytest.mark.gpu
def test_pipelined_loop_with_and_without_the_shared_pinned_ring(golden_dir, tmp_path):
    """Round 4: the loops keep up to `gpu_inflight` batches in flight and copy pool-bound batches device -> pinned shared-memory slots the
    worker processes read in place (fsrnet._ShmPinnedRing).  Same items, same order, same PNG bytes and losses as the synchronous form
    (gpu_inflight = 0, private buffers + file copy) — for the FFHQ strips (PNG pool) and the UCB post-processing pool."""
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    w = init_weights(1)
    runs = {}
    for ucb in (False, True):
        for label, ring, depth in (("sync", False, 0), ("ring", True, 2), ("pipelined_no_ring", False, 2)):
            cfg.CHECKPOINT_DIR = str(tmp_path / ("%s_%d" % (label, ucb)))
            ds = D.Dataset(cfg, "test", ucb=True, workers=2, device_prep=0, device_batch=8)
            ds.name_list = ds.name_list[:20]
            fsr = FSRNet(cfg, weights=w)
            fsr.shm_ring, fsr.gpu_inflight = ring, depth
            fsr.log.gpu_png = False                                        # this test is about the HOST encoder pool and its ring (round 5's device writer: next test)
            fsr.log.png_workers = 0 if ucb else 2
            fsr.post_device = False
            fsr.post_workers, fsr.return_figs = (3 if ucb else 0), False
            res = fsr.test(ds, batch=8) if ucb else fsr.testFFHQ(ds, batch=8)
            ds.close()
            assert len(res) == 20 and len(fsr.log.saved) == 20
            if ring:
                assert fsr.timings.get("shm_ring_slots", 0) > 0, fsr.timings.get("shm_ring_error")      # the ring really carried the batches
            runs[(ucb, label)] = ([r[0] for r in res], [dict(r[2]) if len(r) > 2 else {} for r in res], [open(f, "rb").read() for f in fsr.log.saved])
            fsr.close()
        base = runs[(ucb, "sync")]
        for label in ("ring", "pipelined_no_ring"):
            got = runs[(ucb, label)]
            assert got[0] == base[0] and got[1] == base[1], (ucb, label)
            assert got[2] == base[2], (ucb, label)                         # PNG strips byte for byte


@pytest.mark.gpu
def test_device_png_writer_in_the_loops_gives_the_host_encoders_pixels(golden_dir, tmp_path):
    """Round 5: with the generator on a GPU the loops build the PNG FILES of a batch on the device (gpu_png.StripEncoder, stored deflate,
    checksums computed there) and the host only writes them.  Every file must decode (PIL checks chunk CRCs, zlib the Adler-32) to
    exactly the pixels the host encoder's file decodes to — synchronous and pipelined, full and ragged last batch."""
    import io
    from PIL import Image
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from blindshadowremoval_amd.gpu_png import file_bytes
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    w = init_weights(1)
    runs = {}
    for label, gpu_png, depth in (("host", False, 2), ("device_sync", True, 0), ("device", True, 3)):
        cfg.CHECKPOINT_DIR = str(tmp_path / label)
        ds = D.Dataset(cfg, "test", ucb=True, workers=2, device_prep=0, device_batch=8)
        ds.name_list = ds.name_list[:20]                                   # 8 + 8 + 4
        fsr = FSRNet(cfg, weights=w)
        assert fsr.log.gpu_png is True                                     # the default on a GPU
        fsr.log.gpu_png, fsr.gpu_inflight, fsr.return_figs = gpu_png, depth, False
        res = fsr.testFFHQ(ds, batch=8)
        ds.close()
        assert len(res) == 20 and len(fsr.log.saved) == 20
        runs[label] = [open(f, "rb").read() for f in fsr.log.saved]
        fsr.close()
    assert all(len(b) == file_bytes(256, 768) for b in runs["device"])     # the layout is a pure function of the strip's shape
    assert runs["device"] == runs["device_sync"]
    for a, b in zip(runs["device"], runs["host"]):
        A, B = np.asarray(Image.open(io.BytesIO(a)).convert("RGB")), np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
        assert A.shape == (256, 768, 3) and np.array_equal(A, B)


@pytest.mark.gpu
def test_ucb_post_processing_on_the_device_equals_the_host_form(golden_dir, tmp_path):
    """Round 5: FSRNet.test runs test_step's per-item post-processing on the GPU (ucb_post_gpu / csrc/ucb_kernels.h; masks decoded by the
    loader's workers, strips encoded to PNG files on the device).  Against the host form on the same forwards: the seven figures of
    every item bit for bit, SSIM / PSNR to 1e-4, the PNG files pixel for pixel — with and without the figures coming back, pipelined
    and synchronous, masks through the device-preparing loader and read in the loop."""
    import io
    from PIL import Image
    from blindshadowremoval_amd import dataset as D
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    w = init_weights(1)
    runs = {}
    for label, post_device, figs, depth, ds_kw in (("host", False, True, 0, dict(workers=2, device_prep=0, device_batch=8)),
                                                   ("device", True, True, 2, dict(workers=2, device_prep=0, device_batch=8)),
                                                   ("device_nofigs_sync", True, False, 0, dict(workers=2, device_prep=0, device_batch=8)),
                                                   ("device_host_rows", True, False, 2, dict(workers=0))):
        cfg.CHECKPOINT_DIR = str(tmp_path / label)
        ds = D.Dataset(cfg, "test", ucb=True, **ds_kw)
        ds.name_list = ds.name_list[:20]
        fsr = FSRNet(cfg, weights=w)
        assert fsr.post_device is True
        fsr.post_device, fsr.return_figs, fsr.gpu_inflight = post_device, figs, depth
        res = fsr.test(ds, batch=8)
        ds.close()
        assert len(res) == 20 and len(fsr.log.saved) == 20
        runs[label] = (res, [open(f, "rb").read() for f in fsr.log.saved], dict(fsr.log.losses))
        fsr.close()
    base_res, base_png, base_means = runs["host"]
    for label in ("device", "device_nofigs_sync", "device_host_rows"):
        res, png, means = runs[label]
        assert [r[0] for r in res] == [r[0] for r in base_res]
        for r, b in zip(res, base_res):
            tol = 5e-3 if label == "device_host_rows" else 1e-4          # host-prepared rows are not bit-identical inputs (1e-6)
            assert abs(r[2]["ssim"] - b[2]["ssim"]) < tol and abs(r[2]["psnr"] - b[2]["psnr"]) < 10 * tol, (label, r[0], r[2], b[2])
            if label == "device":
                assert len(r[1]) == 7
                for k in range(7):
                    assert torch.equal(r[1][k].cpu(), b[1][k].cpu()), (r[0], k)
            else:
                assert r[1] is None
        if label != "device_host_rows":                      # (host-prepared rows differ from device-prepared ones by ~1e-6: not the same forward)
            for a, b in zip(png, base_png):
                A, B = np.asarray(Image.open(io.BytesIO(a)).convert("RGB")), np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
                assert A.shape == (256, 7 * 256, 3) and np.array_equal(A, B)
        for k in ("ssim", "psnr"):
            assert abs(means[k][0] / means[k][1] - base_means[k][0] / base_means[k][1]) < (5e-2 if label == "device_host_rows" else 1e-4)


@pytest.mark.gpu
def test_data_parallel_loop_world2_on_one_gpu_over_gloo(golden_dir, tmp_path):
    """The world-2 data-parallel loop with the REAL generator: two ranks of `run_loop` share GPU 0 (`--device 0 --backend gloo`; RCCL needs a
    GPU per rank, the loop's only collective is an all_gather_object, which gloo carries).  Together they must write every item's strip
    exactly once, byte-identical to a one-process run, and rank 0's report must carry the one-process means."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["-m", "blindshadowremoval_amd.run_loop", "--loop", "ucb", "--data", os.path.join(golden_dir, "UCB", "train", "input", "*"),
              "--mask-root", os.path.join(golden_dir, "UCB_masks"), "--random-weights", "1", "--batch", "8"]
    one = subprocess.run([sys.executable] + common + ["--checkpoint-dir", str(tmp_path / "one")], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29667"]
                         + common + ["--checkpoint-dir", str(tmp_path / "two"), "--backend", "gloo", "--device", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert r2["ranks"] == 2 and r2["process_group"] == "gloo" and r2["items"] == r1["items"] and 0 < r2["items_this_rank"] < r2["items"]
    assert r2["means"] == r1["means"]                                   # re-accumulated in list order: the same float sums
    f1 = sorted(os.listdir(os.path.join(str(tmp_path / "one"), "test")))
    f2 = sorted(os.listdir(os.path.join(str(tmp_path / "two"), "test")))
    assert f1 == f2 and len(f1) == r1["items"]
    for f in f1:
        assert open(os.path.join(str(tmp_path / "one"), "test", f), "rb").read() == open(os.path.join(str(tmp_path / "two"), "test", f), "rb").read(), f
