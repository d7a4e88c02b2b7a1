"""Input-preparation counterpart (blindshadowremoval_amd/dataset.py) against the fixture produced by the REFERENCE's own
functions on sample_imgs/02165 (tests/golden/sample_02165.npz, tools/make_sample_fixture.py).  The sample image and
landmarks live under /root/reference, so the end-to-end comparison runs in the build container only."""
import os

import numpy as np
import pytest

from blindshadowremoval_amd import dataset as D

# the reference's sample input (a 256x256 PNG + 68 landmarks) is committed as a data fixture so config 1 runs on the GPU box
REF_SAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_imgs", "02165", "02165")


def test_natural_sort_and_blur():
    assert sorted(["a10", "a9", "a100", "b1"], key=D.natural_key) == ["a9", "a10", "a100", "b1"]
    a = np.zeros((9, 9), np.float32)
    a[4, 4] = 1
    b = D.gaussian_blur5(a)
    np.testing.assert_allclose(b[4, 2:7], np.array([1, 4, 6, 4, 1]) / 16 * 6 / 16, rtol=1e-12)
    assert abs(b.sum() - 1) < 1e-12
    e = np.zeros((6, 6)); e[0, 1] = 1                             # BORDER_REFLECT_101: index -1 mirrors index 1 (edge not repeated)
    assert abs(D.gaussian_blur5(e)[0, 0] - (6 / 16) * (8 / 16)) < 1e-12 and abs(D.gaussian_blur5(e)[0, 1] - (6 / 16) * (6 / 16 + 1 / 16)) < 1e-12


def test_face_model_tables():
    uv, lm_ref = D._face_model()
    assert uv.shape == (68, 3) and lm_ref.shape == (68, 2)
    assert 0 < uv.min() and uv.max() < 1 and 0 < lm_ref.min() and lm_ref.max() < 1


def test_crop_box_and_zero_extension():
    img = np.random.default_rng(0).random((100, 120, 3))
    lm = np.array([[10.0, 20.0], [90.0, 80.0]] * 34)
    crop, lmn, box = D.face_crop_and_resize(img, lm, 64)
    # centre (50,50), half-length 40*1.4 = 56 -> box x 50-56..50+56, y 50-67..50+56+56-67: leaves the image -> zero-extended
    assert box == [-6, -17, 106, 95] and crop.shape == (64, 64, 3)
    assert np.all(crop[0] == 0)                                    # rows above the image are zeros
    np.testing.assert_allclose(lmn[0], [(10 + 6) / 112, (20 + 17) / 112])


def test_row_matches_reference_prepared_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "sample_02165.npz"))
    row, box = D.build_row(REF_SAMPLE + ".png", REF_SAMPLE + ".npy")
    assert row.shape == (256, 256, 16) and row.dtype == np.float32
    np.testing.assert_array_equal(box, z["box"])
    for name, sl, tol in (("img", slice(0, 3), 1e-6), ("gt", slice(3, 6), 1e-6), ("uvm", slice(6, 9), 1e-6),
                          ("reg_in", slice(9, 12), 1e-6), ("reg_out", slice(12, 15), 1e-6), ("face", slice(15, 16), 1e-6)):
        np.testing.assert_allclose(row[..., sl], z["row"][..., sl], atol=tol, err_msg=name)


def test_dataset_feed_layout():
    from blindshadowremoval_amd.fsrnet import Config
    cfg = Config()
    cfg.DATA_DIR_TEST = [os.path.join(os.path.dirname(REF_SAMPLE), "..", "*")]
    ds = D.Dataset(cfg, "test", rows=10)
    assert [os.path.normpath(n) for n in ds.name_list] == [os.path.normpath(REF_SAMPLE + ".npy")]
    img, box, name = next(ds.feed)
    assert img.shape == (1, 10, 256, 256, 16) and box.shape == (1, 4) and name[0].endswith(b"02165.png")
    np.testing.assert_array_equal(img[0, 0], img[0, 9])          # single-image folder: every sibling is the image itself
    with pytest.raises(StopIteration):
        next(ds.feed)
    if not os.path.isdir("/root/reference/UCB/train/input"):
        return
    ucb = D.Dataset(type("C", (), {"DATA_DIR_TEST": ["/root/reference/UCB/train/input/*"], "IMG_SIZE": 256})(), "test", ucb=True)
    assert len(ucb.name_list) == 100                              # 100 items, not 99
    assert ucb._gt_path(ucb.name_list[0]).replace("/gt/", "/input/") == os.path.splitext(ucb.name_list[0])[0] + ".png"
    im, bx, nm = next(ucb.feed)
    assert im.shape == (1, 1, 256, 256, 16) and not np.array_equal(im[0, 0, ..., 0:3], im[0, 0, ..., 3:6])   # UCB gt differs from input


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f32x3"])
def test_config1_end_to_end(tmp_path, golden_dir, dtype):
    """BASELINE config 1: the sample_imgs face (a REAL image with its real, 60 %-zero uv map) through Dataset -> FSRNet.testFFHQ on
    the GPU, against the oracle — on the fp32 path and on the split-precision f32x3 path, same tolerance."""
    import torch
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from blindshadowremoval_amd.weights import init_weights
    from oracle.gsc_oracle import GeneratorOracle, test_step_ffhq
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "sample_imgs", "*")]
    w = init_weights(1)
    ds = D.Dataset(cfg, "test", rows=10)
    res = FSRNet(cfg, weights=w, dtype=dtype).testFFHQ(ds)
    assert len(res) == 1 and os.path.isfile(os.path.join(str(tmp_path), "test", "02165_02165-result.png"))
    row, _ = D.build_row(REF_SAMPLE + ".png", REF_SAMPLE + ".npy")
    ref = test_step_ffhq(GeneratorOracle(w), torch.from_numpy(row)[None])
    for a, b in zip(res[0][1], ref):
        assert float((a.cpu() - b).abs().max()) <= 1e-3


@pytest.mark.gpu
def test_config3_ucb_miniature(tmp_path, golden_dir):
    """BASELINE configs[2] (UCB, fsr.test, batch 16): the first 20 UCB items (the reference's own data files, committed as
    fixtures) through the pooled UCB loader (ground truth from the sibling gt tree) and FSRNet.test at batch 16 — one TRUE B = 16
    forward plus a ragged B = 4 one; PSNR / SSIM of the HIP outputs against the oracle's ("PSNR vs ref")."""
    import torch
    from blindshadowremoval_amd import metrics as M
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from blindshadowremoval_amd.weights import init_weights
    from oracle.gsc_oracle import GeneratorOracle
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    w = init_weights(1)
    N = 20

    def first_n(ds):                        # the fixture tree now holds all 100 UCB items (test_ucb_full_set below): this test keeps its first 20
        assert len(ds.name_list) == 100
        ds.name_list = ds.name_list[:N]
        return ds
    ds = first_n(D.Dataset(cfg, "test", ucb=True, workers=4))
    assert len(ds.name_list) == N
    fsr = FSRNet(cfg, weights=w)
    res = fsr.test(ds, batch=16, postprocess=False)
    assert len(res) == N and fsr.timings["forwards"] == 2 and fsr.timings["items"] == N       # 16 + 4
    ds2 = first_n(D.Dataset(cfg, "test", ucb=True, workers=4))
    rows = torch.from_numpy(np.concatenate([next(ds2.feed)[0][0] for _ in range(N)], axis=0))      # [20,256,256,16]
    assert not torch.equal(rows[..., 0:3], rows[..., 3:6])                                          # gt differs from the shadowed input
    img, gt, uv, reg, face = torch.split(rows, [3, 3, 3, 6, 1], dim=3)
    oracle = GeneratorOracle(w)
    ref_gs, ref_rgb, _, ref_dif = oracle(img, uv)
    hip_rgb = torch.cat([r[1][2] for r in res]).cpu()
    hip_gs = torch.cat([r[1][1] for r in res]).cpu()
    assert float((hip_rgb - ref_rgb).abs().max()) <= 1e-3 and float((hip_gs - ref_gs).abs().max()) <= 1e-3
    psnr = M.psnr(hip_rgb.clamp(0, 1), ref_rgb.clamp(0, 1))
    ssim = M.ssim(hip_rgb.clamp(0, 1), ref_rgb.clamp(0, 1))
    assert float(psnr.min()) > 80.0 and float(ssim.min()) > 0.99999
    # the metric the reference prints for UCB (train_test_GSC.py:724-725): prediction vs ground truth — finite numbers on this input
    assert torch.isfinite(M.psnr(hip_rgb.clamp(0, 1), gt)).all() and torch.isfinite(M.ssim(hip_rgb.clamp(0, 1), gt)).all()

    # the whole UCB step (generator + the reference's host post-processing, train_test_GSC.py:411-748) against the same
    # post-processing applied to the ORACLE's generator outputs: identical shadow masks, SSIM / PSNR within 1e-3
    from blindshadowremoval_amd.ucb_post import ucb_postprocess
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    fsr2 = FSRNet(cfg, weights=w)
    full = fsr2.test(first_n(D.Dataset(cfg, "test", ucb=True, workers=4)), batch=16)
    assert len(full) == N
    ds3 = first_n(D.Dataset(cfg, "test", ucb=True))
    mask_files = fsr2._ucb_masks()
    for j, (name, figs, losses) in enumerate(full):
        box = np.asarray(next(ds3.feed)[1]).reshape(-1)
        stem = os.path.basename(name).split(".")[0]                                               # e.g. 9157-022
        assert os.path.basename(mask_files[j]["face_hair"]) == "%s_%s-result.png" % (stem.split("-")[0], stem)      # masks pair up with items by order
        with np.errstate(invalid="ignore", divide="ignore"):
            ref_losses, ref_figs = ucb_postprocess(img[j].numpy(), gt[j].numpy(), ref_rgb[j].numpy(), ref_dif[j].numpy(), box, fsr2._read_masks(mask_files[j]))
        assert len(figs) == 7 and figs[4].shape == (1, 256, 256, 3)
        assert int((figs[4].numpy() != ref_figs[4]).sum()) <= 3 * 4          # at most a few pixels may sit on a threshold
        for k in ("ssim", "psnr"):
            assert np.isfinite(losses[k]) and abs(losses[k] - ref_losses[k]) < 1e-3 * max(1.0, abs(ref_losses[k]))


def test_pooled_loader_matches_serial_bit_for_bit(golden_dir):
    """Dataset(workers=N): worker processes + prefetch (counterpart of map(num_parallel_calls=AUTOTUNE).prefetch,
    /root/reference/dataset.py:63-72) must deliver the same elements in the same order as the serial path, sibling draws
    (rows > 1) included."""
    cfg = type("C", (), {"DATA_DIR_TEST": [os.path.join(golden_dir, "UCB", "train", "input", "*")], "IMG_SIZE": 256})()
    serial = D.Dataset(cfg, "test", ucb=True, rows=2, seed=5)
    pooled = D.Dataset(cfg, "test", ucb=True, rows=2, seed=5, workers=3, prefetch=4)
    assert serial.name_list == pooled.name_list and len(serial.name_list) == 100
    serial.name_list = serial.name_list[:7]
    pooled.name_list = pooled.name_list[:7]
    n = 0
    for a, b in zip(serial.feed, pooled.feed):
        assert a[0].shape == (1, 2, 256, 256, 16) and a[0].dtype == np.float32
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2][0] == b[2][0]
        n += 1
    assert n == 7
    with pytest.raises(StopIteration):
        next(pooled.feed)
    assert pooled._pool is None                              # workers are shut down when the list is exhausted
    # a failing job surfaces as an exception in the consumer, not a hang
    bad = D.Dataset(cfg, "test", ucb=True, workers=2)
    bad.name_list = [os.path.join(golden_dir, "UCB", "train", "input", "9156", "missing.npy")]
    with pytest.raises(RuntimeError, match="loader worker failed"):
        next(bad.feed)


def test_resize_linear_is_opencv_inter_linear():
    """numpy restatement of cv2.resize(INTER_LINEAR) against torch's half-pixel bilinear (both directions of scale)."""
    import torch
    rng = np.random.default_rng(0)
    for n in (300, 200, 256):
        a = rng.random((n, n, 6))
        t = torch.from_numpy(a).permute(2, 0, 1)[None]
        want = torch.nn.functional.interpolate(t, size=(256, 256), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(D.resize_linear(a, 256) - want).max() < 1e-12


def test_tsm_loaders_match_the_reference_parsers(golden_dir):
    """Dataset(config, 'test', dset='sfw' | 'sfw_video') against tests/golden/sfw_elements.npz — the output of the reference's OWN
    `parse_fn_test_sfw` / `parse_fn_test_sfw_video` (/root/reference/dataset_with_TSM.py:225-287, 289-583) on the synthetic SFW
    folder tests/golden/sfw_synth (tools/make_sfw_fixture.py; every 4th pixel + per-channel sums are stored)."""
    z = np.load(os.path.join(golden_dir, "sfw_elements.npz"))
    cfg = type("C", (), {"DATA_DIR_TEST": [os.path.join(golden_dir, "sfw_synth", "*")], "IMG_SIZE": 256})()
    ds = D.Dataset(cfg, "test", dset="sfw")
    assert [os.path.basename(n) for n in ds.name_list] == ["1_label.png", "10_label.png"]          # natural order
    for n in (1, 10):
        img, box, name = next(ds.feed)
        assert img.shape == (1, 2, 256, 256, 17) and img.dtype == np.float32 and name[0].endswith(b"/%d.png" % n)
        assert np.abs(img[0][:, ::4, ::4, :] - z["pair%d" % n]).max() < 2e-6
        assert np.abs(img[0].astype(np.float64).sum(axis=(1, 2)) - z["pair%d_sum" % n]).max() < 2e-2
        assert np.array_equal(box[0], z["pair%d_box" % n])
        # the mirror row really is the flipped frame with the mirrored landmark maps
        assert np.array_equal(img[0, 1, :, :, :7], img[0, 0, :, ::-1, :7]) and not np.array_equal(img[0, 1, ..., 7:], img[0, 0, :, ::-1, 7:])
    dv = D.Dataset(cfg, "test", dset="sfw_video", workers=2)
    for n in (1, 10):
        img, box, name = next(dv.feed)
        assert img.shape == (1, 10, 256, 256, 13)
        assert np.abs(img[0][:, ::4, ::4, :] - z["video%d" % n]).max() < 2e-6
        assert np.abs(img[0].astype(np.float64).sum(axis=(1, 2)) - z["video%d_sum" % n]).max() < 2e-2
        assert np.array_equal(box[0], z["video%d_box" % n])
    assert D.sfw_video_frames(1) == [1, 3, 5, 7, 9, 11, 13, 15, 17, 2] and D.sfw_video_frames(10) == [10, 11, 13, 15, 17, 19, 8, 6, 4, 2]
    assert D.sfw_video_frames(150)[:3] == [150, 149, 147]


@pytest.mark.gpu
def test_ucb_full_set_100_items(tmp_path, golden_dir):
    """BASELINE configs[2] on the WHOLE UCB test set: 100 items (BASELINE.json says 99; the reference tree holds 100 — SURVEY F10),
    each with its own seven masks, through the fast form of the loop a user would run — rows prepared on the device, batch 16 (six
    full forwards + a ragged one of 4), the reference's post-processing in worker processes — against the CPU oracle's generator
    outputs pushed through the same post-processing: shadow masks equal up to a few threshold pixels, SSIM / PSNR within 1e-3."""
    import torch
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from blindshadowremoval_amd.ucb_post import read_masks, ucb_postprocess
    from blindshadowremoval_amd.weights import init_weights
    from oracle.gsc_oracle import GeneratorOracle
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    w = init_weights(1)
    fsr = FSRNet(cfg, weights=w)
    fsr.post_workers = 8
    ds = D.Dataset(cfg, "test", ucb=True, workers=6, device_prep=0, device_batch=16)
    assert len(ds.name_list) == 100 and len(set(ds.name_list)) == 100
    res = fsr.test(ds, batch=16)
    assert len(res) == 100 and fsr.timings["forwards"] == 7 and len(fsr.log.saved) == 100 and all(os.path.isfile(f) for f in fsr.log.saved)
    mask_files = fsr._ucb_masks()
    assert len(mask_files) == 100
    host = D.Dataset(cfg, "test", ucb=True, workers=6)
    elems = [next(host.feed) for _ in range(100)]
    rows = torch.from_numpy(np.concatenate([e[0][0] for e in elems], axis=0))
    img, gt, uv, _, _ = torch.split(rows, [3, 3, 3, 6, 1], dim=3)
    oracle = GeneratorOracle(w)
    ref_rgb, ref_dif = [], []
    for lo in range(0, 100, 20):                                      # 20 images at a time: bounded host memory
        _, c, _, d = oracle(img[lo:lo + 20], uv[lo:lo + 20])
        ref_rgb.append(c)
        ref_dif.append(d)
    ref_rgb, ref_dif = torch.cat(ref_rgb), torch.cat(ref_dif)
    worst = {"ssim": 0.0, "psnr": 0.0, "mask_px": 0}
    sums = {"ssim": 0.0, "psnr": 0.0}
    for j, (name, figs, losses) in enumerate(res):
        stem = os.path.basename(name).split(".")[0]
        assert os.path.basename(mask_files[j]["face_hair"]) == "%s_%s-result.png" % (stem.split("-")[0], stem)
        box = np.asarray(elems[j][1]).reshape(-1)
        with np.errstate(invalid="ignore", divide="ignore"):
            ref_losses, ref_figs = ucb_postprocess(img[j].numpy(), gt[j].numpy(), ref_rgb[j].numpy(), ref_dif[j].numpy(), box, read_masks(mask_files[j]))
        worst["mask_px"] = max(worst["mask_px"], int((figs[4].numpy() != ref_figs[4]).sum()))
        for k in ("ssim", "psnr"):
            assert np.isfinite(losses[k])
            worst[k] = max(worst[k], abs(losses[k] - ref_losses[k]) / max(1.0, abs(ref_losses[k])))
            sums[k] += losses[k]
    print("UCB 100 items: mean SSIM %.4f, mean PSNR %.3f dB vs ground truth (random-init weights); worst deviation from the oracle path: %s" % (sums["ssim"] / 100, sums["psnr"] / 100, worst))
    assert worst["mask_px"] <= 3 * 8 and worst["ssim"] < 1e-3 and worst["psnr"] < 1e-3
    fsr.close()
    host.close()


def test_select_pool_png_jobs_through_shared_memory(tmp_path):
    """The worker pool the loops use (dataset._SelectPool): a batch of PNG strips parked in ONE shared-memory file, one job per strip,
    results collected by ticket — no helper threads, no megabytes through the pipes."""
    from PIL import Image
    from blindshadowremoval_amd.fsrnet import _shm_file
    rng = np.random.default_rng(3)
    strips = (rng.random((5, 32, 96, 3)) * 255).astype(np.uint8)
    shm = _shm_file("bsr_test_")
    try:
        strips.tofile(shm)
        pool = D._SelectPool(2)
        pool.warm("rows")
        tickets = [pool.submit(("png", os.path.join(str(tmp_path), "s%d.png" % j), (shm, strips.shape, j))) for j in range(5)]
        assert [pool.result(t) for t in reversed(tickets)] == [True] * 5          # any collection order
        bad = pool.submit(("png", os.path.join(str(tmp_path), "x.png"), (shm + ".missing", strips.shape, 0)))
        with pytest.raises(RuntimeError, match="worker failed"):
            pool.result(bad)
        pool.shutdown()
    finally:
        os.unlink(shm)
    for j in range(5):
        assert np.array_equal(np.asarray(Image.open(os.path.join(str(tmp_path), "s%d.png" % j))), strips[j])


def test_ucb_masks_travel_bit_packed_with_the_loader_job(golden_dir):
    """Round 5 (device post-processing): the loader's worker decodes the item's seven segmentation masks next to its image and sends them
    bit-packed; unpacked (on whatever device) they are the grey levels PIL reads, in the order of train_test_GSC.py:386-392.  CPU only:
    the host half of a device-prepared row (prep.host_part) + prep.pack_masks / unpack_masks."""
    import torch
    from PIL import Image
    from blindshadowremoval_amd import prep
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from blindshadowremoval_amd.ucb_post import MASK_DIRS
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    assert prep.MASK_ORDER == tuple(MASK_DIRS)
    ds = D.Dataset(cfg, "test", ucb=True)
    fsr = FSRNet.__new__(FSRNet)
    fsr.config = cfg
    mf = fsr._ucb_masks()
    ds.device_prep = 0                                   # only to make _jobs() emit the device form of the job; nothing touches a GPU here
    ds.ucb_mask_files = mf
    jobs = list(ds._jobs())[:3]
    assert all(len(j[1]) == 3 and j[1][0] == "<device>" and j[1][2] is mf[i] for i, j in enumerate(jobs))
    parts = [D.build_element(j) for j in jobs]
    assert all(len(p) == 6 and p[5][0] == "bits" and p[5][1].shape == (7, 256 * 256 // 8) and p[5][2] == 256 for p in parts)
    un = prep.unpack_masks([p[5] for p in parts], torch.device("cpu")).numpy()
    assert un.shape == (3, 7, 256, 256) and un.dtype == np.uint8
    for i in range(3):
        for k, key in enumerate(prep.MASK_ORDER):
            assert np.array_equal(un[i, k], np.asarray(Image.open(mf[i][key]).convert("L"), np.uint8)), (i, key)
    # a mask that is not binary travels as grey levels; a mixed batch is unpacked on the host
    grey = ("u8", np.arange(7 * 256 * 256, dtype=np.uint32).reshape(7, 256, 256).astype(np.uint8), 256)
    mixed = prep.unpack_masks([parts[0][5], grey], torch.device("cpu")).numpy()
    assert np.array_equal(mixed[0], un[0]) and np.array_equal(mixed[1], grey[1])
    # without mask files the job is the round-3 form
    ds2 = D.Dataset(cfg, "test", ucb=True)
    ds2.device_prep = 0
    assert len(next(iter(ds2._jobs()))[1]) == 2


def test_loader_ring_slot_holds_what_the_pipe_would_carry(tmp_path, golden_dir):
    """Round 5 (shared-memory transport of the device-prepared loops): prep.host_part_ring writes an item's image, ground truth, four
    triangle tables and bit-packed masks into its slot of the ring and answers with a small record; the blob layout built from such
    records (prep._layout_ex) points at exactly the bytes the pipe form (prep.host_part -> _layout) packs.  CPU only: the ring is a
    plain file here, no page-locking, no device."""
    import torch
    from blindshadowremoval_amd import prep
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    ds = D.Dataset(cfg, "test", ucb=True)
    fsr = FSRNet.__new__(FSRNet)
    fsr.config = cfg
    ds.device_prep = 0
    ds.ucb_mask_files = fsr._ucb_masks()
    jobs = list(ds._jobs())[:3]
    cap, nslots = prep.RING_CAP, 4
    path = str(tmp_path / "ring")
    with open(path, "wb") as f:
        f.truncate(nslots * cap)
    slots = [2, 3, 0]                                                      # a batch that wraps around the end of the ring
    pipe = [D.build_element(j) for j in jobs]
    # round 6: the images lie in the slot as inflated, still FILTERED scanlines; the blob gets an unfilter table (where the filtered bytes
    # are, where the RGB image goes) and the row records point at the output areas behind the cells
    recs = [D.build_element(j + ((path, s, cap, True),)) for j, s in zip(jobs, slots)]
    assert all(r[11] == (3, 3) and r[9][0] == "raw8" and r[10] <= cap for r in recs)
    ring = np.fromfile(path, np.uint8)
    total, rows_off, grid_off, pieces, head, cells, (unf_off, n_unf, mask_out) = prep._layout_ex(recs, 256, cap)
    assert n_unf == 6 + 21 and unf_off + n_unf * prep.UNFILTER_DTYPE.itemsize <= head
    assert total == head + 3 * cap + 6 * 256 * 256 * 3 + 3 * 7 * 256 * 256 and sorted(mask_out) == [0, 1, 2]
    blob = np.zeros(total, np.uint8)
    prep.pack_into(blob, pieces)
    for i, slot, base in cells:
        blob[base:base + cap] = ring[slot * cap:(slot + 1) * cap]
    unf = blob[unf_off:unf_off + n_unf * prep.UNFILTER_DTYPE.itemsize].view(prep.UNFILTER_DTYPE)
    rows = blob[rows_off:rows_off + 3 * prep.ROW_DTYPE.itemsize].view(prep.ROW_DTYPE)
    from blindshadowremoval_amd import pngio
    for k, u in enumerate(unf[:6]):
        assert (u["h"], u["w"], u["c"], u["grey_out"]) == (256, 256, 3, 0) and head + 3 * cap <= u["out_off"] and u["out_off"] + 256 * 256 * 3 <= total
        img = pngio.unfilter_host(blob[u["raw_off"]:u["raw_off"] + 256 * 769], 256, 256, 3)
        assert np.array_equal(img, pipe[k // 2][k % 2]) and u["out_off"] == rows[k // 2][("img_off", "gt_off")[k % 2]]
    for k, u in enumerate(unf[6:]):                                        # the masks: seven grey images per item, one output area per item
        i, m = divmod(k, 7)
        assert (u["h"], u["w"], u["c"], u["grey_out"]) == (256, 256, 1, 1) and u["out_off"] == mask_out[i][0] + m * 65536 and u["out_off"] + 65536 <= total
        lv = pngio.unfilter_host(blob[u["raw_off"]:u["raw_off"] + 256 * 257], 256, 256, 1)[:, :, 0]
        assert np.array_equal(np.packbits(lv != 0), pipe[i][5][1][m]) and set(np.unique(lv)) <= {0, 255}
    assert len(set(int(u["out_off"]) for u in unf)) == n_unf
    assert prep.masks_from_raw(("raw8", blob[unf[6]["raw_off"]:unf[6]["raw_off"] + 7 * 256 * 257], 256))[0] == "bits"
    bad = list(recs[0]); bad[11] = (2, 3)
    with pytest.raises(ValueError, match="channels per filtered pixel"):
        prep._layout_ex([tuple(bad)], 256, cap)
    # without the flag (round 5's form, the FFHQ loop's): decoded images in the slot, byte for byte what the pipe carries
    recs = [D.build_element(j + ((path, s, cap),)) for j, s in zip(jobs, slots)]
    assert all(r[0] == "ring" and r[1] == s and r[10] <= cap and r[11] == (0, 0) for r, s in zip(recs, slots))
    assert all(len(pickle_bytes(r)) < 1000 for r in recs) and all(len(pickle_bytes(p)) > 400000 for p in pipe)
    ring = np.fromfile(path, np.uint8)
    total, rows_off, grid_off, pieces, head, cells, (unf_off, n_unf, _) = prep._layout_ex(recs, 256, cap)
    assert [c[:2] for c in cells] == [(0, 2), (1, 3), (2, 0)] and total == head + 3 * cap and n_unf == 0
    blob = np.zeros(total, np.uint8)
    prep.pack_into(blob, pieces)
    for i, slot, base in cells:                                            # what DevicePrep.rows_ex's copies do
        blob[base:base + cap] = ring[slot * cap:(slot + 1) * cap]
    t2, r2, g2, p2 = prep._layout(pipe, 256)
    blob2 = np.zeros(t2, np.uint8)
    prep.pack_into(blob2, p2)
    rows = blob[rows_off:rows_off + 3 * prep.ROW_DTYPE.itemsize].view(prep.ROW_DTYPE)
    rows2 = blob2[r2:r2 + 3 * prep.ROW_DTYPE.itemsize].view(prep.ROW_DTYPE)
    assert np.array_equal(blob[grid_off:grid_off + 2048], blob2[g2:g2 + 2048])
    for a, b, rec, part in zip(rows, rows2, recs, pipe):
        assert (a["h"], a["w"], list(a["box"]), list(a["ntri"])) == (b["h"], b["w"], list(b["box"]), list(b["ntri"]))
        n = int(a["h"]) * int(a["w"]) * 3
        assert np.array_equal(blob[a["img_off"]:a["img_off"] + n], blob2[b["img_off"]:b["img_off"] + n])
        assert np.array_equal(blob[a["gt_off"]:a["gt_off"] + n], blob2[b["gt_off"]:b["gt_off"] + n]) and a["gt_off"] != a["img_off"]
        for m in range(4):
            k = int(a["ntri"][m]) * prep.TRI_DOUBLES * 8
            assert np.array_equal(blob[a["tri_off"][m]:a["tri_off"][m] + k], blob2[b["tri_off"][m]:b["tri_off"][m] + k])
        kind, S, moff, nbytes = rec[9]
        assert (kind, S, nbytes) == ("bits", 256, 7 * 256 * 256 // 8)
        base = [c[2] for c in cells if recs[c[0]] is rec][0]
        assert np.array_equal(blob[base + moff:base + moff + nbytes].reshape(7, -1), part[5][1])
        assert rec[8] == part[4] and np.array_equal(rec[7], part[2])
    # masks that already lie on a "device": the same grey levels as the host form
    dev = [("dev_bits", torch.from_numpy(p[5][1].copy()), 256) for p in pipe]
    assert torch.equal(prep.unpack_masks(dev, torch.device("cpu")), prep.unpack_masks([p[5] for p in pipe], torch.device("cpu")))
    assert torch.equal(prep.unpack_masks([dev[0], pipe[1][5]], torch.device("cpu")), prep.unpack_masks([p[5] for p in pipe[:2]], torch.device("cpu")))
    # an item that does not fit a slot comes back through the pipe; a record that lies about its slot is refused before any kernel sees it
    small = D.build_element(jobs[0] + ((path, 0, 1 << 16),))
    assert not prep._is_ring(small) and np.array_equal(small[0], pipe[0][0])
    bad = list(recs[0]); bad[5] = (cap - 8,) + tuple(bad[5][1:])
    with pytest.raises(ValueError, match="outside its slot"):
        prep._layout_ex([tuple(bad)], 256, cap)
    with pytest.raises(ValueError, match="ring items"):
        prep._layout(recs, 256)


def test_device_unfilter_is_decided_per_loop(golden_dir, monkeypatch):
    """Dataset.device_unfilter (round 6): None = the UCB loop with its masks reconstructs PNG scanlines on the device, the FFHQ loop does not;
    True / False force it; BSR_DEVICE_UNFILTER=0 / 1 overrides both.  The decision travels as the 4th element of the job's ring tuple."""
    from types import SimpleNamespace
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    fsr = FSRNet.__new__(FSRNet)
    fsr.config = cfg
    mf = fsr._ucb_masks()
    monkeypatch.delenv("BSR_DEVICE_UNFILTER", raising=False)

    def flag(ucb, masks, setting=None):
        ds = D.Dataset(cfg, "test", ucb=ucb)
        ds.device_prep, ds.device_unfilter = 0, setting
        ds.ucb_mask_files = mf if masks else None
        ds._ring, ds._ring_seq, ds._ring_copies = SimpleNamespace(nslots=64, cap=1 << 20, path_for_workers="/dev/null"), 0, []
        job = next(iter(ds._jobs()))
        assert len(job) == 5 and job[4][:3] == ("/dev/null", 0, 1 << 20)
        return job[4][3]
    assert flag(True, True) is True and flag(True, False) is False and flag(False, False) is False
    assert flag(False, False, True) is True and flag(True, True, False) is False
    monkeypatch.setenv("BSR_DEVICE_UNFILTER", "0")
    assert flag(True, True) is False and flag(False, False, True) is False
    monkeypatch.setenv("BSR_DEVICE_UNFILTER", "1")
    assert flag(False, False) is True and flag(True, True, False) is True


def pickle_bytes(x) -> bytes:
    import pickle
    return pickle.dumps(x, protocol=pickle.HIGHEST_PROTOCOL)


@pytest.mark.gpu
def test_loader_ring_and_pipe_give_identical_rows(golden_dir, monkeypatch):
    """The device-prepared loader through the page-locked shared-memory ring against the same loader through the workers' pipes
    (BSR_LOADER_RING=0): identical rows, boxes, names and masks, over a list long enough for every slot to be reused several times —
    with the photographs' scanlines reconstructed on the device (round 6: the UCB loop's default), in the workers, and through the pipe."""
    import torch
    from blindshadowremoval_amd import prep
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(golden_dir, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(golden_dir, "UCB_masks")
    fsr = FSRNet.__new__(FSRNet)
    fsr.config = cfg
    mf = fsr._ucb_masks()
    n = 150

    def run(ring: bool, unfilter=None):
        monkeypatch.setenv("BSR_LOADER_RING", "1" if ring else "0")
        ds = D.Dataset(cfg, "test", ucb=True, workers=3, prefetch=6, device_prep=0, device_batch=8)
        ds.device_unfilter = unfilter                      # None: the UCB loop reconstructs the PNG scanlines on the device (round 6)
        base = list(ds.name_list)
        ds.name_list = (base * 2)[:n]
        ds.ucb_mask_files = (mf * 2)[:n]
        ds.warm()
        assert (getattr(ds, "_ring", None) is not None) == ring
        if ring:
            assert ds._ring.pinned and ds._ring.nslots == 32 and ds._ring.path is None          # 6 + 3 x 8 rounded up to batches; the file's name is gone
        sums, boxes, names, msum = [], [], [], []
        for el in ds.feed:
            sums.append(el[0].double().sum(dim=(0, 1, 2, 3)).cpu())
            boxes.append(el[1]); names.append(el[2][0])
            assert el[3][0] == (("dev_bits" if unfilter is False else "dev_u8") if ring else "bits")
            msum.append(prep.unpack_masks([el[3]], torch.device("cuda", 0)).double().sum(dim=(2, 3)).cpu())
        ds.close()
        return torch.stack(sums), np.concatenate(boxes), names, torch.cat(msum)
    a, b, c = run(True), run(False), run(True, unfilter=False)
    assert len(a[2]) == n and a[2] == b[2] == c[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[0], c[0]) and torch.equal(a[3], c[3])


def test_slot_ring_file_lifecycle_and_small_tmpfs(monkeypatch):
    """prep.SlotRing on a box without a GPU: the file exists until unlink(), the mapping stays usable after it, nothing is page-locked;
    a /dev/shm with less than twice the ring's size free refuses the ring (a worker's write beyond a tmpfs limit would be a SIGBUS)."""
    import types
    from blindshadowremoval_amd import prep
    ring = prep.SlotRing(4, 4096)
    path = ring.path
    assert os.path.getsize(path) == 4 * 4096 and ring.tensor.numel() == 4 * 4096 and not ring.pinned
    view = np.memmap(path, np.uint8, "r+")
    ring.unlink()
    assert not os.path.exists(path) and ring.path is None
    view[4096:4100] = [1, 2, 3, 4]                                       # a worker's write after the name is gone
    assert ring.tensor[4096:4100].tolist() == [1, 2, 3, 4]
    del view
    ring.close()
    if os.path.isdir("/dev/shm"):
        real = os.statvfs
        monkeypatch.setattr(os, "statvfs", lambda p: types.SimpleNamespace(f_bavail=16, f_frsize=4096) if p == "/dev/shm" else real(p))
        with pytest.raises(OSError, match="loader ring needs"):
            prep.SlotRing(112)
