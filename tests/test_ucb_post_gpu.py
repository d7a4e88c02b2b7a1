"""bsr_ucb_post (csrc/ucb_kernels.h) against its host statement (blindshadowremoval_amd/ucb_post.py, itself pinned to the reference's own
test_step source by tests/golden/ucb_post_9156.npz): every figure the kernels write — the rounded masks, the gated magnitude, the
detected mask, the composite — bit for bit, SSIM / PSNR to 1e-4 (the gate is 1e-3)."""
import os

import numpy as np
import pytest

from ucb_cases import GOLDEN, cases

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(GOLDEN, "ucb_post_9156.npz"))


def _masks_u8(masks):
    from blindshadowremoval_amd.ucb_post_gpu import MASK_ORDER
    return np.stack([np.rint(masks[k][:, :, 0] * 255.0).astype(np.uint8) for k in MASK_ORDER], axis=0)


def _run(batch, want_figs=True):
    import torch
    from blindshadowremoval_amd.ucb_post_gpu import UcbPostDevice
    rows10 = torch.from_numpy(np.stack([np.concatenate([row[..., 0:3], row[..., 3:6], con, dif], axis=2) for _, row, _, _, con, dif in batch])).cuda()
    masks = torch.from_numpy(np.stack([_masks_u8(m) for _, _, _, m, _, _ in batch])).cuda()
    boxes = torch.from_numpy(np.stack([np.asarray(b, np.float32).reshape(4) for _, _, b, _, _, _ in batch])).cuda()
    post = UcbPostDevice(0)
    for _ in range(2):                                      # twice: nothing may depend on what the scratch held before
        losses, strips, figs, status = post.run(rows10, masks, boxes, want_figs=want_figs)
    torch.cuda.synchronize()
    return losses.cpu().numpy(), strips.cpu().numpy(), (figs.cpu().numpy() if figs is not None else None), status.cpu().numpy()


def test_device_post_processing_matches_the_host_statement_and_the_reference_fixture():
    from blindshadowremoval_amd.ucb_post import ucb_postprocess
    batch = list(cases())
    losses, strips, figs, status = _run(batch)
    assert (status == 0).all()
    for j, (key, row, box, masks, con, dif) in enumerate(batch):
        with np.errstate(invalid="ignore", divide="ignore"):
            l_ref, f_ref = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, masks)
        np.testing.assert_array_equal(figs[j, 4, :, :, 0].astype(np.uint8), FIX[key + "_detected"])       # the reference's own decisions
        for k in range(7):                                   # every figure, bit for bit (resizes, gates, composite)
            np.testing.assert_array_equal(figs[j, k], f_ref[k][0], err_msg="%s fig %d" % (key, k))
        cols = [np.clip(f[0], 0.0, 1.0) * np.float32(255) for f in f_ref]
        np.testing.assert_array_equal(strips[j], np.rint(np.concatenate(cols, axis=1)).astype(np.uint8))
        assert abs(float(losses[j, 0]) - l_ref["ssim"]) < 1e-4 and abs(float(losses[j, 1]) - l_ref["psnr"]) < 1e-4, (key, losses[j], l_ref)
        assert abs(float(losses[j, 0]) - float(FIX[key + "_ssim"])) < 1e-4 and abs(float(losses[j, 1]) - float(FIX[key + "_psnr"])) < 1e-3


def test_other_crop_sizes_and_rule_branches():
    """The same items under crop boxes of other sizes (odd scales put the rounded masks on .5 ties; size == S is the identity resize)
    and with magnitudes pushed through the other threshold branches: decisions and figures still bit-identical to the host statement."""
    from blindshadowremoval_amd.ucb_post import ucb_postprocess
    base = list(cases())
    batch = []
    for i, size in enumerate((256, 255, 192, 200, 171, 129, 128, 233, 250, 96)):
        key, row, box, masks, con, dif = base[i % len(base)]
        b = np.asarray(box, np.float32).reshape(4).copy()
        b[3] = b[1] + size
        batch.append(("%s_s%d" % (key, size), row, b, masks, con, (dif * np.float32(1 + 0.37 * i)).astype(np.float32)))
    losses, strips, figs, status = _run(batch)
    for j, (key, row, box, masks, con, dif) in enumerate(batch):
        try:
            with np.errstate(invalid="ignore", divide="ignore"):
                l_ref, f_ref = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, masks)
        except ValueError:                                  # an emptied mask: the reference raises, the device reports it
            assert status[j] == 1, key
            continue
        assert status[j] == 0, key
        for k in range(7):
            np.testing.assert_array_equal(figs[j, k], f_ref[k][0], err_msg="%s fig %d" % (key, k))
        if np.isfinite(l_ref["psnr"]):
            assert abs(float(losses[j, 0]) - l_ref["ssim"]) < 1e-4 and abs(float(losses[j, 1]) - l_ref["psnr"]) < 1e-3, (key, losses[j], l_ref)


@pytest.mark.parametrize("step", [2, 4])
def test_smaller_images(step):
    """S = 128 and 64 (bsr_ucb_post takes 32 .. 256): the same items subsampled — the stage chain's grids, the run labelling (at S = 64 a
    wave spans a whole image row), leaf counts and the sum tree are functions of S; still bit-identical to the host statement."""
    from blindshadowremoval_amd.ucb_post import ucb_postprocess
    batch = []
    for i, (key, row, box, masks, con, dif) in enumerate(cases()):
        S = row.shape[0] // step
        b = np.asarray(box, np.float32).reshape(4).copy()
        b[3] = b[1] + (S if i % 2 == 0 else S - 1 - 3 * i)
        sub = lambda a: np.ascontiguousarray(a[::step, ::step])
        batch.append(("%s_S%d" % (key, S), sub(row), b, {k: sub(v) for k, v in masks.items()}, sub(con), sub(dif)))
    losses, strips, figs, status = _run(batch)
    done = 0
    for j, (key, row, box, masks, con, dif) in enumerate(batch):
        try:
            with np.errstate(invalid="ignore", divide="ignore"):
                l_ref, f_ref = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, masks)
        except ValueError:
            assert status[j] == 1, key
            continue
        assert status[j] == 0, key
        done += 1
        for k in range(7):
            np.testing.assert_array_equal(figs[j, k], f_ref[k][0], err_msg="%s fig %d" % (key, k))
        if np.isfinite(l_ref["psnr"]):
            assert abs(float(losses[j, 0]) - l_ref["ssim"]) < 1e-4 and abs(float(losses[j, 1]) - l_ref["psnr"]) < 1e-3, (key, losses[j], l_ref)
    assert done >= len(batch) // 2


def test_empty_mask_and_bad_box_are_reported():
    import torch
    from blindshadowremoval_amd.ucb_post_gpu import UcbPostDevice, raise_for_status
    key, row, box, masks, con, dif = next(iter(cases()))
    rows10 = torch.from_numpy(np.concatenate([row[..., 0:3], row[..., 3:6], con, dif], axis=2)[None].repeat(3, 0)).cuda()
    m = _masks_u8(masks)[None].repeat(3, 0)
    m[1, 3] = 0                                             # item 1: no nose
    boxes = np.asarray(box, np.float32).reshape(1, 4).repeat(3, 0)
    boxes[2, 3] = boxes[2, 1] + 300                         # item 2: a 300-pixel box in a 256-pixel image
    losses, strips, figs, status = UcbPostDevice(0).run(rows10, torch.from_numpy(m).cuda(), torch.from_numpy(boxes).cuda())
    st = status.cpu().numpy()
    assert list(st) == [0, 1, 2] and np.isfinite(losses[0].cpu().numpy()).all() and np.isnan(losses[1:].cpu().numpy()).all()
    with pytest.raises(ValueError, match="item b"):
        raise_for_status(st, ["a", "b", "c"])
    with pytest.raises(TypeError):
        UcbPostDevice(0).run(rows10.cpu(), torch.from_numpy(m).cuda(), torch.from_numpy(boxes).cuda())
