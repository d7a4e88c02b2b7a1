"""UCB post-processing of `FSRNet.test_step` (/root/reference/train_test_GSC.py:411-748) — SURVEY §8f row N2.

Host-side, per image: the generator outputs of row 0 are resized to the crop-box size and zero-padded back to 256x256, the
predicted shadow magnitude is gated by hand-tuned per-region thresholds (hair, forehead, mustache, mouth, "mouth and
below", left eyebrow), the largest 4-connected components survive, a nose rule may clear a block, and the prediction is
composited over the input inside the detected shadow only; SSIM / PSNR against the ground truth are the reported losses.
This is data-dependent control flow on 256x256 arrays (numpy + scipy.ndimage.label), not part of the GPU hot path.

The numbers (0.018, 0.02, 0.252 ...) are the reference's; line references are given per step.  Pinned by
tests/golden/ucb_post_9156.npz, produced by executing the reference's own `test_step` source
(tools/make_ucb_post_fixture.py).
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from .metrics import psnr as _psnr, ssim as _ssim

MASK_DIRS = {          # train_test_GSC.py:386-392, relative to Config.UCB_MASK_ROOT
    "face_hair": "UCB_input_images_face_masks_cropped_and_padded_with_hair",
    "face": "UCB_input_images_face_masks_cropped_and_padded",
    "mouth": "UCB_input_images_mouth_masks_cropped_and_padded",
    "nose": "UCB_input_images_nose_masks_cropped_and_padded",
    "eyebrow": "UCB_input_images_eyebrow_masks_cropped_and_padded",
    "eye": "UCB_input_images_eye_masks_cropped_and_padded",
    "glasses": "UCB_input_images_glasses_masks_cropped_and_padded",
}


def resize_weights(out_size: int, in_size: int):
    """Per output index: (lower, upper, lerp) of TensorFlow's bilinear kernel with half-pixel centres — compute_interpolation_weights
    + HalfPixelScaler of tensorflow/core/kernels/image/resize_bilinear_op.cc / image_resizer_state.h (TF 2.3), float32 throughout:
    in = (i + 0.5f) * (in_size / out_size) - 0.5f;  lower = max(floor(in), 0);  upper = min(ceil(in), in_size - 1);  lerp = in - floor(in)."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lower = np.maximum(fl.astype(np.int64), 0)
    upper = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
    return lower, upper, (src - fl).astype(np.float32)


def resize_bilinear(x: np.ndarray, size: int) -> np.ndarray:
    """tf.image.resize(x, [size, size]) for an [H,W,C] array (the reference's calls: train_test_GSC.py:437-471) -> float32.
    Bilinear, half-pixel centres, no antialiasing, in the ARITHMETIC of TensorFlow's CPU kernel (compute_lerp of
    resize_bilinear_op.cc): top = tl + (tr - tl) * x_lerp; bottom = bl + (br - bl) * x_lerp; out = top + (bottom - top) * y_lerp,
    every operation a rounded float32 operation, no fused multiply-add.  Round 5: plain numpy instead of torch's F.interpolate,
    whose rounding depended on the thread count and the channel count (different vector kernels) — the rounded masks
    (round(resize(mask)), :439-471) sit on exact .5 ties for some crop sizes, so the operation order decides pixels; this form is
    deterministic, and the device kernel (csrc/ucb_kernels.h) executes the same operations, bit for bit."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h, w = x.shape[0], x.shape[1]
    ylo, yhi, yl = resize_weights(size, h)
    xlo, xhi, xl = resize_weights(size, w)
    top_rows, bot_rows = x[ylo], x[yhi]                                      # [size, W, C]
    xl_ = xl[None, :, None]
    tl, tr = top_rows[:, xlo], top_rows[:, xhi]
    top = tl + (tr - tl) * xl_
    bl, br = bot_rows[:, xlo], bot_rows[:, xhi]
    bottom = bl + (br - bl) * xl_
    return (top + (bottom - top) * yl[:, None, None]).astype(np.float32)


def _pad(x: np.ndarray, size: int, full: int) -> np.ndarray:
    return np.pad(x, [[0, full - size], [0, full - size], [0, 0]])


def _bbox(mask2d: np.ndarray) -> Tuple[int, int, int, int]:
    rows, cols = np.where(mask2d == 1)
    return int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max())


def ucb_postprocess(img0: np.ndarray, gt0: np.ndarray, con_rgb0: np.ndarray, mask_pred0: np.ndarray, box: np.ndarray,
                    masks: Dict[str, np.ndarray]) -> Tuple[Dict[str, float], List[np.ndarray]]:
    """img0 / gt0 / con_rgb0: [S,S,3]; mask_pred0: [S,S,1] (the generator's `dif`); box: [4]; masks: the seven [S,S,3] (or, from read_masks(grey=True), [S,S,1])
    {0,1} maps of MASK_DIRS.  Returns ({'ssim','psnr'}, figs) with figs as in train_test_GSC.py:744: input, composite,
    2 x gated magnitude, ground truth, detected shadow mask, full prediction, nose image — each [1,S,S,3] float32."""
    full = img0.shape[0]
    box = np.asarray(box).reshape(4)
    size = int(box[3] - box[1])                                                            # :417-418
    rs = lambda a: resize_bilinear(a, size)
    gt_sc = _pad(rs(gt0), size, full)                                                      # :437,454
    pred = rs(con_rgb0)                                                                    # :438
    # :439-471: the seven masks, resized and rounded (round-half-even, as tf.round).  They come from cv2.imread of grey PNGs — three
    # IDENTICAL channels — so one channel of each is resized (all seven in one call) and repeated; any other input takes the plain path
    keys = list(masks)
    grey = all(v.shape[2] == 1 for v in masks.values())                                    # read_masks(grey=True): one channel, known to stand for three equal ones
    if grey or all(v.shape[2] == 3 and np.array_equal(v[..., 0], v[..., 1]) and np.array_equal(v[..., 0], v[..., 2]) for v in masks.values()):
        mr = np.round(rs(np.stack([masks[k][:, :, 0] for k in keys], axis=2).astype(np.float32)))
        m = {k: _pad(np.repeat(mr[..., i:i + 1], 3, axis=2), size, full).astype(np.float32) for i, k in enumerate(keys)}
    else:
        m = {k: _pad(np.round(rs(v)), size, full).astype(np.float32) for k, v in masks.items()}
    face_hair, face, mouth, nose, brow = m["face_hair"], m["face"], m["mouth"], m["nose"], m["eyebrow"]
    tmp = _pad(rs(img0), size, full)                                                       # :456-458
    mp = _pad(rs(mask_pred0), size, full) * face_hair                                      # :473-477, now 3 channels

    # mustache / mouth false positives (:479-499)
    n_top, n_bot, n_left, n_right = _bbox(nose[:, :, 0])
    mid_nose_height, lower_nose, mid_nose_width = (n_bot + n_top) / 2.0, n_bot, (n_right + n_left) / 2.0
    upper_mouth, lower_mouth, left_mouth, right_mouth = _bbox(mouth[:, :, 0])
    region = np.zeros((full, full, 3))
    region[int(mid_nose_height):int(upper_mouth), int(left_mouth):int(right_mouth)] = 1
    mp = mp * np.logical_not(np.logical_and(mp < 0.018, region == 1)).astype(np.float32)
    region = np.zeros((full, full, 3))
    region[int(upper_mouth):int(lower_mouth), int(left_mouth):int(right_mouth)] = 1
    mp = mp * np.logical_not(np.logical_and(mp < 0.02, region == 1)).astype(np.float32)

    hair = (face_hair - face).astype(np.float32)                                           # :501
    intensity = np.repeat(np.mean(tmp, axis=2, dtype=np.float32).reshape(full, full, 1), 3, axis=2)   # :525-526
    threshold = np.zeros((full, full, 3)) + 0.01                                           # :523-524
    threshold[hair > 0] = 0.02                                                             # :528
    threshold[np.logical_and(hair > 0, intensity < 0.13)] = 0.004                          # :529

    if np.sum(brow) > 30:                                                                  # forehead (:533-544)
        forehead = face.copy()
        upper_brow = int(np.where(brow[:, :, 0] == 1)[0].min())
        forehead[upper_brow:full, :, :] = 0
        f_top, _, f_left, f_right = _bbox(forehead[:, :, 0])
        fm = np.zeros((full, full, 3))
        fm[int(f_top + 20):int(upper_brow - 40), int(f_left + 40):int(f_right - 40)] = 1
        threshold[np.logical_and(fm > 0, intensity < 0.4)] = -0.001

    below = np.zeros((full, full, 3), np.float32)                                          # mouth and below (:547-564)
    below[int(upper_mouth):full, :, :] = 1.0
    roi = below * face
    shadowed = (mp > 0.01).astype(np.float32)
    frac = np.sum(shadowed * roi, dtype=np.float32) / np.sum(roi, dtype=np.float32)
    if 0.252 < frac < 0.268:
        threshold[roi > 0] = 1.0
    mean_below = np.sum(np.mean(roi * tmp * shadowed, 2)) / np.sum(roi[:, :, 0] * shadowed[:, :, 0])
    if 0.3 < frac < 0.31 and mean_below > 0.358:
        threshold[roi > 0] = 1.0
    if 0.295 < frac < 0.3 and mean_below > 0.22:
        threshold[roi > 0] = 1.0
    if np.sum(brow) > 0:                                                                   # left eyebrow on the face edge (:565-579)
        left_brow = int(np.where(brow[:, :, 0] == 1)[1].min())
        _, _, left_face, right_face = _bbox(face[:, :, 0])
        if left_brow - left_face == 0:
            left_mask = np.zeros((full, full, 3))
            left_mask[:, 0:int(left_face * 0.8 + right_face * 0.2), :] = 1.0
            threshold[np.logical_and(brow * left_mask > 0, intensity > 0.1)] = 1.0

    detected = (mp > threshold.astype(np.float32)).astype(np.uint8)                        # :586-590

    # keep the big 4-connected components that are not hair-only (:594-615)
    from scipy import ndimage
    labels, ncomp = ndimage.label(detected[:, :, 0], structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    sizes = np.bincount(labels.reshape(-1), minlength=ncomp + 1)[1:]
    keep = np.zeros((full, full, 1))
    if ncomp:
        min_size = 0.45 * np.max(sizes)
        for i in range(ncomp):
            comp = labels == i + 1
            if sizes[i] >= min_size and np.sum(hair[:, :, 0] * comp) / sizes[i] < 0.8:
                keep[comp] = 1

    # nose rule (:650-666)
    shadow_image = keep * np.mean(tmp, 2).reshape(full, full, 1)
    mean_intensity = np.sum(shadow_image) / np.sum(keep)
    frac_nose = np.sum((nose[:, :, 0:1] * shadow_image) > 0) / np.sum(nose[:, :, 0])
    if (0.15 < frac_nose < 0.25) or (0.30 < frac_nose < 0.31) or (0.34 < frac_nose < 0.35):
        reach = 5 if mean_intensity < 0.15 else 65
        keep[int(mid_nose_height):int(lower_nose + reach), int(mid_nose_width - 35):int(mid_nose_width + 35)] = 0

    detected3 = np.concatenate((keep, keep, keep), axis=2).astype(np.float32)              # :690-693
    full_pred = _pad(pred, size, full)                                                     # :711-712
    out = np.clip(full_pred * detected3 + tmp * (1 - detected3), 0, 1).astype(np.float32)  # :714,722
    g, o = torch.from_numpy(gt_sc)[None], torch.from_numpy(out)[None]
    losses = {"ssim": float(_ssim(g, o).sum()), "psnr": float(_psnr(g, o).sum())}          # :724-725
    figs = [tmp, out, mp * 2, gt_sc, detected3, full_pred, nose * tmp]                     # :744
    return losses, [np.asarray(f, np.float32).reshape(1, full, full, 3) for f in figs]


def read_masks(paths: Dict[str, str], grey: bool = False) -> Dict[str, np.ndarray]:
    """The seven mask images of one item: cv2.imread(...)/255.0, three equal channels (train_test_GSC.py:386-393).  grey=True keeps
    ONE channel ([S,S,1]), which ucb_postprocess expands itself after the resize — same values, a third of the copies."""
    from PIL import Image
    out = {}
    for k, path in paths.items():
        a = np.asarray(Image.open(path).convert("L"), np.float64) / 255.0
        out[k] = a[:, :, None] if grey else np.repeat(a[:, :, None], 3, axis=2)
    return out


def run_post_job(job: dict):
    """One item of FSRNet.test's post-processing in a worker process (fsrnet.FSRNet(post_workers=N)): reads the item's masks,
    runs ucb_postprocess, optionally writes the PNG strip of the seven figures itself.  -> (losses, figs | None)."""
    if "shm" in job:           # the batch's [B,S,S,10] float32 block (im3 | gt3 | con_rgb3 | dif1) parked in shared memory by the parent
        shape, idx = job["shape"], job["index"]
        n = int(np.prod(shape[1:]))
        a = np.fromfile(job["shm"], np.float32, count=n, offset=idx * n * 4).reshape(shape[1:])
        job = dict(job, im=a[..., 0:3], gt=a[..., 3:6], con=a[..., 6:9], mp=a[..., 9:10])
    with np.errstate(invalid="ignore", divide="ignore"):
        losses, figs = ucb_postprocess(job["im"], job["gt"], job["con"], job["mp"], job["box"], read_masks(job["masks"], grey=True))
    if job.get("png"):
        from .pngio import write_png
        cols = [np.clip(f[0], 0.0, 1.0) * np.float32(255) for f in figs]
        strip = np.rint(np.concatenate(cols, axis=1)).astype(np.uint8)           # = fsrnet.Logging.get_imgs
        write_png(job["png"], strip)
    return losses, (figs if job.get("return_figs", True) else None)
