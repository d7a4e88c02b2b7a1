"""Device-side input preparation (blindshadowremoval_amd/prep.py + csrc/prep_kernels.h) against the fixtures that pin the host
path: tests/golden/sample_02165.npz — produced by the REFERENCE's own `face_crop_and_resize` / `generate_face_region` /
`generate_uv_map` / `generate_offset_map` (tools/make_sample_fixture.py) — and the host `build_row` (itself pinned to that fixture) on
the UCB items.  CPU: the host half (crop box, triangle tables) through a numpy emulation of the kernel's arithmetic.  GPU: the kernel."""
import os

import numpy as np
import pytest

from blindshadowremoval_amd import dataset as D
from blindshadowremoval_amd import prep


def emulate(part, size):
    """numpy statement of csrc/prep_kernels.h for one row (test infrastructure)."""
    img, gt, box, tabs, _ = part
    out = np.zeros((size, size, 16), np.float64)
    n = int(box[2] - box[0])

    def crop(im):
        c = np.zeros((n, n, 3), np.float64)
        ys, xs = np.arange(n) + box[1], np.arange(n) + box[0]
        oky, okx = (ys >= 0) & (ys < im.shape[0]), (xs >= 0) & (xs < im.shape[1])
        c[np.ix_(oky, okx)] = im[np.ix_(ys[oky], xs[okx])].astype(np.float64) / 255.0
        return c
    out[..., 0:3] = D.resize_linear(crop(img), size)
    out[..., 3:6] = D.resize_linear(crop(gt if gt is not None else img), size)
    lin = np.linspace(0, 1, size)
    px, py = np.meshgrid(lin, lin)
    hull = None
    for m, t in enumerate(tabs):
        l = np.stack([(t[:, 3 * i][:, None, None] * px + t[:, 3 * i + 1][:, None, None] * py) + t[:, 3 * i + 2][:, None, None] for i in range(3)], 0).min(0)
        best = l.argmax(0)
        inside = np.take_along_axis(l, best[None], 0)[0] >= -1e-12
        z = [(t[best, 9 + 3 * k] * px + t[best, 10 + 3 * k] * py) + t[best, 11 + 3 * k] for k in range(3)]
        if m == 0:
            for k in range(3):
                out[..., 6 + k] = np.where(inside, z[k], 0.0)
        elif m < 3:
            my, mx = np.where(inside, z[0], np.nan), np.where(inside, z[1], np.nan)
            out[..., 6 + 3 * m], out[..., 7 + 3 * m], out[..., 8 + 3 * m] = my, mx, mx * 0
        else:
            hull = (inside & (z[0] > 0)).astype(np.float32)
    out[..., 15] = D.gaussian_blur5(hull)
    return out.astype(np.float32)


def _sample_part(golden_dir, size=256):
    base = os.path.join(golden_dir, "sample_imgs", "02165", "02165")
    return prep.host_part((base + ".npy", None, size))


def test_host_half_and_kernel_arithmetic_reproduce_the_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "sample_02165.npz"))
    part = _sample_part(golden_dir)
    assert np.array_equal(part[2].astype(np.float32), z["box"].reshape(-1)[:4])
    row = emulate(part, 256)
    err = np.abs(row - z["row"]).max(axis=(0, 1))
    assert err.max() <= 1e-6, err
    assert [t.shape[1] for t in part[3]] == [prep.TRI_DOUBLES] * 4 and all(0 < t.shape[0] <= prep.MAX_TRI for t in part[3])


def test_blob_layout(golden_dir):
    part = _sample_part(golden_dir)
    blob, rows_off, grid_off = prep.pack_batch([part, part], 256)
    rows = np.frombuffer(blob, prep.ROW_DTYPE, count=2, offset=rows_off)
    assert rows_off % 8 == 0 and grid_off % 8 == 0 and all(int(o) % 8 == 0 for r in rows for o in r["tri_off"])
    assert np.array_equal(np.frombuffer(blob, "<f8", count=256, offset=grid_off), np.linspace(0, 1, 256))
    r = rows[1]
    img = np.frombuffer(blob, np.uint8, count=int(r["h"]) * int(r["w"]) * 3, offset=int(r["img_off"])).reshape(int(r["h"]), int(r["w"]), 3)
    assert np.array_equal(img, part[0]) and r["gt_off"] == r["img_off"]
    t0 = np.frombuffer(blob, "<f8", count=int(r["ntri"][0]) * 18, offset=int(r["tri_off"][0])).reshape(-1, 18)
    assert np.array_equal(t0, part[3][0])


@pytest.mark.gpu
def test_device_rows_match_the_reference_fixture_and_the_host_path(golden_dir):
    import glob
    import torch
    z = np.load(os.path.join(golden_dir, "sample_02165.npz"))
    dp = prep.DevicePrep(0, 256)
    parts = [_sample_part(golden_dir)]
    ucb = sorted(glob.glob(os.path.join(golden_dir, "UCB", "train", "input", "*", "*.npy")), key=D.natural_key)
    assert len(ucb) == 100
    ucb = ucb[::4]                                    # every fourth item: 25 of the 100, all subjects
    host_rows = [z["row"]]
    for lm_path in ucb:
        parts_gt = lm_path.replace("\\", "/").split("/")
        gt = os.path.splitext("/".join(parts_gt[:-3] + ["gt"] + parts_gt[-2:]))[0] + ".png"
        parts.append(prep.host_part((lm_path, gt, 256)))
        host_rows.append(D.build_row(os.path.splitext(lm_path)[0] + ".png", lm_path, gt, 256)[0])
    out, boxes = dp.rows(parts)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == (len(parts), 256, 256, 16) and np.array_equal(boxes[0], z["box"].reshape(-1)[:4])
    for i, ref in enumerate(host_rows):
        err = np.abs(out[i] - ref).max(axis=(0, 1))
        assert np.isfinite(out[i]).all() and err.max() <= 1e-6, (i, err)


def _border_case(tmp_path):
    """A synthetic image whose landmarks hug the top-left corner: the crop box leaves the image on two sides (zero extension,
    utils.py:414-425)."""
    from PIL import Image
    rng = np.random.default_rng(5)
    img = (rng.random((200, 240, 3)) * 255).astype(np.uint8)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_02165.npz"))
    lm0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_imgs", "02165", "02165.npy")).astype(np.float32)
    lm = (lm0 - lm0.min(0)) * 0.55 + np.float32(4.0)            # face squeezed into the corner: box starts at negative coordinates
    base = os.path.join(str(tmp_path), "corner")
    Image.fromarray(img).save(base + ".png")
    np.save(base + ".npy", lm)
    return base


def test_zero_extended_crop_matches_the_host_path(tmp_path):
    base = _border_case(tmp_path)
    part = prep.host_part((base + ".npy", None, 256))
    assert part[2][0] < 0 and part[2][1] < 0                      # the box really leaves the image
    row_host, box_host = D.build_row(base + ".png", base + ".npy", None, 256)
    assert np.array_equal(part[2].astype(np.float32), box_host)
    err = np.abs(emulate(part, 256) - row_host).max(axis=(0, 1))
    assert err.max() <= 1e-6, err


@pytest.mark.gpu
def test_device_zero_extended_crop(tmp_path):
    import torch
    base = _border_case(tmp_path)
    row_host, _ = D.build_row(base + ".png", base + ".npy", None, 256)
    out, _ = prep.DevicePrep(0, 256).rows([prep.host_part((base + ".npy", None, 256))])
    torch.cuda.synchronize()
    err = np.abs(out[0].cpu().numpy() - row_host).max(axis=(0, 1))
    assert err.max() <= 1e-6, err


def _culled_winner(t, lin, by, bx):
    """numpy statement of prep_rows_kernel's round-5 block culling for one 16x16-pixel block of one mesh: -> (index of the winning
    triangle per pixel, -1 when every triangle was culled; its min barycentric; how many triangles the block walks)."""
    gx0, gx1, gy0, gy1 = lin[bx * 16], lin[bx * 16 + 15], lin[by * 16], lin[by * 16 + 15]
    keep = np.ones(t.shape[0], bool)
    for i in range(3):
        mx = np.maximum(t[:, 3 * i] * gx0, t[:, 3 * i] * gx1)
        my = np.maximum(t[:, 3 * i + 1] * gy0, t[:, 3 * i + 1] * gy1)
        keep &= ((mx + my) + t[:, 3 * i + 2] >= -1.0e-9)
    idx = np.nonzero(keep)[0]                                   # survivors in their original order
    px, py = np.meshgrid(lin[bx * 16:bx * 16 + 16], lin[by * 16:by * 16 + 16])
    if idx.size == 0:
        return np.full(px.shape, -1), np.full(px.shape, -1.0e300), 0
    s = t[idx]
    l = np.stack([(s[:, 3 * i][:, None, None] * px + s[:, 3 * i + 1][:, None, None] * py) + s[:, 3 * i + 2][:, None, None] for i in range(3)], 0).min(0)
    best = l.argmax(0)                                          # first of equal maxima, as the kernel's strict `>`
    return idx[best], np.take_along_axis(l, best[None], 0)[0], int(idx.size)


def test_block_culling_keeps_every_winner(golden_dir, tmp_path):
    """Round 5: a workgroup of prep_rows_kernel walks only the triangles whose edge functions can reach -1e-9 somewhere on its 16x16-pixel
    block.  For every block of every mesh of the sample and of the corner case: wherever the full search finds a triangle `inside`
    (>= -1e-12), the culled search finds THE SAME triangle; where it finds none, neither does the culled one — the two cases the
    kernel's outputs depend on.  And the point of it: a block walks a small fraction of the mesh."""
    size = 256
    lin = np.linspace(0, 1, size)
    px, py = np.meshgrid(lin, lin)
    corner = prep.host_part((_border_case(tmp_path) + ".npy", None, size))
    walked, total = 0, 0
    for part in (_sample_part(golden_dir), corner):
        for t in part[3]:
            l = np.stack([(t[:, 3 * i][:, None, None] * px + t[:, 3 * i + 1][:, None, None] * py) + t[:, 3 * i + 2][:, None, None] for i in range(3)], 0).min(0)
            full_best = l.argmax(0)
            full_l = np.take_along_axis(l, full_best[None], 0)[0]
            for by in range(size // 16):
                for bx in range(size // 16):
                    cb, cl, n = _culled_winner(t, lin, by, bx)
                    sl = (slice(by * 16, by * 16 + 16), slice(bx * 16, bx * 16 + 16))
                    inside = full_l[sl] >= -1e-12
                    assert np.array_equal(cb[inside], full_best[sl][inside])
                    assert np.array_equal(cl >= -1e-12, inside)
                    walked += n
                    total += t.shape[0]
    assert walked < 0.2 * total, (walked, total)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [64, 128])
def test_device_rows_at_other_sizes_match_the_host_path(golden_dir, size):
    """The block mapping of the kernel (16x16 pixels per workgroup, S / 16 blocks per row) at other S than the reference's 256."""
    import torch
    base = os.path.join(golden_dir, "sample_imgs", "02165", "02165")
    row_host, _ = D.build_row(base + ".png", base + ".npy", None, size)
    out, _ = prep.DevicePrep(0, size).rows([prep.host_part((base + ".npy", None, size))])
    torch.cuda.synchronize()
    got = out[0].cpu().numpy()
    assert got.shape == (size, size, 16)
    err = np.abs(got - row_host).max(axis=(0, 1))
    assert err.max() <= 1e-6, err
