"""bsr_png_unfilter (csrc/prep_kernels.h: PNG scanline reconstruction on the device, round 6) against the host reconstruction
(pngio.unfilter_host = hostsrc/png_unfilter.c) — on the reference's own photographs and on synthetic scanlines that use every filter
type, 1 / 3 / 4 channels, odd widths, up to the kernel's 256 rows.  Bit-exact: it is byte arithmetic."""
import glob
import os

import numpy as np
import pytest

from ucb_cases import GOLDEN

pytestmark = pytest.mark.gpu


def _filter_rows(img: np.ndarray, fts) -> np.ndarray:
    """uint8 [h,w,c] + one filter type per row -> the filtered scanlines a PNG encoder would deflate (RFC 2083 section 6)."""
    h, w, c = img.shape
    x = img.reshape(h, w * c).astype(np.int32)
    out = np.zeros((h, 1 + w * c), np.uint8)
    zero = np.zeros(w * c, np.int32)
    for y in range(h):
        cur, up = x[y], (x[y - 1] if y else zero)
        a = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), up[:-c]])
        ft = int(fts[y])
        if ft == 0:
            pred = zero
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (a + up) >> 1
        else:
            p = a + up - ul
            pa, pb, pc = np.abs(p - a), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, ul))
        out[y, 0] = ft
        out[y, 1:] = (cur - pred) & 255
    return out


def _run(items, grey=False):
    """items: [(raw uint8 [h, 1 + w c], h, w, c)] -> the RGB8 images the kernel wrote (grey: c = 1 images as one byte per pixel)."""
    import torch
    from blindshadowremoval_amd import _lib, prep
    lib = _lib.load()
    tab = np.zeros(len(items), prep.UNFILTER_DTYPE)
    off = ((tab.nbytes + 7) & ~7) + 16                     # (16 readable bytes in front of the first image; the output areas follow the last)
    for k, (raw, h, w, c) in enumerate(items):
        tab[k] = (off, 0, h, w, c, 1 if grey else 0)
        off = (off + raw.size + 7) & ~7
    for k, (raw, h, w, c) in enumerate(items):
        tab[k]["out_off"] = off
        off = (off + h * w * (1 if grey else 3) + 7) & ~7
    blob = np.full(off, 0xA5, np.uint8)
    blob[:tab.nbytes] = tab.view(np.uint8)
    for k, (raw, h, w, c) in enumerate(items):
        blob[tab[k]["raw_off"]:tab[k]["raw_off"] + raw.size] = raw.reshape(-1)
    d = torch.from_numpy(blob).cuda()
    _lib.check(lib.bsr_png_unfilter(0, d.data_ptr(), d.numel(), 0, len(items), torch.cuda.current_stream().cuda_stream), "bsr_png_unfilter")
    torch.cuda.synchronize()
    res = d.cpu().numpy()
    ob = 1 if grey else 3
    return [res[t["out_off"]:t["out_off"] + t["h"] * t["w"] * ob].reshape(t["h"], t["w"], ob) for t in tab]


def test_device_reconstruction_of_the_reference_photographs():
    from blindshadowremoval_amd import pngio
    files = sorted(glob.glob(os.path.join(GOLDEN, "UCB", "train", "input", "*", "*.png")))[:12] + sorted(glob.glob(os.path.join(GOLDEN, "sample_imgs", "*", "*.png")))
    raws = [pngio.read_rgb_raw(f) for f in files]
    assert all(isinstance(r, pngio.RawScanlines) for r in raws)
    out = _run([(r.raw, r.h, r.w, r.c) for r in raws])
    used = np.zeros(5, int)
    for f, r, o in zip(files, raws, out):
        np.testing.assert_array_equal(o, pngio.read_rgb_u8(f), err_msg=f)
        used += np.bincount(r.raw.reshape(r.h, -1)[:, 0], minlength=5)
    assert used[2] > 0 and used[4] > 0                      # the photographs are Up / Paeth filtered: the case that matters


@pytest.mark.parametrize("h,w,c", [(256, 256, 3), (200, 37, 3), (64, 129, 1), (255, 40, 4), (1, 5, 3), (256, 16, 1), (3, 4, 1)])
def test_every_filter_type_and_channel_count(h, w, c):
    from blindshadowremoval_amd import pngio
    rng = np.random.RandomState(h * 131 + w * 7 + c)
    img = rng.randint(0, 256, (h, w, c)).astype(np.uint8)
    img[h // 3:h // 2] = img[h // 3]                            # flat stretches: ties in the Paeth predictor
    items = []
    for fts in (rng.randint(0, 5, h), np.full(h, 4), np.full(h, 3), np.arange(h) % 5):
        raw = _filter_rows(img, fts)
        np.testing.assert_array_equal(pngio.unfilter_host(raw, h, w, c), img)          # the helper above and the host statement agree
        items.append((raw, h, w, c))
    want = img if c == 3 else (np.repeat(img, 3, axis=2) if c == 1 else img[:, :, :3])
    for o in _run(items):
        np.testing.assert_array_equal(o, want)


def test_random_geometries_in_one_batch():
    """Forty images of random size (1-256 rows, 2-300 pixels wide with w c >= 4), channel count and per-row filter types in ONE launch — the
    items of a batch need not look alike."""
    from blindshadowremoval_amd import pngio
    rng = np.random.RandomState(2026)
    items, want = [], []
    while len(items) < 40:
        h, w, c = int(rng.randint(1, 257)), int(rng.randint(2, 301)), int(rng.choice([1, 3, 4]))
        if w * c < 4:
            continue
        img = rng.randint(0, 256, (h, w, c)).astype(np.uint8)
        if rng.rand() < 0.5:
            img[:, : w // 2] = img[:1, :1]                    # flat areas: predictor ties
        raw = _filter_rows(img, rng.randint(0, 5, h))
        items.append((raw, h, w, c))
        want.append(img if c == 3 else (np.repeat(img, 3, axis=2) if c == 1 else img[:, :, :3]))
    for o, wv in zip(_run(items), want):
        np.testing.assert_array_equal(o, wv)


def test_the_ucb_masks_as_grey_levels():
    """The seven segmentation masks of an item the way the UCB loop sends them (prep._masks_raw): filtered grey scanlines in, one byte per
    pixel out — the grey levels pngio.read_grey_u8 reads from the files."""
    from blindshadowremoval_amd import pngio, prep
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    cfg = Config(0)
    cfg.DATA_DIR_TEST = [os.path.join(GOLDEN, "UCB", "train", "input", "*")]
    cfg.UCB_MASK_ROOT = os.path.join(GOLDEN, "UCB_masks")
    fsr = FSRNet.__new__(FSRNet)
    fsr.config = cfg
    for paths in fsr._ucb_masks()[:3]:
        kind, raw, S = prep.pack_masks(paths, raw=True)
        assert kind == "raw8" and raw.size == 7 * S * (1 + S)
        n = S * (1 + S)
        out = _run([(raw[i * n:(i + 1) * n], S, S, 1) for i in range(7)], grey=True)
        for k, o in zip(prep.MASK_ORDER, out):
            np.testing.assert_array_equal(o[:, :, 0], pngio.read_grey_u8(paths[k]), err_msg=k)
    img = np.random.RandomState(5).randint(0, 256, (77, 130, 1)).astype(np.uint8)
    raw = _filter_rows(img, np.arange(77) % 5)
    np.testing.assert_array_equal(_run([(raw, 77, 130, 1)], grey=True)[0], img)


def test_bad_arguments_are_refused():
    import torch
    from blindshadowremoval_amd import _lib
    lib = _lib.load()
    d = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    assert lib.bsr_png_unfilter(0, d.data_ptr(), 1024, 4, 1, None) != 0          # unaligned table
    assert lib.bsr_png_unfilter(0, d.data_ptr(), 1024, 1000, 1, None) != 0       # the table leaves the blob
    assert lib.bsr_png_unfilter(0, None, 1024, 0, 1, None) != 0
