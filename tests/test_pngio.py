"""pngio.encode_png: the loops' PNG writer must be lossless and readable by an independent decoder (PIL)."""
import io

import numpy as np
import pytest
from PIL import Image

from blindshadowremoval_amd.pngio import encode_png, write_png


@pytest.mark.parametrize("shape", [(1, 1, 3), (7, 5, 3), (256, 1024, 3), (33, 17), (9, 4, 1), (16, 16, 4)])
def test_round_trip_through_pil(shape):
    rng = np.random.RandomState(sum(shape))
    a = rng.randint(0, 256, size=shape).astype(np.uint8)
    if len(shape) == 3 and shape[0] >= 16:
        a[: shape[0] // 2] = 200                     # flat areas: runs for the RLE strategy
    im = Image.open(io.BytesIO(encode_png(a)))
    im.load()
    back = np.asarray(im)
    assert back.shape == (a[:, :, 0].shape if a.ndim == 3 and a.shape[2] == 1 else a.shape)
    assert np.array_equal(back.reshape(a.shape), a)


def test_writes_file_and_creates_directory(tmp_path):
    a = (np.arange(8 * 12 * 3) % 251).astype(np.uint8).reshape(8, 12, 3)
    out = tmp_path / "a" / "b" / "strip.png"
    write_png(str(out), a)
    assert np.array_equal(np.asarray(Image.open(out).convert("RGB")), a)


@pytest.mark.parametrize("bad", [np.zeros((4, 4, 3), np.float32), np.zeros((4, 4, 2), np.uint8), np.zeros((0, 4, 3), np.uint8), np.zeros((4,), np.uint8)])
def test_rejects_what_it_cannot_write(bad):
    with pytest.raises(ValueError):
        encode_png(bad)


def test_fast_grey_reader_equals_pil(tmp_path):
    """pngio.read_grey_u8: the inflate + cumulative-sum path for 8-bit greyscale PNGs with filter types 0 / 1 / 2 (what the UCB masks
    are) and its PIL fallback both return PIL's convert("L") — on the shipped masks, on files with mixed filters, on RGB and palette."""
    import glob
    import os
    import zlib
    import struct
    from PIL import Image
    from blindshadowremoval_amd.pngio import read_grey_u8, _chunk, _SIGNATURE
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "UCB_masks")
    files = sorted(glob.glob(os.path.join(golden, "*", "*.png")))[::37]
    assert len(files) >= 10
    for f in files:
        assert np.array_equal(read_grey_u8(f), np.asarray(Image.open(f).convert("L"), np.uint8)), f
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (19, 23), dtype=np.uint8)
    # hand-filtered file: rows cycle through None / Sub / Up
    raw = bytearray()
    for y in range(a.shape[0]):
        ft = y % 3
        row = a[y].astype(np.int16)
        if ft == 1:
            row = row - np.concatenate([[0], a[y, :-1].astype(np.int16)])
        elif ft == 2:
            row = row - (a[y - 1].astype(np.int16) if y else 0)
        raw += bytes([ft]) + (row % 256).astype(np.uint8).tobytes()
    p = tmp_path / "mixed.png"
    p.write_bytes(_SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", a.shape[1], a.shape[0], 8, 0, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(bytes(raw))) + _chunk(b"IEND", b""))
    assert np.array_equal(np.asarray(Image.open(p)), a) and np.array_equal(read_grey_u8(str(p)), a)
    for mode in ("RGB", "P", "L"):                              # PIL-written files (adaptive filters, other colour types): the fallback
        q = tmp_path / ("pil_%s.png" % mode)
        Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).convert(mode).save(q)
        assert np.array_equal(read_grey_u8(str(q)), np.asarray(Image.open(q).convert("L"), np.uint8))
