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
