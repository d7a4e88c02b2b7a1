"""The device-side PNG writer (csrc/png_kernels.h) against its host statement (pngio.encode_png_stored) byte for byte, and both
against an independent decoder (PIL verifies every chunk CRC, zlib the Adler-32)."""
import io
import zlib

import numpy as np
import pytest

from blindshadowremoval_amd import pngio


def _decode(data: bytes) -> np.ndarray:
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    return np.asarray(im.convert("RGB"))


@pytest.mark.parametrize("shape", [(256, 768), (256, 1792), (5, 7), (1, 1), (37, 5461), (300, 21844 // 4)])
def test_host_statement_decodes_to_the_pixels(shape):
    h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    data = pngio.encode_png_stored(a)
    rb, r, nblocks, zlen, total = pngio.stored_layout(h, w)
    assert len(data) == total and data[37:41] == b"IDAT"
    assert np.array_equal(_decode(data), a)
    # the zlib stream on its own: stored blocks, Adler-32 accepted
    raw = zlib.decompress(data[41:41 + zlen])
    assert len(raw) == h * rb and raw[0] == 0


def test_crc_combination_identity():
    """What the device reduction rests on (zlib's crc32_combine, restated in csrc/png_kernels.h): the CRC register after A | B from
    initial value I equals shift(register after A from I, |B|) ^ register after B from 0 — checked here on the host with a bitwise
    CRC and the same multmodp / x^(8n) arithmetic."""
    POLY = 0xEDB88320

    def raw_crc(data, c):
        for byte in data:
            c ^= byte
            for _ in range(8):
                c = (c >> 1) ^ POLY if c & 1 else c >> 1
        return c

    def multmodp(a, b):
        m, p = 1 << 31, 0
        while True:
            if a & m:
                p ^= b
                if a & (m - 1) == 0:
                    break
            m >>= 1
            b = (b >> 1) ^ POLY if b & 1 else b >> 1
        return p

    x2n = [1 << 30]
    for _ in range(31):
        x2n.append(multmodp(x2n[-1], x2n[-1]))

    def shift(c, n):
        k = 3
        while n:
            if n & 1:
                c = multmodp(x2n[k & 31], c)
            n >>= 1
            k += 1
        return c

    rng = np.random.default_rng(7)
    for la, lb in ((1, 1), (5, 300), (1000, 3), (257, 4099)):
        a, b = bytes(rng.integers(0, 256, la, dtype=np.uint8)), bytes(rng.integers(0, 256, lb, dtype=np.uint8))
        whole = raw_crc(a + b, 0xFFFFFFFF)
        assert whole == shift(raw_crc(a, 0xFFFFFFFF), lb) ^ raw_crc(b, 0)
        assert whole ^ 0xFFFFFFFF == zlib.crc32(a + b)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 256, 768), (3, 256, 1792), (2, 5, 7), (1, 1, 1), (2, 37, 5461), (5, 64, 250)])
def test_device_encoder_matches_the_host_statement(shape):
    import torch
    from blindshadowremoval_amd.gpu_png import StripEncoder, file_bytes
    b, h, w = shape
    g = torch.Generator().manual_seed(b * 1000 + w)
    strips = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8)
    strips[0, :, : max(1, w // 3)] = 255                       # runs of equal bytes too
    enc = StripEncoder(0)
    for _ in range(2):                                         # second call: the accumulators are cleared per call
        out = enc.encode(strips.cuda()).cpu().numpy()
    assert out.shape == (b, file_bytes(h, w))
    for i in range(b):
        data = out[i].tobytes()
        assert data == pngio.encode_png_stored(strips[i].numpy()), (shape, i)
        assert np.array_equal(_decode(data), strips[i].numpy())


@pytest.mark.gpu
def test_device_encoder_refuses_bad_arguments():
    import torch
    from blindshadowremoval_amd.gpu_png import StripEncoder
    enc = StripEncoder(0)
    with pytest.raises(TypeError):
        enc.encode(torch.zeros((1, 4, 4, 3), device="cuda"))
    with pytest.raises(ValueError):
        enc.encode(torch.zeros((1, 4, 4, 3), dtype=torch.uint8))
    with pytest.raises(ValueError):
        enc.encode(torch.zeros((1, 4, 6000, 3), dtype=torch.uint8, device="cuda"))
