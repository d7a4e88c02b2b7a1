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


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(16, 256, 256), (3, 64, 40), (2, 5, 7), (1, 1, 1)])
def test_files_from_figures_equal_files_from_strips(geom):
    """bsr_png_encode_figs (round 5): the PNG files written straight from the float figures — channel slices of a wider NHWC tensor, a
    one-channel figure, a one-channel figure times a multiplier image times 2 (test_step_FFHQ's strip) — are byte for byte the files of
    the strip path (Logging.strips_on_device -> bsr_png_encode), values below 0, above 1 and exactly on .5 ties included."""
    import torch
    from blindshadowremoval_amd.fsrnet import Logging
    from blindshadowremoval_amd.gpu_png import StripEncoder
    b, h, wf = geom
    g = torch.Generator(device="cpu").manual_seed(b * 1000 + h * 10 + wf)
    rows = (torch.rand((b, h, wf, 16), generator=g) * 1.6 - 0.3).cuda(0)                 # spills over both ends of [0, 1]
    ties = ((torch.arange(b * h * wf * 3, dtype=torch.float32).reshape(b, h, wf, 3) % 255) + 0.5) / 255.0      # k + 0.5 after the x 255 where fp32 allows it
    packed = torch.cat([ties.cuda(0), torch.rand((b, h, wf, 1), generator=g).cuda(0) * 1.5 - 0.25], dim=3).contiguous()
    im, face = rows[..., 0:3], rows[..., 15:16]
    con, mask = packed[..., 0:3], packed[..., 3:4]
    enc = StripEncoder(0)
    lazy = enc.encode_figs([im, con, (mask, face, 2.0), mask])
    assert lazy is not None
    strips = Logging.strips_on_device([im, torch.clamp(con, 0, 1), mask * face * 2, mask])
    want = enc.encode(strips)
    torch.cuda.synchronize()
    assert torch.equal(lazy, want)
    assert np.array_equal(_decode(lazy[b - 1].cpu().numpy().tobytes()), strips[b - 1].cpu().numpy())
    # a second call reuses the scratch (its accumulators are cleared by the call itself)
    again = enc.encode_figs([im, con, (mask, face, 2.0), mask])
    torch.cuda.synchronize()
    assert torch.equal(again, want)
    # layouts the kernel does not address go back to the caller
    assert enc.encode_figs([im.permute(0, 2, 1, 3), con]) is None or h == wf
    assert enc.encode_figs([im.double(), con]) is None
    assert enc.encode_figs([rows[..., 0:2], con]) is None
