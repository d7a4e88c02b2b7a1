"""pngio.encode_png: the loops' PNG writer must be lossless and readable by an independent decoder (PIL)."""
import io
import os

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


def _filtered_png(a: np.ndarray, filters) -> bytes:
    """A PNG file of `a` [H,W,C] whose scanline y uses filter type filters[y % len(filters)] — the forward filters of the PNG
    specification (9.2) written out with numpy, so that every reconstruction branch of hostsrc/png_unfilter.c gets real input."""
    import struct
    import zlib
    from blindshadowremoval_amd.pngio import _SIGNATURE, _chunk
    h, w, c = a.shape
    x = a.reshape(h, w * c).astype(np.int32)
    left = np.zeros_like(x); left[:, c:] = x[:, :-c]
    up = np.zeros_like(x); up[1:] = x[:-1]
    ul = np.zeros_like(x); ul[1:, c:] = x[:-1, :-c]
    p = left + up - ul
    pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    forms = [x, x - left, x - up, x - ((left + up) >> 1), x - paeth]
    raw = np.empty((h, 1 + w * c), np.uint8)
    for y in range(h):
        ft = filters[y % len(filters)]
        raw[y, 0] = ft
        raw[y, 1:] = forms[ft][y] & 255
    ihdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0)
    z = zlib.compress(raw.tobytes(), 1)
    cut = len(z) // 3                                  # several IDAT chunks, as real encoders write them
    return b"".join((_SIGNATURE, _chunk(b"IHDR", ihdr), _chunk(b"IDAT", z[:cut]), _chunk(b"IDAT", z[cut:]), _chunk(b"IEND", b"")))


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("filters", [(0,), (1,), (2,), (3,), (4,), (4, 3, 1, 2, 0), (2, 4, 4, 3)])
def test_fast_reader_reconstructs_every_filter_type_like_pil(tmp_path, c, filters):
    """pngio.read_rgb_u8 / read_grey_u8 through libbsr_host.so (hostsrc/png_unfilter.c: Sub / Up / Average / Paeth, the SIMD Paeth for
    3 and 4 channels) against PIL's decoder, on smooth and on random content, odd sizes included."""
    from blindshadowremoval_amd import pngio
    assert pngio._host_lib() is not None, "libbsr_host.so did not build"
    rng = np.random.default_rng(7 * c + len(filters))
    for h, w in ((1, 1), (2, 3), (29, 37), (64, 64)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = ((yy * 3 + xx * 5)[:, :, None] + np.arange(c) * 40 + rng.integers(0, 6, (h, w, c))) & 255
        for a in (smooth.astype(np.uint8), rng.integers(0, 256, (h, w, c), dtype=np.uint8)):
            path = tmp_path / "f.png"
            path.write_bytes(_filtered_png(a, filters))
            want = Image.open(path)
            np.testing.assert_array_equal(np.asarray(want).reshape(h, w, c), a)                      # the test's own encoder is a valid PNG
            fast = pngio._decode_fast(path.read_bytes())
            assert fast is not None
            np.testing.assert_array_equal(fast, a)
            np.testing.assert_array_equal(pngio.read_rgb_u8(str(path)), np.asarray(want.convert("RGB")))
            if c == 1:
                np.testing.assert_array_equal(pngio.read_grey_u8(str(path)), a[:, :, 0])


def test_fast_reader_leaves_other_files_to_pil(tmp_path):
    """Palette, 16-bit, interlaced-free-but-gamma'd and non-PNG files: not a case for the C path, PIL's conversion comes back."""
    from blindshadowremoval_amd import pngio
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
    pal = tmp_path / "p.png"
    Image.fromarray(rgb).quantize(16).save(pal)
    deep = tmp_path / "d.png"
    Image.fromarray(rng.integers(0, 65536, (20, 24), dtype=np.uint16)).save(deep)
    jpg = tmp_path / "j.jpg"
    Image.fromarray(rgb).save(jpg, quality=90)
    for path in (pal, deep, jpg):
        assert pngio._decode_fast(path.read_bytes()) is None
        np.testing.assert_array_equal(pngio.read_rgb_u8(str(path)), np.asarray(Image.open(path).convert("RGB"), np.uint8))
    # a corrupt stream / an undefined filter type is refused by the C path (and then by PIL: the error is PIL's)
    bad = bytearray(_filtered_png(rgb, (1,)))
    assert pngio._decode_fast(bytes(bad[:-40])) is None
    import zlib, struct
    raw = np.zeros((4, 1 + 12), np.uint8); raw[2, 0] = 7
    body = b"".join((pngio._SIGNATURE, pngio._chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)), pngio._chunk(b"IDAT", zlib.compress(raw.tobytes())),
                     pngio._chunk(b"IEND", b"")))
    assert pngio._decode_fast(body) is None


def test_host_library_is_bound_to_its_source():
    from blindshadowremoval_amd import build
    path = build.build_host_library()
    assert build.host_library_sha16(path) == build.host_source_sha16() != ""


def _inflate(z: bytes, n: int):
    """libbsr_host.so's inflate on a zlib stream expected to hold n bytes: -> (return code, output)."""
    from blindshadowremoval_amd import pngio
    lib = pngio._host_lib()
    out = np.empty(n + 16, np.uint8)
    return lib.bsr_inflate_zlib(z + bytes(16), len(z), out.ctypes.data, n), out[:n].tobytes()


def test_host_inflate_equals_zlib_on_every_kind_of_block():
    """hostsrc/inflate.c against zlib: stored, fixed and dynamic blocks, every strategy and several window sizes, literal-only and
    match-heavy data (distance 1, short periods, long distances), sizes from 0 bytes to several blocks."""
    import random
    import zlib
    rng = random.Random(5)
    for level in (0, 1, 6, 9):
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
            for kind in range(6):
                for n in (rng.choice([0, 1, 2, 3, 7]), rng.choice([100, 5000, 70000, 200000])):
                    if kind == 0:
                        data = rng.randbytes(n)
                    elif kind == 1:
                        data = bytes([rng.choice(b"ab")]) * n
                    elif kind == 2:
                        data = (b"0123456789abcdef" * (n // 16 + 1))[:n]
                    elif kind == 3:
                        data = bytes((i * i >> 3) & 255 for i in range(n))
                    elif kind == 4:
                        data = bytes(n)
                    else:
                        data = bytes(rng.choice(b"aaaaabcd\n") for _ in range(n))
                    co = zlib.compressobj(level, zlib.DEFLATED, rng.choice([9, 12, 15]), rng.choice([1, 8, 9]), strat)
                    z = co.compress(data) + co.flush()
                    rc, got = _inflate(z, len(data))
                    assert rc == 0 and got == data, (level, strat, kind, n, rc)


def test_host_inflate_on_the_reference_files_and_on_damaged_streams(golden_dir):
    """Every IDAT stream of the UCB fixtures (compressed photographs, run-length masks) and of the FFHQ sample (nearly stored) inflates to
    zlib's bytes; a truncated or corrupted stream is either refused (negative code: zlib then decides) or — when the damage misses
    everything but padding — still the right bytes; nothing is ever accepted with wrong content, nothing reads or writes out of bounds."""
    import glob
    import random
    import struct
    import zlib

    def idat(b):
        o, parts = 8, []
        while o + 12 <= len(b):
            m, = struct.unpack(">I", b[o:o + 4])
            if b[o + 4:o + 8] == b"IDAT":
                parts.append(b[o + 8:o + 8 + m])
            o += 12 + m
        return b"".join(parts)
    files = (sorted(glob.glob(os.path.join(golden_dir, "UCB", "train", "*", "*", "*.png")))[::7] + sorted(glob.glob(os.path.join(golden_dir, "UCB_masks", "*", "*")))[::11]
             + glob.glob(os.path.join(golden_dir, "sample_imgs", "*", "*.png")))
    assert len(files) > 40
    for f in files:
        z = idat(open(f, "rb").read())
        want = zlib.decompress(z)
        rc, got = _inflate(z, len(want))
        assert rc == 0 and got == want, (f, rc)
    z = idat(open(files[0], "rb").read())
    want = zlib.decompress(z)
    assert _inflate(z, len(want) - 1)[0] == -6 and _inflate(z, len(want) + 1)[0] == -6          # the size must be the stream's
    rng = random.Random(9)
    refused = 0
    for _ in range(600):
        zz = bytearray(z)
        k = rng.randrange(3)
        if k == 0:
            zz = zz[:rng.randrange(len(zz))]
        elif k == 1:
            for _ in range(rng.randrange(1, 4)):
                zz[rng.randrange(len(zz))] ^= 1 << rng.randrange(8)
        else:
            zz[rng.randrange(len(zz))] = rng.randrange(256)
        rc, got = _inflate(bytes(zz), len(want))
        assert rc <= 0 and (rc != 0 or got == want)
        refused += rc != 0
    assert refused > 550


def test_inflate_adler32_over_lengths_and_extreme_bytes():
    """The Adler-32 that vouches for the inflate's output (hostsrc/inflate.c: 32 bytes per step on AVX2 since round 6, scalar elsewhere) must
    ACCEPT every valid stream — a wrong sum would only show as a silent fallback to zlib: lengths around the 32-byte step and the 5 536-byte
    run, all-0xFF data (the largest sums), random data."""
    import zlib
    from blindshadowremoval_amd import pngio
    lib = pngio._host_lib()
    if lib is None:
        pytest.skip("no C compiler: libbsr_host.so unavailable")
    rng = np.random.RandomState(7)
    for n in list(range(0, 70)) + [5535, 5536, 5537, 5567, 5568, 11071, 11072, 11073, 65521, 200000]:
        for data in (bytes([255]) * n, rng.randint(0, 256, n).astype(np.uint8).tobytes()):
            for level in (0, 6):
                z = zlib.compress(data, level) + bytes(16)
                out = np.empty(n + 16, np.uint8)
                assert lib.bsr_inflate_zlib(z, len(z) - 16, out.ctypes.data, n) == 0, (n, level)
                assert out[:n].tobytes() == data


def test_raw_scanlines_and_the_host_reconstruction_without_the_c_library(tmp_path, monkeypatch):
    """pngio.read_rgb_raw hands back the inflated, still filtered scanlines of a plain 8-bit PNG (what the UCB loop's workers put into
    their ring slot since round 6) and everything else decoded; RawScanlines.decode / unfilter_host give PIL's pixels — through
    libbsr_host.so and through the plain numpy statement used where no C compiler exists."""
    from blindshadowremoval_amd import pngio
    rng = np.random.RandomState(11)
    for mode, c in (("RGB", 3), ("L", 1), ("RGBA", 4)):
        a = rng.randint(0, 256, (23, 31, c)).astype(np.uint8)
        a[5:9] = a[5]
        f = str(tmp_path / ("x_%s.png" % mode))
        Image.fromarray(a[:, :, 0] if c == 1 else a, mode).save(f)
        r = pngio.read_rgb_raw(f)
        assert isinstance(r, pngio.RawScanlines) and (r.h, r.w, r.c) == (23, 31, c) and r.raw.size == 23 * (1 + 31 * c) and r.shape == (23, 31, 3)
        want = np.asarray(Image.open(f).convert("RGB"))
        assert np.array_equal(r.decode(), want) and np.array_equal(pngio.read_rgb_u8(f), want)
        monkeypatch.setattr(pngio, "_HOST", [None, True])                  # no C library: the numpy statement
        assert np.array_equal(pngio._to_rgb(pngio.unfilter_host(r.raw, r.h, r.w, r.c)), want)
        monkeypatch.undo()
    pal = str(tmp_path / "p.png")
    Image.fromarray(rng.randint(0, 4, (9, 9)).astype(np.uint8), "P").save(pal)      # a palette file is PIL's: decoded, not raw
    d = pngio.read_rgb_raw(pal)
    assert isinstance(d, np.ndarray) and d.shape == (9, 9, 3)
    with pytest.raises(ValueError):
        pngio.unfilter_host(np.zeros(10, np.uint8), 2, 2, 3)
