"""PNG writer of the test loops' figure strips (`Logging.save_img` -> cv2.imwrite in the reference,
/root/reference/utils.py:196-204): 8-bit grey / RGB / RGBA, lossless, so any conforming encoder gives the reference's pixels back.

The strips are 256 x (256 * 4..8) photographs: PIL's encoder (adaptive filter heuristics + deflate level 1) needs 15-50 ms per strip,
which made PNG encoding the largest CPU cost of both loops (16 usable CPUs on the GPU box).  This writer does what cv2.imwrite does by
default — the Sub filter on every row and zlib's run-length strategy (IMWRITE_PNG_STRATEGY_RLE, level 1) — with the filter as ONE numpy
subtraction over the whole strip: 2-3x faster than PIL at the same file size (photographic content is Huffman-bound, not match-bound).
"""
import os
import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_COLOR_TYPE = {1: 0, 3: 2, 4: 6}        # channels -> PNG colour type (grey, truecolour, truecolour + alpha)


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)


def encode_png(a: np.ndarray) -> bytes:
    """uint8 [H,W] / [H,W,1|3|4] -> the bytes of a PNG file."""
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim not in (2, 3):
        raise ValueError("encode_png takes a uint8 [H,W] or [H,W,C] array, got %s %s" % (a.dtype, a.shape))
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    if c not in _COLOR_TYPE or h == 0 or w == 0:
        raise ValueError("encode_png: unsupported shape %s" % (a.shape,))
    flat = np.ascontiguousarray(a).reshape(h, w * c)
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 1                                                        # filter type 1 (Sub) on every scanline
    raw[:, 1:1 + c] = flat[:, :c]                                        # the first pixel has no left neighbour
    np.subtract(flat[:, c:], flat[:, :-c], out=raw[:, 1 + c:])           # byte-wise, modulo 256
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    data = co.compress(memoryview(raw).cast("B")) + co.flush()
    return b"".join((_SIGNATURE, _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, _COLOR_TYPE[c], 0, 0, 0)), _chunk(b"IDAT", data),
                     _chunk(b"IEND", b"")))


_HOST = [None, False]          # [ctypes handle of libbsr_host.so, tried]


def _host_lib():
    """libbsr_host.so (hostsrc/png_unfilter.c, gcc) — built on first use if the tree holds none for the current source; None when no C
    compiler exists (the readers then go through PIL: same pixels, 4x the time)."""
    if not _HOST[1]:
        _HOST[1] = True
        try:
            import ctypes
            from .build import build_host_library
            lib = ctypes.CDLL(build_host_library())
            lib.bsr_png_unfilter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            lib.bsr_png_unfilter.restype = ctypes.c_int
            lib.bsr_inflate_zlib.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            lib.bsr_inflate_zlib.restype = ctypes.c_int
            _HOST[0] = lib
        except (OSError, RuntimeError) as e:
            import warnings
            warnings.warn("libbsr_host.so unavailable (%s): PNG files are decoded by PIL" % e)
    return _HOST[0]


def _parse_8bit(b: bytes):
    """-> (w, h, channels, inflated scanlines) of a non-interlaced 8-bit grey / RGB / RGBA file without palette, transparency or gamma
    chunks; ValueError for anything else (PIL's business).  The chunks' CRC-32 fields are NOT verified (PIL would reject a file whose
    chunk CRC is wrong; here the zlib stream's Adler-32 and its exact length h * (1 + w c) are what vouch for the pixels)."""
    if b[:8] != _SIGNATURE:
        raise ValueError
    o, idat, hdr = 8, [], None
    while o + 12 <= len(b):
        n, = struct.unpack(">I", b[o:o + 4])
        tag = b[o + 4:o + 8]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", b[o + 8:o + 8 + n])
        elif tag == b"IDAT":
            idat.append(b[o + 8:o + 8 + n])
        elif tag == b"IEND":
            break
        elif tag in (b"PLTE", b"tRNS", b"gAMA"):
            raise ValueError
        o += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (0, 2, 6) or interlace != 0 or w == 0 or h == 0 or w * h > 1 << 28:      # absurd sizes: PIL's bomb check decides
        raise ValueError
    c = {0: 1, 2: 3, 6: 4}[ctype]
    n = h * (1 + w * c)
    lib = _HOST[0]
    if lib is not None:
        # libbsr_host.so's inflate (hostsrc/inflate.c: 1.9x zlib's rate on photographs); it wants 16 readable bytes behind the stream and
        # 16 writable ones behind the output.  Whatever it refuses goes to zlib, whose verdict counts.
        z = b"".join(idat + [bytes(16)])
        raw = np.empty(n + 16, np.uint8)
        if lib.bsr_inflate_zlib(z, len(z) - 16, raw.ctypes.data, n) == 0:
            return w, h, c, raw[:n]
    # the size is known: ask for at most n + 1 bytes — a crafted stream that inflates far beyond h * (1 + w * c) is cut off there instead of
    # being materialised whole before the length check (zlib.decompress's third argument is only an initial buffer size)
    d = zlib.decompressobj(15)
    data = d.decompress(b"".join(idat) if len(idat) != 1 else idat[0], n + 1)
    if len(data) != n or d.unconsumed_tail or not d.eof:
        raise ValueError
    return w, h, c, np.frombuffer(data, np.uint8)


def _decode_fast(b: bytes):
    """uint8 [H,W,C] of a plain 8-bit file through libbsr_host.so, or None when the file (or the box: no library) is not a case for it."""
    try:
        lib = _host_lib()
        if lib is None:
            return None
        w, h, c, raw = _parse_8bit(b)
        out = np.empty((h, w, c), np.uint8)
        return out if lib.bsr_png_unfilter(raw.ctypes.data, h, w * c, c, out.ctypes.data) == 0 else None
    except (ValueError, TypeError, struct.error, zlib.error):
        return None


def read_rgb_u8(path: str) -> np.ndarray:
    """An image file as uint8 [H,W,3] RGB = PIL's open(path).convert("RGB").  Fast path for what the reference's inputs are (8-bit,
    non-interlaced grey / RGB / RGBA PNG: grey replicated, alpha dropped, as PIL converts them): zlib.decompress of the whole IDAT
    stream + the scanline reconstruction in C (hostsrc/png_unfilter.c: one pixel per step with the channels in SIMD lanes) — 0.5 ms
    for a 256x256 RGB photograph against PIL's 2.2 ms, whose decoder spends 2.0 of them in the Paeth / Average reconstruction.
    Everything else (palette, 16-bit, interlaced, tRNS / gAMA chunks, not a PNG at all) is PIL's."""
    with open(path, "rb") as f:
        b = f.read()
    a = _decode_fast(b)
    if a is None:
        import io
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(b)).convert("RGB"), np.uint8))
    return _to_rgb(a)


class RawScanlines:
    """An inflated, still FILTERED 8-bit PNG image: `raw` = h x (1 + w c) bytes (filter-type byte first), c = 1 | 3 | 4.  What a
    loader's worker hands over when the scanline reconstruction runs on the device (csrc/prep_kernels.h png_unfilter_kernel)."""
    __slots__ = ("h", "w", "c", "raw")

    def __init__(self, h: int, w: int, c: int, raw: np.ndarray):
        self.h, self.w, self.c, self.raw = int(h), int(w), int(c), raw

    @property
    def shape(self):                       # of the RGB image it reconstructs to
        return (self.h, self.w, 3)

    @property
    def nbytes(self) -> int:
        return int(self.raw.nbytes)

    def decode(self) -> np.ndarray:
        """The RGB8 image on the host (= read_rgb_u8 of the file): the C reconstruction, or PIL's arithmetic restated in numpy without it."""
        return _to_rgb(unfilter_host(self.raw, self.h, self.w, self.c))


def _to_rgb(a: np.ndarray) -> np.ndarray:
    if a.shape[2] == 3:
        return a
    return np.repeat(a, 3, axis=2) if a.shape[2] == 1 else np.ascontiguousarray(a[:, :, :3])


def unfilter_host(raw: np.ndarray, h: int, w: int, c: int) -> np.ndarray:
    """Filtered scanlines -> uint8 [h,w,c] on the host: libbsr_host.so, or (no C compiler) a plain numpy statement of RFC 2083 section 6."""
    raw = np.ascontiguousarray(raw, np.uint8).reshape(-1)
    if raw.size != h * (1 + w * c):
        raise ValueError("unfilter_host: %d bytes for a %dx%dx%d image" % (raw.size, h, w, c))
    lib = _host_lib()
    if lib is not None:
        out = np.empty((h, w, c), np.uint8)
        if lib.bsr_png_unfilter(raw.ctypes.data, h, w * c, c, out.ctypes.data) == 0:
            return out
        raise ValueError("unfilter_host: a scanline has a filter type above 4")
    rows = raw.reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.int32)
    for y in range(h):
        ft, f = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        up = out[y - 1] if y > 0 else np.zeros(w * c, np.int32)
        if ft == 0:
            out[y] = f
        elif ft == 2:
            out[y] = (f + up) & 255
        elif ft in (1, 3, 4):
            for i in range(w * c):
                a = out[y, i - c] if i >= c else 0
                b = up[i]
                cc = up[i - c] if i >= c else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    p0 = a + b - cc
                    pa, pb, pc = abs(p0 - a), abs(p0 - b), abs(p0 - cc)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                out[y, i] = (f[i] + pr) & 255
        else:
            raise ValueError("unfilter_host: filter type %d" % ft)
    return out.astype(np.uint8).reshape(h, w, c)


def read_rgb_raw(path: str):
    """An image file for the DEVICE reconstruction: RawScanlines (the inflated stream of a plain 8-bit grey / RGB / RGBA PNG — the
    worker stops after the inflate) or, for every other file, the decoded uint8 [H,W,3] image exactly as read_rgb_u8 returns it."""
    with open(path, "rb") as f:
        b = f.read()
    try:
        _host_lib()                                            # (its inflate, when there is one)
        w, h, c, raw = _parse_8bit(b)
        return RawScanlines(h, w, c, raw)
    except (ValueError, TypeError, struct.error, zlib.error):
        import io
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(b)).convert("RGB"), np.uint8))


def read_grey_u8(path: str) -> np.ndarray:
    """An image file as 8-bit grey levels [H,W] (= PIL's open(path).convert("L")).  Fast path for what the UCB segmentation masks are
    (cv2.imwrite output: 8-bit greyscale, non-interlaced, filter type 0 / 1 / 2 on every scanline): inflate + one cumulative sum —
    0.06 ms instead of PIL's 0.45 ms per 256x256 mask, seven masks per item in the loaders' workers.  Anything else goes to PIL."""
    with open(path, "rb") as f:
        b = f.read()
    a = _decode_fast(b)                                        # round 5: any filter type, 0.1 ms (libbsr_host.so); below: the numpy form without it
    if a is not None and a.shape[2] == 1:
        return a[:, :, 0]
    try:
        if b[:8] != _SIGNATURE:
            raise ValueError
        o, idat, hdr = 8, [], None
        while o + 12 <= len(b):
            n, = struct.unpack(">I", b[o:o + 4])
            tag = b[o + 4:o + 8]
            if tag == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", b[o + 8:o + 8 + n])
            elif tag == b"IDAT":
                idat.append(b[o + 8:o + 8 + n])
            elif tag == b"IEND":
                break
            elif tag in (b"PLTE", b"tRNS", b"gAMA"):          # palette / transparency / gamma: PIL's business
                raise ValueError
            o += 12 + n
        w, h, depth, ctype, _, _, interlace = hdr
        if depth != 8 or ctype != 0 or interlace != 0:
            raise ValueError
        raw = np.frombuffer(zlib.decompress(b"".join(idat), 15, h * (1 + w)), np.uint8).reshape(h, 1 + w)
        ft = raw[:, 0]
        if ft.max() > 2:
            raise ValueError
        out = raw[:, 1:].copy()
        sub = ft == 1
        if sub.all():
            return np.cumsum(out, axis=1, dtype=np.uint8)      # Sub: x[i] = f[i] + x[i-1] (mod 256)
        if sub.any():
            out[sub] = np.cumsum(out[sub], axis=1, dtype=np.uint8)
        for y in np.nonzero(ft == 2)[0]:                       # Up: x[y] = f[y] + x[y-1] (mod 256), top row: + 0
            if y > 0:
                out[y] += out[y - 1]
        return out
    except (ValueError, TypeError, struct.error, zlib.error):
        from PIL import Image
        return np.asarray(Image.open(path).convert("L"), np.uint8)


def stored_layout(h: int, w: int):
    """(scanline bytes, scanlines per stored block, blocks, zlib-stream bytes, file bytes) of the STORED-deflate RGB file the device
    encoder writes (csrc/png_kernels.h: its layout is a pure function of H and W)."""
    rb = 1 + 3 * w
    r = 65535 // rb
    if r < 1:
        raise ValueError("a %d-pixel scanline does not fit one stored deflate block" % w)
    nblocks = (h + r - 1) // r
    zlen = 2 + 5 * nblocks + h * rb + 4
    return rb, r, nblocks, zlen, 8 + 25 + 8 + zlen + 4 + 12


def encode_png_stored(a: np.ndarray) -> bytes:
    """uint8 [H,W,3] -> the bytes of the PNG file csrc/png_kernels.h builds on the device, byte for byte: filter type 0 on every
    scanline, the zlib stream as stored deflate blocks of whole scanlines.  The host statement of that format: tests compare the
    device encoder with it, and it is itself checked against an independent decoder (PIL)."""
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("encode_png_stored takes a uint8 [H,W,3] array, got %s %s" % (a.dtype, a.shape))
    h, w, _ = a.shape
    rb, r, nblocks, zlen, total = stored_layout(h, w)
    raw = np.zeros((h, rb), np.uint8)
    raw[:, 1:] = a.reshape(h, 3 * w)
    parts = [b"\x78\x01"]
    for b in range(nblocks):
        rows = raw[b * r:(b + 1) * r]
        n = rows.size
        parts.append(struct.pack("<BHH", 1 if b == nblocks - 1 else 0, n, n ^ 0xFFFF))
        parts.append(rows.tobytes())
    parts.append(struct.pack(">I", zlib.adler32(raw.tobytes()) & 0xFFFFFFFF))
    data = b"".join(parts)
    assert len(data) == zlen
    out = b"".join((_SIGNATURE, _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)), _chunk(b"IDAT", data), _chunk(b"IEND", b"")))
    assert len(out) == total
    return out


def write_png(path: str, a: np.ndarray) -> None:
    """Write `a` to `path` (the directory is created when missing)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    if os.environ.get("BSR_PNG_WRITER") == "pil":          # A/B switch for measurements only (scratch/post_scaling.py, loop_tune.py)
        from PIL import Image
        Image.fromarray(np.asarray(a)).save(path, compress_level=1)
        return
    with open(path, "wb") as f:
        f.write(encode_png(a))
