"""Reader for TensorFlow tensor-bundle checkpoints (``ckpt-N.index`` + ``ckpt-N.data-00000-of-00001``).

The reference stores its generator weights with ``tf.train.Checkpoint(generator=self.gen, ...)``
(/root/reference/train_test_GSC.py:142-148) and restores them in ``test``/``testFFHQ``
(/root/reference/train_test_GSC.py:362-365, 842-845).  No TensorFlow exists on the MI355X box, so
this module parses the on-disk format directly:

* ``.index`` is a LevelDB *table*: data blocks of prefix-compressed (key, value) entries, an index
  block of block handles, and a 48-byte footer ending in the magic ``0xdb4775248b80fb57``.
* each value is a ``BundleEntryProto`` {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset,
  5: size, 6: crc32c}; the key ``""`` holds the ``BundleHeaderProto``.
* tensor bytes live at ``[offset, offset+size)`` of the data shard, little-endian, C order.

Only what the hot path needs is implemented: uncompressed blocks, float32 tensors, a single shard.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT_FLOAT = 1
GEN_PREFIX = "generator/"
VAR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(buf: bytes, off: int, size: int) -> Iterator[Tuple[bytes, bytes]]:
    if buf[off + size] != 0:
        raise ValueError("compressed table block (type %d) is not supported" % buf[off + size])
    blk = buf[off:off + size]
    n_restarts = struct.unpack_from("<I", blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        yield key, blk[pos:pos + vlen]
        pos += vlen


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire parser: field number -> list of raw values."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for d in _parse_proto(buf).get(2, []):
        dims.append(_parse_proto(d).get(1, [0])[0])
    return tuple(dims)


def _is_generator_var(key: str) -> bool:
    """Model variables only: Adam slots live under ``<var>/.OPTIMIZER_SLOT/...`` and are skipped."""
    return key.startswith(GEN_PREFIX) and key.endswith(VAR_SUFFIX) and ".OPTIMIZER_SLOT" not in key


class BundleEntry:
    __slots__ = ("dtype", "shape", "shard", "offset", "size")

    def __init__(self, dtype, shape, shard, offset, size):
        self.dtype, self.shape, self.shard, self.offset, self.size = dtype, shape, shard, offset, size

    def __repr__(self):
        return "BundleEntry(dtype=%d, shape=%s, offset=%d, size=%d)" % (self.dtype, self.shape, self.offset, self.size)


def read_index(index_path: str) -> Dict[str, BundleEntry]:
    """Parse ``ckpt-N.index`` into {key: BundleEntry}; the header entry (key "") is skipped."""
    with open(index_path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad magic)" % index_path)
    foot = len(buf) - 48
    _, p = _varint(buf, foot)          # metaindex offset
    _, p = _varint(buf, p)             # metaindex size
    idx_off, p = _varint(buf, p)
    idx_size, p = _varint(buf, p)
    entries: Dict[str, BundleEntry] = {}
    for _, handle in _block_entries(buf, idx_off, idx_size):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        for key, val in _block_entries(buf, boff, bsize):
            if key == b"":
                continue
            pr = _parse_proto(val)
            entries[key.decode("utf-8")] = BundleEntry(
                pr.get(1, [0])[0], _shape(pr[2][0]) if 2 in pr else (),
                pr.get(3, [0])[0], pr.get(4, [0])[0], pr.get(5, [0])[0])
    return entries


def generator_inventory(index_path: str) -> Dict[str, Tuple[int, ...]]:
    """{reference variable name (e.g. ``res_stack/0/conv1/kernel``): shape} for ``generator/*``."""
    inv = {}
    for key, e in read_index(index_path).items():
        if _is_generator_var(key) and e.dtype == _DT_FLOAT:
            inv[key[len(GEN_PREFIX):-len(VAR_SUFFIX)]] = e.shape
    return inv


def latest_checkpoint(ckpt_dir: str) -> Optional[str]:
    """Counterpart of ``tf.train.latest_checkpoint`` (/root/reference/train_test_GSC.py:362,842):
    reads the text ``checkpoint`` file and returns the prefix path, or None."""
    state = os.path.join(ckpt_dir, "checkpoint")
    if not os.path.isfile(state):
        return None
    with open(state, "r") as f:
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
    if not m:
        return None
    prefix = m.group(1)
    if not os.path.isabs(prefix):
        prefix = os.path.join(ckpt_dir, prefix)
    return prefix if os.path.isfile(prefix + ".index") else None


def load_generator_weights(prefix: str) -> Dict[str, np.ndarray]:
    """Load every ``generator/*`` float32 variable of checkpoint ``prefix`` (path without suffix).

    Raises FileNotFoundError when the data shard is absent (the reference repo ships only the
    ``.index`` files: /root/reference/.MISSING_LARGE_BLOBS)."""
    entries = read_index(prefix + ".index")
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.isfile(data_path):
        raise FileNotFoundError("checkpoint data shard missing: " + data_path)
    out = {}
    with open(data_path, "rb") as f:
        for key, e in entries.items():
            if not _is_generator_var(key) or e.dtype != _DT_FLOAT:
                continue
            if e.shard != 0:
                raise ValueError("multi-shard bundles are not supported")
            f.seek(e.offset)
            raw = f.read(e.size)
            arr = np.frombuffer(raw, dtype="<f4").reshape(e.shape).copy()
            out[key[len(GEN_PREFIX):-len(VAR_SUFFIX)]] = arr
    return out


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write a single-block-per-entry tensor bundle (used by tests to round-trip the reader and to
    let users stage real weights).  Keys get the ``generator/`` prefix and variable suffix."""
    items: List[Tuple[bytes, np.ndarray]] = []
    for name in sorted(tensors):
        items.append(((GEN_PREFIX + name + VAR_SUFFIX).encode(), np.ascontiguousarray(tensors[name], dtype="<f4")))
    items.sort(key=lambda kv: kv[0])

    def enc_varint(v: int) -> bytes:
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            if v:
                out.append(b | 0x80)
            else:
                out.append(b)
                return bytes(out)

    def field(num: int, wt: int, payload: bytes) -> bytes:
        return enc_varint((num << 3) | wt) + payload

    data = bytearray()
    block = bytearray()
    # header entry: key "" -> BundleHeaderProto{num_shards=1}
    hdr = field(1, 0, enc_varint(1))
    block += enc_varint(0) + enc_varint(0) + enc_varint(len(hdr)) + hdr
    restarts = [0]
    for key, arr in items:
        shape = b"".join(field(2, 2, enc_varint(len(d)) + d) for d in
                         (field(1, 0, enc_varint(int(s))) for s in arr.shape))
        val = (field(1, 0, enc_varint(_DT_FLOAT)) + field(2, 2, enc_varint(len(shape)) + shape) +
               field(4, 0, enc_varint(len(data))) + field(5, 0, enc_varint(arr.nbytes)))
        restarts.append(len(block))
        block += enc_varint(0) + enc_varint(len(key)) + enc_varint(len(val)) + key + val
        data += arr.tobytes()
    for r in restarts:
        block += struct.pack("<I", r)
    block += struct.pack("<I", len(restarts))
    out = bytearray(block) + b"\x00" + b"\x00\x00\x00\x00"          # type byte + (unchecked) crc
    data_handle = enc_varint(0) + enc_varint(len(block))
    # metaindex block (empty) and index block (one entry pointing at the data block)
    meta_off = len(out)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)
    out += meta + b"\x00" + b"\x00\x00\x00\x00"
    idx_off = len(out)
    last_key = items[-1][0] + b"\xff" if items else b"\xff"
    idx = enc_varint(0) + enc_varint(len(last_key)) + enc_varint(len(data_handle)) + last_key + data_handle
    idx += struct.pack("<I", 0) + struct.pack("<I", 1)
    out += idx + b"\x00" + b"\x00\x00\x00\x00"
    footer = enc_varint(meta_off) + enc_varint(len(meta)) + enc_varint(idx_off) + enc_varint(len(idx))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' %
                (os.path.basename(prefix), os.path.basename(prefix)))
