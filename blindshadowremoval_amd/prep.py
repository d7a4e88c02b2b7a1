"""Device-side input preparation (round 3): the reference's per-sample loader work — `parse_fn_test_FFHQ` / `parse_fn_test`
(/root/reference/dataset.py:619-638, 148-170) — split so that only what is tiny and irregular stays on the host.

host (`host_part`, runs in the loader's worker processes): PNG decode, `face_crop_and_resize`'s crop box and landmark
    normalisation (utils.py:366-433, the float32 / float64 dtype flow of dataset.face_crop_and_resize), the Delaunay
    triangulations (matplotlib.tri.Triangulation = qhull, exactly the reference's call, <= 101 points each) and
    `Triangulation.calculate_plane_coefficients` — the numbers LinearTriInterpolator evaluates;
device (`device_rows` -> bsr_prep_rows, csrc/prep_kernels.h): the bilinear crop-resize of image + ground truth, the seven
    interpolated channels (uv map, reg_in, reg_out), the face-hull mask and its 5x5 Gaussian, all in float64 in the reference's
    operation order, written as the packed `[B,S,S,16]` float32 tensor directly in HBM.

Pinned by the same fixtures as the host path (tests/test_prep_gpu.py: tests/golden/sample_02165.npz — made by the reference's OWN
functions — and the host `build_row` on the UCB items, 1e-6)."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import dataset as D

TRI_DOUBLES = 18          # csrc/prep_kernels.h kPrepTriDoubles
MAX_TRI = 256
ROW_DTYPE = np.dtype([("img_off", "<i8"), ("gt_off", "<i8"), ("h", "<i4"), ("w", "<i4"), ("box", "<i4", (4,)),
                      ("tri_off", "<i8", (4,)), ("ntri", "<i4", (4,))], align=True)
assert ROW_DTYPE.itemsize == 88


def _imread_u8(path: str) -> np.ndarray:
    from .pngio import read_rgb_u8            # round 5: plain 8-bit PNG files bypass PIL's decoder (4x faster; pngio.read_rgb_u8)
    return read_rgb_u8(path)


def _imread_raw(path: str):
    from .pngio import read_rgb_raw           # round 6: the worker stops after the inflate, the scanlines are reconstructed on the device
    return read_rgb_raw(path)


UNFILTER_DTYPE = np.dtype([("raw_off", "<i8"), ("out_off", "<i8"), ("h", "<i4"), ("w", "<i4"), ("c", "<i4"), ("grey_out", "<i4")], align=True)
assert UNFILTER_DTYPE.itemsize == 32          # csrc/prep_kernels.h UnfilterItem
UNFILTER_MAX_ROWS = 256                       # csrc/prep_kernels.h kUnfilterMaxRows: one workgroup, one thread per row


def _tri_table(tri, zs: Sequence[np.ndarray]) -> np.ndarray:
    """[ntri, 18] float64: three edge functions (normalised barycentrics l_i = A_i x + B_i y + C_i) + up to three channels of
    plane coefficients (a, b, c), the latter from matplotlib's own `calculate_plane_coefficients` (what LinearTriInterpolator uses)."""
    t = tri.triangles
    x, y = np.asarray(tri.x, np.float64), np.asarray(tri.y, np.float64)
    x0, y0, x1, y1, x2, y2 = x[t[:, 0]], y[t[:, 0]], x[t[:, 1]], y[t[:, 1]], x[t[:, 2]], y[t[:, 2]]
    d = (y1 - y2) * (x0 - x2) + (x2 - x1) * (y0 - y2)
    a0, b0 = (y1 - y2) / d, (x2 - x1) / d
    a1, b1 = (y2 - y0) / d, (x0 - x2) / d
    c0, c1 = -(a0 * x2 + b0 * y2), -(a1 * x2 + b1 * y2)
    out = np.zeros((t.shape[0], TRI_DOUBLES), np.float64)
    out[:, 0:9] = np.stack([a0, b0, c0, a1, b1, c1, -(a0 + a1), -(b0 + b1), 1.0 - c0 - c1], axis=1)
    for k, z in enumerate(zs):
        out[:, 9 + 3 * k:12 + 3 * k] = tri.calculate_plane_coefficients(np.asarray(z, np.float64))
    if t.shape[0] > MAX_TRI:
        raise ValueError("a mesh has %d triangles (> %d)" % (t.shape[0], MAX_TRI))
    return out


_REF_TRI: dict = {}      # the Delaunay mesh of the canonical landmarks (one per process: matplotlib / qhull take ~0.4 ms to rebuild it per item)


def meshes(lm: np.ndarray) -> List[np.ndarray]:
    """The four triangle tables of one set of normalised landmarks, with the reference's vertex sets and dtype flow
    (dataset.generate_uv_map / generate_offset_map / generate_face_region = warp.py:194-232, utils.py:255-276)."""
    import matplotlib.tri as mtri
    uv, lm_ref = D._face_model()
    tabs = [_tri_table(mtri.Triangulation(lm[:, 0], lm[:, 1]), [uv[:, 1], uv[:, 0], uv[:, 2]])]          # stacked [y, x, z] (warp.py:228-230)
    for k, (source, target) in enumerate(((lm, lm_ref), (lm_ref, lm))):                                   # reg_in, reg_out
        s = np.concatenate([source, D._ANCHORS], axis=0).astype(np.float32)
        t = np.concatenate([target, D._ANCHORS], axis=0).astype(np.float32)
        off = s - t
        if k == 0:          # reg_in triangulates the TARGET points = the canonical landmarks + anchors: the same mesh for every item
            tri = _REF_TRI.get("t")
            if tri is None or not np.array_equal(_REF_TRI["pts"], t):
                tri = mtri.Triangulation(t[:, 0], t[:, 1])
                _REF_TRI["t"], _REF_TRI["pts"] = tri, t.copy()
        else:
            tri = mtri.Triangulation(t[:, 0], t[:, 1])
        tabs.append(_tri_table(tri, [off[:, 1], off[:, 0]]))                                              # [my, mx] (warp.py:210-213)
    more = np.copy(lm[0:17, :])
    more[:, 1] = more[0, 1] - (more[:, 1] - more[0, 1]) * 0.8
    src = np.concatenate([lm, more], axis=0)
    tabs.append(_tri_table(mtri.Triangulation(src[:, 0], src[:, 1]), [src[:, 0]]))                         # hull: interpolated x > 0
    return tabs


def crop_box(lm0: np.ndarray) -> Tuple[List[int], np.ndarray]:
    """Crop box and normalised landmarks of dataset.face_crop_and_resize (aug=False), without touching pixels."""
    lm = np.array(lm0, np.float32)
    two = np.float32(2)
    center = [(lm[:, 0].min() + lm[:, 0].max()) / two, (lm[:, 1].min() + lm[:, 1].max()) / two]
    length = float(max((lm[:, 0].max() - lm[:, 0].min()) / two, (lm[:, 1].max() - lm[:, 1].min()) / two)) * 1.4
    box = [int(center[0]) - int(length), int(center[1]) - int(length * 1.2),
           int(center[0]) + int(length), int(center[1]) + int(length) + int(length) - int(length * 1.2)]
    lm[:, 0] = lm[:, 0] - np.float32(box[0])
    lm[:, 1] = lm[:, 1] - np.float32(box[1])
    return box, lm / np.float32(length * 2)


MASK_ORDER = ("face_hair", "face", "mouth", "nose", "eyebrow", "eye", "glasses")      # train_test_GSC.py:386-392 (= the keys of ucb_post.MASK_DIRS)


def read_masks_u8(paths) -> np.ndarray:
    """The seven mask images of one item as grey levels, [7,S,S] uint8 in MASK_ORDER: cv2.imread(...) of the reference (:386-393) returns
    three equal channels of exactly these values; the / 255.0 happens on the device.  (Lives here, not in ucb_post_gpu: the loaders'
    worker processes call it and must not pay for an `import torch`.)"""
    from .pngio import read_grey_u8
    return np.stack([read_grey_u8(paths[k]) for k in MASK_ORDER], axis=0)


def _masks_raw(paths):
    """("raw8", the seven masks' inflated, still filtered scanlines back to back, S) when every mask is a plain 8-bit grey S x S PNG
    the device kernel takes (bsr_png_unfilter, grey output), else None: the worker then neither reconstructs, compares nor packs."""
    from . import pngio
    raws, S = [], None
    try:
        pngio._host_lib()
        for k in MASK_ORDER:
            with open(paths[k], "rb") as f:
                w, h, c, raw = pngio._parse_8bit(f.read())
            if c != 1 or w != h or h > UNFILTER_MAX_ROWS or w < 4 or (S is not None and h != S):
                return None
            S = h
            raws.append(raw)
    except (ValueError, TypeError, KeyError, OSError, __import__("struct").error, __import__("zlib").error):
        return None
    return ("raw8", np.concatenate(raws), S)


def masks_from_raw(packed: tuple) -> tuple:
    """A "raw8" record decoded on the host into the form pack_masks gives without `raw` (an item that went through the pipe after all)."""
    from .pngio import unfilter_host
    _, raw, S = packed
    n = S * (1 + S)
    return _pack_levels(np.stack([unfilter_host(raw[i * n:(i + 1) * n], S, S, 1)[:, :, 0] for i in range(7)], axis=0))


def pack_masks(paths, raw: bool = False) -> tuple:
    """The seven UCB segmentation masks of one item (dict in MASK_ORDER -> path) as grey levels, for the device post-processing
    (ucb_post_gpu): ("bits", [7, S*S/8] uint8) when every level is 0 or 255 — what the reference's masks are; an eighth of the bytes
    through the worker's pipe — else ("u8", [7,S,S] uint8).  raw = True (the ring path with the reconstruction on the device, round 6):
    ("raw8", ...) of _masks_raw where the files allow it."""
    if raw:
        r = _masks_raw(paths)
        if r is not None:
            return r
    m = read_masks_u8(paths)
    return _pack_levels(m)


def _pack_levels(m: np.ndarray) -> tuple:
    if m.shape[1] * m.shape[2] % 8 == 0 and bool(np.all((m == 0) | (m == 255))):
        return ("bits", np.packbits((m != 0).reshape(7, -1), axis=1), m.shape[1])
    return ("u8", m, m.shape[1])


def unpack_masks(packed: Sequence[tuple], device):
    """[pack_masks(...)] of a batch -> uint8 [B,7,S,S] grey levels on `device` (bit-packed items are expanded there)."""
    import torch
    S = packed[0][2]
    if any(p[0].startswith("dev_") for p in packed):           # ring items: the bytes are already in HBM (views of the batch's blob)
        shifts = torch.arange(7, -1, -1, device=device, dtype=torch.uint8)

        def full(p):
            t = p[1] if p[0].startswith("dev_") else torch.from_numpy(p[1]).to(device, non_blocking=True)
            return (((t[..., None] >> shifts) & 1) * 255).to(torch.uint8).reshape(7, S, S) if p[0].endswith("bits") else t.reshape(7, S, S)
        if all(p[0] == "dev_bits" for p in packed):
            return (((torch.stack([p[1] for p in packed], dim=0)[..., None] >> shifts) & 1) * 255).to(torch.uint8).reshape(len(packed), 7, S, S)
        return torch.stack([full(p) for p in packed], dim=0)
    if all(p[0] == "bits" for p in packed):
        bits = torch.from_numpy(np.stack([p[1] for p in packed], axis=0)).to(device, non_blocking=True)          # [B,7,S*S/8]
        shifts = torch.arange(7, -1, -1, device=bits.device, dtype=torch.uint8)
        return (((bits[..., None] >> shifts) & 1) * 255).to(torch.uint8).reshape(len(packed), 7, S, S)
    full = [np.unpackbits(p[1], axis=1).reshape(7, S, S) * np.uint8(255) if p[0] == "bits" else p[1] for p in packed]
    return torch.from_numpy(np.stack(full, axis=0)).to(device, non_blocking=True)


def host_part(job, raw: bool = False):
    """(lm_path, gt_path, size[, mask paths]) -> the host half of one row: (img u8, gt u8 | None, box, [4 triangle tables], name[, packed masks]).
    raw = True (the ring path, round 6): img / gt may be pngio.RawScanlines — inflated, still filtered; the device reconstructs them."""
    masks = None
    if len(job) > 3:
        masks = pack_masks(job[3], raw=raw)
        job = job[:3]
    lm_path, gt_path, size = job
    img_path = os.path.splitext(lm_path)[0] + ".png"
    read = _imread_raw if raw else _imread_u8
    img = read(img_path)
    gt = read(gt_path) if gt_path else None
    if gt is not None and gt.shape != img.shape:
        raise ValueError("ground truth %s and image %s differ in size" % (gt_path, img_path))
    box, lm = crop_box(np.load(lm_path))
    out = (img, gt, np.asarray(box, np.int32), meshes(lm), (gt_path or img_path).encode())
    return out + (masks,) if masks is not None else out


RING_CAP = 1 << 20        # bytes of one slot of the loaders' shared-memory ring (a 256x256 UCB item with ground truth, tables and masks: ~0.55 MB)
_RING_VIEWS: dict = {}


def _is_ring(part) -> bool:
    return isinstance(part[0], str)


def host_part_ring(job, ring):
    """`host_part(job)` written INTO slot `slot` of the shared-memory ring the parent page-locked (SlotRing) instead of pickled through
    the worker's pipe: -> ("ring", slot, (h, w), has_gt, image offsets, table offsets, table lengths, box, name, mask record | None,
    bytes used) — a few hundred bytes.  The loop's own thread then neither reads, unpickles nor repacks the ~0.5 MB of an item (0.1 ms
    per item of the one thread every batch goes through): the slot goes to the device as it lies, by one copy per batch.  An item that
    does not fit a slot comes back the old way.  Round 6: images that are plain 8-bit PNG files lie in the slot as their inflated,
    still FILTERED scanlines (the last tuple element: channels per image, 0 = decoded RGB) and are reconstructed on the device
    (bsr_png_unfilter) when the job's ring tuple says so (its 4th element: dataset.Dataset.device_unfilter)."""
    path, slot, cap = ring[:3]
    use_raw = bool(ring[3]) if len(ring) > 3 else False       # dataset.Dataset.device_unfilter decides
    part = host_part(job, raw=use_raw)
    img, gt, box, tabs, name = part[:5]
    # the device kernel takes images of at most UNFILTER_MAX_ROWS rows whose scanlines hold at least one dword; anything else is decoded here
    fits = lambda a: not hasattr(a, "raw") or (a.h <= UNFILTER_MAX_ROWS and a.w * a.c >= 4)
    if not (fits(img) and (gt is None or fits(gt))):
        img, gt = (img.decode() if hasattr(img, "raw") else img), (gt.decode() if gt is not None and hasattr(gt, "raw") else gt)
        if masks is not None and masks[0] == "raw8":
            masks = masks_from_raw(masks)
    masks = part[5] if len(part) > 5 else None
    rawc = tuple(int(getattr(a, "c", 0)) if hasattr(a, "raw") else 0 for a in ([img] + ([gt] if gt is not None else [])))
    arrays = [getattr(a, "raw", a) for a in ([img] + ([gt] if gt is not None else []))] + list(tabs) + ([masks[1]] if masks is not None else [])
    offs, off = [], 0
    for a in arrays:
        offs.append(off)
        off = (off + a.nbytes + 7) & ~7
    if off > cap:                                  # through the pipe after all: decoded here
        dec = lambda a: a.decode() if hasattr(a, "raw") else a
        rest = tuple(part[2:5]) + ((masks_from_raw(masks) if masks[0] == "raw8" else masks,) if masks is not None else ())
        return (dec(img), dec(gt) if gt is not None else None) + rest
    view = _RING_VIEWS.get(path)
    if view is None:
        view = _RING_VIEWS[path] = np.memmap(path, np.uint8, "r+")
    base = int(slot) * int(cap)
    for o, a in zip(offs, arrays):
        raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        view[base + o:base + o + raw.size] = raw
    k = 2 if gt is not None else 1
    mrec = None if masks is None else (masks[0], int(masks[2]), offs[k + 4], int(masks[1].nbytes))
    return ("ring", int(slot), (int(img.shape[0]), int(img.shape[1])), gt is not None, tuple(offs[:k]), tuple(offs[k:k + 4]),
            tuple(int(t.shape[0]) for t in tabs), np.asarray(box, np.int32), name, mrec, off, rawc)


class SlotRing:
    """Parent side of the ring: `nslots` x `cap` bytes of shared memory (a file under /dev/shm, unlinked as soon as every worker has
    mapped it), page-locked with hipHostRegister so that the copy engine reads the workers' bytes where they wrote them."""

    def __init__(self, nslots: int, cap: int = RING_CAP):
        import mmap
        import tempfile
        import torch
        self.nslots, self.cap = int(nslots), int(cap)
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        if d is not None:
            # a tmpfs smaller than the ring would let ftruncate succeed and kill a worker with SIGBUS at its first write beyond the limit
            # (containers often mount 64 MB there): the ring is only built where twice its size is free
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < 2 * self.nslots * self.cap:
                raise OSError("/dev/shm has %d MB free, the loader ring needs %d MB" % (st.f_bavail * st.f_frsize >> 20, self.nslots * self.cap >> 20))
        fd, self.path = tempfile.mkstemp(prefix="bsr_ring_%d_" % os.getpid(), dir=d)
        try:
            os.ftruncate(fd, self.nslots * self.cap)
            self._mm = mmap.mmap(fd, self.nslots * self.cap)
        finally:
            os.close(fd)
        self.tensor = torch.frombuffer(self._mm, dtype=torch.uint8)
        self._torch = torch
        self.pinned = False
        if torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.tensor.data_ptr(), self.nslots * self.cap, 0)
            self.pinned = int(rc) == 0 and bool(self.tensor.is_pinned())

    def unlink(self) -> None:
        path, self.path = self.path, None
        if path:
            try:
                os.unlink(path)
            except OSError:
                pass

    def close(self) -> None:
        self.unlink()
        t, self.tensor = self.tensor, None
        if t is not None and self.pinned:
            try:
                self._torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
            except Exception:
                pass
        self.pinned = False
        del t
        mm, self._mm = getattr(self, "_mm", None), None
        if mm is not None:
            try:
                mm.close()                       # refused (BufferError) while a view of the mapping is still alive somewhere: the GC takes it then
            except (BufferError, ValueError):
                pass


def _layout(parts, size: int):
    """The blob of a batch of `host_part` results that all came through the pipe: see _layout_ex."""
    total, rows_off, grid_off, pieces, _, cells, _ = _layout_ex(parts, size, RING_CAP)
    if cells:
        raise ValueError("_layout: ring items need DevicePrep.rows_ex")
    return total, rows_off, grid_off, pieces


def _layout_ex(parts, size: int, cap: int):
    """Offsets of every section of the blob (all 8-byte aligned): -> (total bytes, rows_off, grid_off, [(offset, array)], head bytes,
    ring cells, (unfilter table offset, records)).  Items that came through the pipe are packed behind the records (the `head`, staged
    by the caller); every ring item gets one `cap`-byte cell behind the head, in batch order — cells = [(part index, slot, cell
    offset)] — which the caller fills with the slot's bytes.  Ring images that lie in their slot as filtered scanlines (round 6) get
    a record in the unfilter table (in the head) and an output area behind the cells, where the row records then point."""
    B = len(parts)
    rows = np.zeros(B, ROW_DTYPE)
    pieces = []
    off = 0

    def take(nbytes: int) -> int:
        nonlocal off
        o = off
        off += (nbytes + 7) & ~7
        return o
    rows_off = take(B * ROW_DTYPE.itemsize)
    grid_off = take(size * 8)
    pieces.append((grid_off, np.linspace(0, 1, size).astype("<f8")))
    for i, part in enumerate(parts):
        if _is_ring(part):
            continue
        img, gt, box, tabs = part[:4]
        r = rows[i]
        r["h"], r["w"] = img.shape[0], img.shape[1]
        r["img_off"] = take(img.nbytes)
        pieces.append((int(r["img_off"]), img))
        if gt is not None:
            r["gt_off"] = take(gt.nbytes)
            pieces.append((int(r["gt_off"]), gt))
        else:
            r["gt_off"] = r["img_off"]
        r["box"] = box
        for m, t in enumerate(tabs):
            r["tri_off"][m] = take(t.nbytes)
            r["ntri"][m] = t.shape[0]
            pieces.append((int(r["tri_off"][m]), t))
    pieces.append((rows_off, rows))
    ring_idx = [i for i, part in enumerate(parts) if _is_ring(part)]
    n_unf = sum(sum(1 for c in (parts[i][11] if len(parts[i]) > 11 else ()) if c) + (7 if parts[i][9] is not None and parts[i][9][0] == "raw8" else 0)
                for i in ring_idx)
    unf = np.zeros(n_unf, UNFILTER_DTYPE)
    unf_off = take(max(n_unf, 1) * UNFILTER_DTYPE.itemsize)
    if n_unf:
        pieces.append((unf_off, unf))
    head, cells, mask_out = off, [], {}
    if ring_idx:
        # the records of the ring items in whole columns (per-field assignments on a structured array cost ~15 us each: 0.3 ms per batch
        # of the loop's own thread when done item by item)
        rp = [parts[i] for i in ring_idx]
        n = len(rp)
        bases = off + cap * np.arange(n, dtype=np.int64)
        off += cap * n
        hw = np.array([p[2] for p in rp], np.int64).reshape(n, 2)
        has_gt = np.array([p[3] for p in rp], bool)
        io0 = np.array([p[4][0] for p in rp], np.int64)
        io1 = np.where(has_gt, np.array([p[4][-1] for p in rp], np.int64), io0)
        toff = np.array([p[5] for p in rp], np.int64).reshape(n, 4)
        ntri = np.array([p[6] for p in rp], np.int64).reshape(n, 4)
        used = np.array([p[10] for p in rp], np.int64)
        rawc = np.array([(tuple(p[11]) + (0, 0))[:2] if len(p) > 11 else (0, 0) for p in rp], np.int64).reshape(n, 2)
        rawc[:, 1] = np.where(has_gt, rawc[:, 1], rawc[:, 0])
        if ((rawc != 0) & (rawc != 1) & (rawc != 3) & (rawc != 4)).any():
            raise ValueError("prep blob: a ring item names %s channels per filtered pixel" % sorted(set(rawc.reshape(-1).tolist())))
        if ((rawc > 0) & ((hw[:, :1] > UNFILTER_MAX_ROWS) | (hw[:, 1:2] * rawc < 4))).any():
            raise ValueError("prep blob: a filtered ring image has more than %d rows or scanlines under 4 bytes" % UNFILTER_MAX_ROWS)
        nb = np.where(rawc > 0, hw[:, :1] * (1 + hw[:, 1:2] * rawc), hw[:, :1] * hw[:, 1:2] * 3)      # bytes of each image as it lies in the slot
        npx = hw[:, 0] * hw[:, 1] * 3
        ends = np.maximum(np.maximum(io0 + nb[:, 0], io1 + nb[:, 1]), (toff + ntri * (TRI_DOUBLES * 8)).max(axis=1))
        bad = (used > cap) | (np.minimum(np.minimum(io0, io1), toff.min(axis=1)) < 0) | (ends > cap) | (hw.min(axis=1) < 0) | (ntri.min(axis=1) < 0)
        if bad.any():
            raise ValueError("prep blob: ring item %d points outside its slot (%d-byte slots)" % (ring_idx[int(np.argmax(bad))], cap))
        sel = np.array(ring_idx)
        rows["h"][sel], rows["w"][sel] = hw[:, 0], hw[:, 1]
        img_at, gt_at = bases + io0, bases + io1
        if n_unf:
            # filtered images: reconstructed into their own area behind the cells (bsr_png_unfilter), the row record points there.  Whole
            # columns again (a Python loop over the images of a batch of 32 cost the loop's thread ~1 ms per batch)
            sel01 = np.stack([rawc[:, 0] > 0, has_gt & (rawc[:, 1] > 0)], axis=1).reshape(-1)
            jj, ww = np.repeat(np.arange(n), 2)[sel01], np.tile(np.array([0, 1]), n)[sel01]
            sizes = (npx[jj] + 7) & ~7
            outs = off + np.cumsum(sizes) - sizes
            off += int(sizes.sum())
            ni = len(jj)
            unf["raw_off"][:ni] = np.where(ww == 0, img_at[jj], gt_at[jj])
            unf["out_off"][:ni], unf["h"][:ni], unf["w"][:ni], unf["c"][:ni] = outs, hw[jj, 0], hw[jj, 1], rawc[jj, ww]
            img_at[jj[ww == 0]] = outs[ww == 0]
            gt_at[jj[ww == 1]] = outs[ww == 1]
            gt_at = np.where(has_gt, gt_at, img_at)
            # the seven masks of an item that lie in its slot as filtered scanlines ("raw8"): seven records, grey output, one area
            mj = [j for j in range(n) if rp[j][9] is not None and rp[j][9][0] == "raw8"]
            if mj:
                mj = np.array(mj)
                mS = np.array([rp[j][9][1] for j in mj], np.int64)
                moff = np.array([rp[j][9][2] for j in mj], np.int64)
                mlen = np.array([rp[j][9][3] for j in mj], np.int64)
                if ((mS < 4) | (mS > UNFILTER_MAX_ROWS) | (mlen != 7 * mS * (1 + mS)) | (moff < 0) | (moff + mlen > cap)).any():
                    raise ValueError("prep blob: a ring item's filtered masks do not fit its slot")
                area = (7 * mS * mS + 7) & ~7
                mout = off + np.cumsum(area) - area
                off += int(area.sum())
                seven = np.arange(7, dtype=np.int64)[None, :]
                sl = slice(ni, ni + 7 * len(mj))
                unf["raw_off"][sl] = (bases[mj][:, None] + moff[:, None] + seven * (mS * (1 + mS))[:, None]).reshape(-1)
                unf["out_off"][sl] = (mout[:, None] + seven * (mS * mS)[:, None]).reshape(-1)
                unf["h"][sl] = unf["w"][sl] = np.repeat(mS, 7)
                unf["c"][sl], unf["grey_out"][sl] = 1, 1
                mask_out = {ring_idx[int(j)]: (int(o), int(S_)) for j, o, S_ in zip(mj, mout, mS)}
        rows["img_off"][sel] = img_at
        rows["gt_off"][sel] = gt_at
        rows["box"][sel] = np.stack([np.asarray(p[7], np.int32).reshape(4) for p in rp])
        rows["tri_off"][sel] = bases[:, None] + toff
        rows["ntri"][sel] = ntri
        cells = [(i, int(p[1]), int(bs)) for i, p, bs in zip(ring_idx, rp, bases)]
    # the kernel dereferences these offsets on the device without bounds information: every record is checked against the blob here
    h64, w64 = rows["h"].astype(np.int64), rows["w"].astype(np.int64)
    ends = np.maximum(np.maximum(rows["img_off"], rows["gt_off"]) + h64 * w64 * 3, (rows["tri_off"] + rows["ntri"].astype(np.int64) * (TRI_DOUBLES * 8)).max(axis=1))
    lows = np.minimum(np.minimum(rows["img_off"], rows["gt_off"]), rows["tri_off"].min(axis=1))
    bad = (lows < 0) | (ends > off) | (rows["ntri"].max(axis=1) > MAX_TRI) | (rows["ntri"].min(axis=1) < 0) | (h64 < 0) | (w64 < 0)
    if bad.any():
        raise ValueError("prep blob: row %d points outside the %d-byte blob" % (int(np.argmax(bad)), off))
    return off, rows_off, grid_off, pieces, head, cells, (unf_off, n_unf, mask_out)


def pack_into(buf: np.ndarray, pieces) -> None:
    """Copy every section into `buf` (a uint8 view of the staging memory)."""
    for off, arr in pieces:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        buf[off:off + raw.size] = raw


def pack_batch(parts, size: int):
    """One blob for bsr_prep_rows: [row records | grid | images | triangle tables], every section 8-byte aligned."""
    total, rows_off, grid_off, pieces = _layout(parts, size)
    buf = np.zeros(total, np.uint8)
    pack_into(buf, pieces)
    return buf.tobytes(), rows_off, grid_off


class DevicePrep:
    """`rows(parts)` -> packed `[B,S,S,16]` float32 CUDA tensor (+ the crop boxes) for a list of `host_part` results."""

    def __init__(self, device: int = 0, size: int = 256):
        import torch
        from . import _lib
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePrep needs a ROCm GPU: the host path is blindshadowremoval_amd.dataset.build_row")
        self._torch, self._lib, self._check = torch, _lib.load(), _lib.check
        self.device, self.size = int(device), int(size)
        self._stage, self._copied = [None, None], [None, None]

    def warm(self, nbytes: int = 8 << 20) -> None:
        """Page-lock the two staging buffers now (~60 ms each) instead of inside the first two batches of a timed loop."""
        for k in (0, 1):
            if self._stage[k] is None or self._stage[k].numel() < nbytes:
                self._stage[k] = self._torch.empty(nbytes, dtype=self._torch.uint8).pin_memory()

    def rows(self, parts):
        out, boxes, _, _ = self.rows_ex(parts)
        return out, boxes

    def rows_ex(self, parts):
        """-> (rows [B,S,S,16] float32 on the device, boxes [B,4], per item its masks as device views | None, per item its name).
        `parts`: `host_part` results (pickled through a worker's pipe) and / or `host_part_ring` records (the bytes lie in self.ring)."""
        torch = self._torch
        B, S = len(parts), self.size
        ring = getattr(self, "ring", None)
        cap = ring.cap if ring is not None else RING_CAP
        total, rows_off, grid_off, pieces, head, cells, (unf_off, n_unf, mask_out) = _layout_ex(parts, S, cap)
        if cells and ring is None:
            raise RuntimeError("DevicePrep.rows_ex: ring records without a ring")
        dev = "cuda:%d" % self.device
        with torch.cuda.device(self.device):
            # two pinned staging buffers used alternately: the sections are copied straight into page-locked memory and go to the
            # device in one asynchronous copy; a buffer is reused only after the copy that read it has finished
            k = self._turn = (getattr(self, "_turn", 0) + 1) & 1
            stage = self._stage[k]
            if stage is None or stage.numel() < head:
                stage = self._stage[k] = torch.empty(max(head, 1 << 22) * 5 // 4, dtype=torch.uint8).pin_memory()
            if self._copied[k] is not None:
                self._copied[k].synchronize()
            pack_into(stage.numpy(), pieces)
            # the host-to-device copies run on their OWN stream: the copy engine moves batch k + 1 while the compute stream is still in
            # batch k's forward (on one stream the ~16 MB of a batch sat between two forwards: ~0.4 ms of a 4 ms step)
            main = torch.cuda.current_stream()
            if getattr(self, "_h2d", None) is None:
                self._h2d = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._h2d):
                d_blob = torch.empty(total, dtype=torch.uint8, device=dev)
                d_blob[:head].copy_(stage[:head], non_blocking=True)
                # ring items: consecutive slots of consecutive cells go in ONE copy (the usual case: the whole batch), whole slots as they lie
                c = 0
                while c < len(cells):
                    e = c + 1
                    while e < len(cells) and cells[e][1] == cells[e - 1][1] + 1:
                        e += 1
                    src = ring.tensor[cells[c][1] * cap:(cells[e - 1][1] + 1) * cap]
                    d_blob[cells[c][2]:cells[c][2] + (e - c) * cap].copy_(src, non_blocking=True)
                    c = e
                ev = self._copied[k] = self.last_copy = torch.cuda.Event()
                ev.record()
                # the preparation kernel follows its input on the same side stream (BSR_PREP_SIDE=0: on the compute stream): a short
                # bandwidth-bound kernel that shares the chip with the previous batch's forward instead of standing in line behind it
                side = os.environ.get("BSR_PREP_SIDE", "1") != "0"
                if side:
                    out = torch.empty((B, S, S, 16), dtype=torch.float32, device=dev)
                    tmp = torch.empty((B, S, S), dtype=torch.float32, device=dev)
                    if n_unf:                      # the filtered images of the ring items become RGB8 where their row records point
                        self._check(self._lib.bsr_png_unfilter(self.device, d_blob.data_ptr(), total, unf_off, n_unf, self._h2d.cuda_stream), "bsr_png_unfilter")
                    rc = self._lib.bsr_prep_rows(self.device, d_blob.data_ptr(), total, rows_off, grid_off, B, S, out.data_ptr(), tmp.data_ptr(),
                                                 self._h2d.cuda_stream)
                    done = torch.cuda.Event()
                    done.record()
            if side:
                main.wait_event(done)
                out.record_stream(main)
            else:
                main.wait_event(ev)
            d_blob.record_stream(main)             # allocated on the side stream, read by the compute stream (the mask views; the kernel when it runs there)
            if not side:
                out = torch.empty((B, S, S, 16), dtype=torch.float32, device=dev)
                tmp = torch.empty((B, S, S), dtype=torch.float32, device=dev)
                if n_unf:
                    self._check(self._lib.bsr_png_unfilter(self.device, d_blob.data_ptr(), total, unf_off, n_unf, main.cuda_stream), "bsr_png_unfilter")
                rc = self._lib.bsr_prep_rows(self.device, d_blob.data_ptr(), total, rows_off, grid_off, B, S, out.data_ptr(), tmp.data_ptr(),
                                             main.cuda_stream)
        self._check(rc, "bsr_prep_rows")
        boxes = np.stack([np.asarray(p[7] if _is_ring(p) else p[2], np.float32) for p in parts], axis=0)
        names = [p[8] if _is_ring(p) else p[4] for p in parts]
        masks = [(p[5] if len(p) > 5 else None) if not _is_ring(p) else None for p in parts]
        for i, _, base in cells:
            m = parts[i][9]
            if m is not None:
                kind, ms, moff, nbytes = m
                if kind == "raw8":                 # reconstructed by bsr_png_unfilter into their own area: grey levels [7,S,S]
                    o = mask_out[i][0]
                    masks[i] = ("dev_u8", d_blob[o:o + 7 * ms * ms].view(7, ms, ms), ms)
                    continue
                v = d_blob[base + moff:base + moff + nbytes]
                masks[i] = ("dev_" + kind, v.view(7, -1) if kind == "bits" else v.view(7, ms, ms), ms)
        return out, boxes, masks, names
