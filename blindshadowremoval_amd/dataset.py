"""Host-side input preparation — counterpart of the reference's test-time loaders
(`Dataset(config, 'test')`, `parse_fn_test_FFHQ`, `parse_fn_test`: /root/reference/dataset.py:18-72, 148-302, 616-770;
helpers `face_crop_and_resize`, `generate_face_region`: /root/reference/utils.py:356-433, 255-276;
`generate_uv_map`, `generate_offset_map`: /root/reference/warp.py:194-232).

`Dataset(config, 'test', dset='sfw' | 'sfw_video')` is the counterpart of the TSM script's loaders
(/root/reference/dataset_with_TSM.py:19-79, 225-287, 289-583): elements `[1,2,256,256,17]` (frame + mirror) / `[1,10,256,256,13]`
(ten frames of a video) for `FSRNetTSM.testsfw` / `testsfw_video`.

It yields what `FSRNet.testFFHQ` / `test` consume: `.name_list` and `.feed`, an iterator of
`(img[1,R,256,256,16], box[1,4], name)` with channel layout [img3, gt3, uvm3, reg_in3, reg_out3, face1]
(SURVEY.md Appendix D).  No TensorFlow / OpenCV: PNGs are decoded with PIL and the three OpenCV calls on the path are
restated with OpenCV's documented semantics (INTER_LINEAR resize; 5x5 Gaussian with sigma 0 = [1,4,6,4,1]/16,
BORDER_REFLECT_101).  Delaunay interpolation uses matplotlib.tri exactly as the reference does.

The reference stacks 10 rows per element (row 0 = the image, rows 1-9 = random same-folder images) and consumes only
row 0 (utils.py:231); rows are independent at inference, so `rows=1` is the default and `rows=10` reproduces the
reference layout (siblings drawn with a seeded RNG instead of the reference's unseeded `random`).
"""
from __future__ import annotations

import glob
import os
import random
import re
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "face_model.npz")
_ANCHORS = np.asarray([[0, 0], [0, 255], [255, 0], [255, 255], [0, 127], [127, 0], [255, 127], [127, 255],
                       [0, 63], [0, 191], [255, 63], [255, 191], [63, 0], [191, 0], [63, 255], [191, 255]]) / 255   # warp.py:195-198


_FACE_MODEL: list = []


def _face_model():
    """(uv, lm_ref) of the canonical face (data/face_model.npz), read once per process — the loaders' workers called it per item
    (0.5 ms of npz parsing each); the arrays are read-only."""
    if not _FACE_MODEL:
        z = np.load(_DATA)
        uv, lm_ref = z["uv"], z["lm_ref"]
        uv.setflags(write=False)
        lm_ref.setflags(write=False)
        _FACE_MODEL.append((uv, lm_ref))
    return _FACE_MODEL[0]


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, further limited by a cgroup CPU quota (a container that SEES 256 CPUs
    but is throttled to 16 runs slower, not faster, with 64 worker processes — measured on the GPU box)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),                                   # cgroup v2: "<quota> <period>" | "max <period>"
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):            # cgroup v1
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if quota in ("max", "-1"):
                continue
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            q = int(quota) / int(period)
            if q > 0:
                n = min(n, max(1, int(q + 0.5)))
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_share() -> int:
    """CPUs ONE rank of a node may plan its worker pools on: usable_cpus() divided by the ranks sharing the node (torchrun's
    LOCAL_WORLD_SIZE; 1 outside a launcher) — eight ranks each sizing their loader / PNG / post-processing pools for the whole quota
    would oversubscribe it eightfold."""
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    return max(1, usable_cpus() // local_world)


def natural_key(s: str):
    """natsort-style key: digit runs compare numerically (dataset.py:47,57 use natsorted)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def imread_rgb(path: str) -> np.ndarray:
    """cv2.cvtColor(cv2.imread(p), COLOR_BGR2RGB) / 255. (dataset.py:627)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), np.float64) / 255.0


def resize_linear(img: np.ndarray, size: int) -> np.ndarray:
    """cv2.resize(img, (size, size)) with the default INTER_LINEAR: half-pixel centres, edge clamp, no antialias
    (source coordinate max((o + 0.5) * n / size - 0.5, 0), neighbours clamped to n - 1).  Pure numpy so that the loader's worker
    processes never import torch."""
    img = np.ascontiguousarray(img, np.float64)

    def axis(n):
        src = np.maximum((np.arange(size) + 0.5) * (n / size) - 0.5, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), n - 1)
        return i0, np.minimum(i0 + 1, n - 1), src - i0
    y0, y1, wy = axis(img.shape[0])
    x0, x1, wx = axis(img.shape[1])
    wx = wx[None, :, None]
    top = img[y0][:, x0] * (1 - wx) + img[y0][:, x1] * wx
    bot = img[y1][:, x0] * (1 - wx) + img[y1][:, x1] * wx
    wy = wy[:, None, None]
    return top * (1 - wy) + bot * wy


def gaussian_blur5(a: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(a, (5,5), 0): OpenCV's fixed 5-tap kernel [1,4,6,4,1]/16, BORDER_REFLECT_101."""
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16.0
    p = np.pad(np.asarray(a, np.float64), 2, mode="reflect")
    tmp = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(5))
    return sum(k[i] * tmp[i:i + a.shape[0], :] for i in range(5))


# left/right landmark correspondence of the 68-point scheme (utils.py:360-364, 1-based there)
LM_REVERSE = np.array([17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 28, 29, 30, 31, 36, 35, 34, 33,
                       32, 46, 45, 44, 43, 48, 47, 40, 39, 38, 37, 42, 41, 55, 54, 53, 52, 51, 50, 49, 60, 59, 58, 57, 56, 65, 64, 63, 62, 61, 68,
                       67, 66], np.int32) - 1


def face_crop_and_resize(img0: np.ndarray, lm0: np.ndarray, fsize: int, with_mirror: bool = False):
    """utils.face_crop_and_resize with aug=False (utils.py:356-433): crop box from the landmark extent (x1.4, shifted up
    by 20 %), zero-extend when it leaves the image, resize to fsize; landmarks normalised by the box size.
    `with_mirror` also returns the landmarks of the horizontally flipped crop (utils.py:383-385,401-410,433), which the TSM
    loaders pair with `flip(crop)`."""
    # dtype flow mirrors the reference: landmarks stay float32, the box centre is float32 arithmetic (int() truncation of
    # e.g. 127.999998 vs 128.0 changes the crop), the half-length is promoted to float64 by the `* 1.4` (numpy-1.x
    # scalar rules, the reference's environment)
    img, lm = np.copy(img0), np.array(lm0, np.float32)
    h, w = img.shape[0], img.shape[1]
    two = np.float32(2)
    center = [(lm[:, 0].min() + lm[:, 0].max()) / two, (lm[:, 1].min() + lm[:, 1].max()) / two]
    length = float(max((lm[:, 0].max() - lm[:, 0].min()) / two, (lm[:, 1].max() - lm[:, 1].min()) / two)) * 1.4
    box = [int(center[0]) - int(length), int(center[1]) - int(length * 1.2),
           int(center[0]) + int(length), int(center[1]) + int(length) + int(length) - int(length * 1.2)]
    box0 = list(box)
    lm_m = None
    if with_mirror:
        lm_m = np.array(lm, np.float32)
        lm_m[:, 0] = np.float32(w) - lm_m[:, 0]
        lm_m = lm_m[LM_REVERSE, :]
        lm_m[:, 0] = lm_m[:, 0] - np.float32(w - box[2])          # box_m = [W - box[2], box[1], W - box[0], box[3]]
        lm_m[:, 1] = lm_m[:, 1] - np.float32(box[1])
    lm[:, 0] = lm[:, 0] - np.float32(box[0])
    lm[:, 1] = lm[:, 1] - np.float32(box[1])
    px = max(-box[0], box[2] - w) if (box[0] < 0 or box[2] > w) else 0
    py = max(-box[1], box[3] - h) if (box[1] < 0 or box[3] > h) else 0
    if px > 0 or py > 0:
        big = np.zeros((h + 2 * py + 2, w + 2 * px + 2, img.shape[2]))
        big[py:py + h, px:px + w, :] = img
        img = big
        box = [box[0] + px, box[1] + py, box[2] + px, box[3] + py]
    img = img[box[1]:box[3], box[0]:box[2], :]
    if img.shape[0] == img.shape[1] and img.shape[0] > 0:
        img = resize_linear(img, fsize)
    else:
        img = np.zeros((fsize, fsize, img.shape[2]))
    if with_mirror:
        return img, lm / np.float32(length * 2), lm_m / np.float32(length * 2), box0
    return img, lm / np.float32(length * 2), box0


def _grid(size: int):
    return np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size))


def generate_face_region(lm: np.ndarray, size: int) -> np.ndarray:
    """utils.generate_face_region (utils.py:255-276): convex hull of the landmarks + mirrored jaw line, blurred."""
    import matplotlib.tri as mtri
    more = np.copy(lm[0:17, :])
    more[:, 1] = more[0, 1] - (more[:, 1] - more[0, 1]) * 0.8
    src = np.concatenate([lm, more], axis=0)
    xi, yi = _grid(size)
    interp = mtri.LinearTriInterpolator(mtri.Triangulation(src[:, 0], src[:, 1]), src[:, 0])
    m = np.nan_to_num(np.ma.filled(interp(xi, yi), np.nan))
    return gaussian_blur5((m > 0).astype(np.float32)).astype(np.float32).reshape(size, size, 1)


def generate_uv_map(lm: np.ndarray, uv: np.ndarray, size: int) -> np.ndarray:
    """warp.generate_uv_map (warp.py:215-232): barycentric interpolation of the canonical UVZ table, 0 outside the hull."""
    import matplotlib.tri as mtri
    xi, yi = _grid(size)
    tri = mtri.Triangulation(lm[:, 0], lm[:, 1])
    ch = [np.ma.filled(mtri.LinearTriInterpolator(tri, uv[:, c])(xi, yi), np.nan) for c in (1, 0, 2)]   # stacked [y, x, z]
    return np.nan_to_num(np.stack(ch, axis=2))


def generate_offset_map(source: np.ndarray, target: np.ndarray, size: int) -> np.ndarray:
    """warp.generate_offset_map (warp.py:194-213): landmark offsets (+16 fixed anchors) interpolated over the target mesh."""
    import matplotlib.tri as mtri
    xi, yi = _grid(size)
    s = np.concatenate([source, _ANCHORS], axis=0).astype(np.float32)
    t = np.concatenate([target, _ANCHORS], axis=0).astype(np.float32)
    off = s - t
    tri = mtri.Triangulation(t[:, 0], t[:, 1])
    mx = np.ma.filled(mtri.LinearTriInterpolator(tri, off[:, 0])(xi, yi), np.nan)
    my = np.ma.filled(mtri.LinearTriInterpolator(tri, off[:, 1])(xi, yi), np.nan)
    return np.stack([my, mx, mx * 0], axis=2)


def build_row(img_path: str, lm_path: str, gt_path: Optional[str] = None, size: int = 256) -> Tuple[np.ndarray, np.ndarray]:
    """One `[size,size,16]` row + crop box, in the order of dataset.py:627-638."""
    uv, lm_ref = _face_model()
    img = imread_rgb(img_path)
    gt = imread_rgb(gt_path) if gt_path else img
    both = np.concatenate([img, gt], axis=2)
    crop, lm, box = face_crop_and_resize(both, np.load(lm_path), size)
    face = generate_face_region(lm, size)
    uvm = generate_uv_map(lm, uv, size)
    reg_in = generate_offset_map(lm, lm_ref, size)
    reg_out = generate_offset_map(lm_ref, lm, size)
    return np.concatenate([crop, uvm, reg_in, reg_out, face], axis=2).astype(np.float32), np.asarray(box, np.float32)


def _maps(lm: np.ndarray, size: int):
    """uvm, reg_in, reg_out, face of one set of normalised landmarks (dataset_with_TSM.py:256-259)."""
    uv, lm_ref = _face_model()
    return (generate_uv_map(lm, uv, size), generate_offset_map(lm, lm_ref, size), generate_offset_map(lm_ref, lm, size),
            generate_face_region(lm, size))


def imread_gray(path: str) -> np.ndarray:
    """cv2.imread(path, 0): 8-bit grey levels, NOT divided by 255 (dataset_with_TSM.py:243: label values 0 / 1 / 2)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), np.float64)


def build_sfw_pair(label_path: str, size: int = 256) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """`parse_fn_test_sfw` of /root/reference/dataset_with_TSM.py:225-287: one SFW frame + its mirror image as a group of two,
    `[2, size, size, 17]` = [img3, cmap3, mask1, uvm3, reg_in3, reg_out3, face1].  Files: `<f>_label.png` (mask), `<f>_label_cmap.png`,
    `<f>.png`, `<f>.npy`."""
    stem = label_path.rsplit(".", 1)[0]
    frame = stem[:-6]                                                   # strips "_label"
    img = imread_rgb(frame + ".png")
    cmap = imread_rgb(stem + "_cmap.png")
    mask = imread_gray(label_path)[:, :, None]
    crop, lm, lm_m, box = face_crop_and_resize(np.concatenate([img, cmap, mask], axis=2), np.load(frame + ".npy"), size, with_mirror=True)
    uvm, reg_in, reg_out, face = _maps(lm, size)
    img1 = np.concatenate([crop, uvm, reg_in, reg_out, face], axis=2)
    uvm_m, reg_in_m, reg_out_m, face_m = _maps(lm_m, size)
    img2 = np.concatenate([img1[:, ::-1, :7], uvm_m, reg_in_m, reg_out_m, face_m], axis=2)      # cv2.flip(img1[:, :, :7], 1)
    return np.stack([img1, img2], axis=0).astype(np.float32)[None], np.asarray(box, np.float32)[None], np.array([(frame + ".png").encode()])


def sfw_video_frames(frame: int) -> List[int]:
    """The ten frame numbers `parse_fn_test_sfw_video` groups with `frame` (dataset_with_TSM.py:325-382)."""
    if frame < 3:
        off = (2, 4, 6, 8, 10, 12, 14, 16, 1)
    elif frame < 5:
        off = (1, 3, 5, 7, 9, 11, 13, 15, -2)
    elif frame < 7:
        off = (1, 3, 5, 7, 9, 11, 13, -2, -4)
    elif frame < 9:
        off = (1, 3, 5, 7, 9, 11, -2, -4, -6)
    elif frame > 100:
        off = (-1, -3, -5, -7, -9, -11, -2, -4, -6)
    else:
        off = (1, 3, 5, 7, 9, -2, -4, -6, -8)
    return [frame] + [frame + o for o in off]


def build_sfw_video(label_path: str, size: int = 256) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """`parse_fn_test_sfw_video` (dataset_with_TSM.py:289-583): ten frames of one video as a group, `[10, size, size, 13]` =
    [img3, uvm3, reg_in3, reg_out3, face1] each; the box returned is the LAST frame's (the reference overwrites `box`)."""
    stem = label_path.rsplit(".", 1)[0]                                  # the reference names frames <n>.png / <n>.npy and lists <n>.png's label
    folder, first = os.path.dirname(stem), int(os.path.basename(stem).split("_")[0].split(".")[0])
    rows, box = [], None
    for f in sfw_video_frames(first):
        base = os.path.join(folder, str(f))
        if not os.path.isfile(base + ".png"):
            raise FileNotFoundError("SFW video group of frame %d needs %s.png (the reference blocks on input() here)" % (first, base))
        crop, lm, box = face_crop_and_resize(imread_rgb(base + ".png"), np.load(base + ".npy"), size)
        uvm, reg_in, reg_out, face = _maps(lm, size)
        rows.append(np.concatenate([crop, uvm, reg_in, reg_out, face], axis=2))
    return np.stack(rows, axis=0).astype(np.float32)[None], np.asarray(box, np.float32)[None], np.array([(os.path.join(folder, str(first)) + ".png").encode()])


def build_element(job) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One dataset element `(img[1,R,size,size,16], box[1,4], name[1])` from a job `(lm_path, gt_path, sibling lm paths, size)`.
    Top-level so that worker processes can run it (the rows of an element never depend on another element)."""
    lm_path, gt_path, siblings, size = job[:4]
    if isinstance(gt_path, tuple) and gt_path[0] == "<device>":            # the host half of a device-prepared row (prep.py)
        from .prep import host_part, host_part_ring
        hjob = (lm_path, gt_path[1], size) + ((gt_path[2],) if len(gt_path) > 2 else ())
        return host_part_ring(hjob, job[4]) if len(job) > 4 else host_part(hjob)            # job[4]: (ring file, slot, slot bytes)
    if gt_path == "<sfw>":
        return build_sfw_pair(lm_path, size)
    if gt_path == "<sfw_video>":
        return build_sfw_video(lm_path, size)
    img_path = os.path.splitext(lm_path)[0] + ".png"
    row0, box = build_row(img_path, lm_path, gt_path, size)
    rows = [row0]
    for sib in siblings:                                               # rows 1..9: random same-folder images (dataset.py:641-762)
        rows.append(row0 if sib == lm_path else build_row(os.path.splitext(sib)[0] + ".png", sib, gt_path or img_path, size)[0])
    name = (gt_path or img_path).encode()
    return np.stack(rows, axis=0)[None], box[None], np.array([name])


class Dataset:
    """`Dataset(config, 'test')` counterpart (dataset.py:18-72).  `ucb=True` switches to `parse_fn_test`, whose ground truth
    comes from the sibling `gt` tree (dataset.py:155).

    `workers` / `prefetch` are the counterpart of the reference pipeline's `map(parse_fn, num_parallel_calls=AUTOTUNE).batch(1)
    .prefetch(AUTOTUNE)` (dataset.py:63-72): with `workers` > 0 elements are prepared by that many worker PROCESSES (the
    Delaunay interpolations are Python / GIL-bound), up to `prefetch` elements ahead of the consumer, and delivered in
    `name_list` order with bit-identical contents to the serial path (`workers=0`, the default); `workers=-1` picks
    min(cpu_count, 16)."""

    def __init__(self, config, mode: str = "test", dset=None, ucb: bool = False, rows: int = 1, seed: int = 0,
                 workers: int = 0, prefetch: Optional[int] = None, device_prep: Optional[int] = None, device_batch: int = 16):
        if mode != "test" or dset not in (None, "sfw", "sfw_video"):
            raise NotImplementedError("only the test loaders are provided (GSC: dset=None; TSM: dset='sfw' | 'sfw_video'); training loaders are out of scope")
        self.config, self.mode, self.ucb, self.rows, self.dset = config, mode, ucb, rows, dset
        self._rng = random.Random(seed)
        self._shard = None
        if workers < 0:
            workers = min(cpu_share(), 16)
        self.workers = int(workers)
        # elements the workers may run ahead of the consumer: two per worker on the host path; four with device preparation, whose
        # consumer (the pipelined FSRNet loop) spends milliseconds at a time away from the feed
        self.prefetch = int(prefetch) if prefetch is not None else max(2, (4 if device_prep is not None else 2) * self.workers)
        self._pool = None
        # device_prep = GPU index: rows are prepared ON THE DEVICE (prep.py / csrc/prep_kernels.h) in groups of `device_batch`; the
        # workers then only decode PNGs and triangulate, and `feed` yields (img CUDA tensor [1,1,S,S,16], box[1,4], name) — the
        # same values as the host path to 1e-6 (tests/test_prep_gpu.py), never leaving HBM before the generator reads them
        self.device_prep, self.device_batch = device_prep, int(device_batch)
        if device_prep is not None and (rows != 1 or dset is not None):
            raise NotImplementedError("device_prep prepares row 0 of the GSC loaders (rows=1, dset=None)")
        self.ucb_mask_files: Optional[List[Dict[str, str]]] = None       # per item of name_list: the seven mask paths (FSRNet.test sets it; device_prep only)
        # round 6: PNG scanline reconstruction on the device (prep.host_part_ring / bsr_png_unfilter) instead of in the workers.  None =
        # where it pays: the UCB loop (two photographs and seven masks per item: the loop waits for its loader; +5-10 % measured) and not
        # the FFHQ one (one photograph per item: the GPU side is the longer one there, and the kernel is 0.17 ms per batch of it; -3 %).
        # BSR_DEVICE_UNFILTER=0 / 1 overrides.
        self.device_unfilter: Optional[bool] = None
        self.name_list: List[str] = []
        pattern = "*.npy" if dset is None else "*_label.png"               # dataset.py:55-61 | dataset_with_TSM.py:63
        for d in config.DATA_DIR_TEST:
            for folder in sorted(glob.glob(d), key=natural_key):
                self.name_list += sorted(glob.glob(os.path.join(folder, pattern)), key=natural_key)
        self.feed: Iterator = self._iterate()

    def shard(self, lo: int, hi: int) -> None:
        """Data-parallel loops (FSRNet under a process group): `feed` yields only items [lo, hi) of `name_list`, which itself stays
        the FULL list (the loops index masks and report progress by the global position).  Must be called before the first element
        is drawn.  The seeded sibling draws of `rows > 1` are still consumed for every item of the list, so an item's element is the
        same whichever rank prepares it."""
        if getattr(self, "_started", False):
            raise RuntimeError("Dataset.shard() after the first element was drawn")
        if not 0 <= lo <= hi <= len(self.name_list):
            raise ValueError("shard [%d, %d) outside the %d-item name list" % (lo, hi, len(self.name_list)))
        self._shard = (int(lo), int(hi))

    def _gt_path(self, lm_path: str) -> Optional[str]:
        if not self.ucb:
            return None
        parts = lm_path.replace("\\", "/").split("/")
        return os.path.splitext("/".join(parts[:-3] + ["gt"] + parts[-2:]))[0] + ".png"      # .../train/input/x/y -> .../train/gt/x/y

    def _jobs(self):
        """Jobs in name_list order; the sibling draws consume the seeded RNG in that order whatever the worker count."""
        size = self.config.IMG_SIZE
        self._started = True
        lo, hi = self._shard if self._shard is not None else (0, len(self.name_list))
        if self.dset is not None:                                          # TSM loaders: the element is a group of 2 / 10 coupled frames
            for label in self.name_list[lo:hi]:
                yield (label, "<" + self.dset + ">", [], size)
            return
        for i, lm_path in enumerate(self.name_list):
            if i >= hi:
                break
            sibs = []
            if self.rows > 1:
                siblings = sorted(glob.glob(os.path.join(os.path.dirname(lm_path), "*.npy")), key=natural_key)
                sibs = [siblings[self._rng.randint(0, len(siblings) - 1)] for _ in range(self.rows - 1)]
            if i < lo:
                continue                                                   # another rank's item: only its RNG draws are consumed
            gt = self._gt_path(lm_path)
            if self.device_prep is not None:
                # ucb_mask_files (set by FSRNet.test for its device post-processing): the item's seven mask PNGs are decoded by the same
                # worker that decodes its image, and travel bit-packed
                masks = self.ucb_mask_files[i] if self.ucb_mask_files is not None else None
                job = (lm_path, ("<device>", gt) + ((masks,) if masks is not None else ()), sibs, size)
                ring = getattr(self, "_ring", None)
                if ring is not None:
                    # the k-th device job of this Dataset writes slot k mod nslots; that slot last held job k - nslots, whose batch has
                    # been handed to the copy engine long ago (nslots >= prefetch + 3 batches) — its copy must have FINISHED
                    k = self._ring_seq
                    self._ring_seq += 1
                    if k >= ring.nslots:
                        while self._ring_copies and self._ring_copies[0][0] <= k - ring.nslots:       # batches that ended before that job: older copies
                            self._ring_copies.pop(0)
                        if not self._ring_copies:
                            raise RuntimeError("loader ring: slot %d is still waiting for its batch (ring of %d slots too small)" % (k % ring.nslots, ring.nslots))
                        if not self._ring_copies[0][2]:
                            self._ring_copies[0][1].synchronize()
                            self._ring_copies[0][2] = True
                    unf = self.device_unfilter if self.device_unfilter is not None else (self.ucb and masks is not None)
                    unf = {"0": False, "1": True}.get(os.environ.get("BSR_DEVICE_UNFILTER", ""), unf)
                    job = job + ((ring.path_for_workers, k % ring.nslots, ring.cap, bool(unf)),)
                yield job
            else:
                yield (lm_path, gt, sibs, size)

    def _close_pool(self) -> None:
        pool, self._pool = self._pool, None
        if pool is not None:
            pool.shutdown()

    def close(self) -> None:
        self._close_pool()
        ring, self._ring = getattr(self, "_ring", None), None
        if ring is not None:
            ring.close()

    def poll(self) -> None:
        """Non-blocking: move the results the workers have finished out of their pipes (a worker whose pipe is full waits with its
        next element).  The loops call it while they are busy with the GPU / the PNG pool."""
        if self._pool is not None:
            self._pool._pump(block=False)

    def warm(self) -> None:
        """Start the worker processes and let them import their modules now (otherwise the first elements pay for it)."""
        from .pngio import _host_lib
        _host_lib()                               # libbsr_host.so (the workers' PNG reconstruction) is built HERE, once, not by the first worker that needs it
        if self.workers > 0 and self._pool is None:
            self._pool = _SelectPool(self.workers)
            self._pool.warm("rows")
        if self.device_prep is not None and getattr(self, "_dp", None) is None:
            from .prep import DevicePrep
            self._dp = DevicePrep(self.device_prep, self.config.IMG_SIZE)
            self._dp.warm(max(8 << 20, self.device_batch * (2 * 3 * self.config.IMG_SIZE ** 2 + (64 << 10)) * 5 // 4))
        self._ensure_ring()

    def _ensure_ring(self) -> None:
        """Device preparation with worker processes: the workers write their results into a page-locked shared-memory ring
        (prep.SlotRing / host_part_ring) instead of pickling ~0.5 MB per item through their pipes.  BSR_LOADER_RING=0 keeps the pipes."""
        if (self.device_prep is None or self.workers <= 0 or getattr(self, "_ring", None) is not None or getattr(self, "_started", False)
                or os.environ.get("BSR_LOADER_RING", "1") == "0"):
            return
        from .prep import SlotRing
        if self._pool is None:
            self._pool = _SelectPool(self.workers)
        b = max(1, self.device_batch)
        try:
            ring = SlotRing(((self.prefetch + 3 * b + b - 1) // b) * b)
        except OSError as e:                      # no room in /dev/shm: the pipes it is
            self.ring_error = str(e)
            return
        if not ring.pinned:                       # registration refused: the pipes it is
            self.ring_error = "hipHostRegister refused the shared mapping"
            ring.close()
            return
        ring.path_for_workers = ring.path
        self._pool.warm("rows", ring.path)        # every worker maps the file now ...
        ring.unlink()                             # ... so its name can go: nothing is left behind whatever happens to this process
        self._ring, self._ring_seq, self._ring_copies = ring, 0, []
        if getattr(self, "_dp", None) is not None:
            self._dp.ring = ring

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _iterate(self):
        if self.device_prep is None:
            yield from self._iterate_host()
            return
        from .prep import DevicePrep
        dp = self._dp if getattr(self, "_dp", None) is not None else DevicePrep(self.device_prep, self.config.IMG_SIZE)
        self._ensure_ring()
        dp.ring = getattr(self, "_ring", None)
        group = []
        self._emitted = 0

        def emit():
            out, boxes, masks, names = dp.rows_ex(group)
            self._emitted += len(group)
            if dp.ring is not None:
                self._ring_copies.append([self._emitted, dp.last_copy, False])       # jobs < _emitted have left their slots once this event is done
            for i in range(len(group)):
                if masks[i] is not None:      # + the item's seven segmentation masks (bit-packed or grey levels, host or device: prep.pack_masks / rows_ex)
                    yield out[i:i + 1][None], boxes[i][None], np.array([names[i]]), masks[i]
                else:
                    yield out[i:i + 1][None], boxes[i][None], np.array([names[i]])
        try:
            for part in self._iterate_host():
                group.append(part)
                if len(group) >= self.device_batch:
                    yield from emit()
                    group = []
            if group:
                yield from emit()
        finally:
            if dp.ring is not None and getattr(dp, "last_copy", None) is not None:
                dp.last_copy.synchronize()          # the copy engine may still be reading the last slots
            dp.ring = None
            self.close()

    def _iterate_host(self):
        jobs = self._jobs()
        if self.workers <= 0:
            for job in jobs:
                yield build_element(job)
            return
        if self._pool is None:
            self._pool = _SelectPool(self.workers)
        try:
            yield from self._pool.imap(jobs, self.prefetch)
        finally:
            self._close_pool()              # the ring outlives the workers: the last group's slots are still to be copied (_iterate closes it)


class _JobFailed(Exception):
    """A worker's 'err' reply, parked in _SelectPool._done under the failed job's ticket until result() / imap() reaches it."""


class _SelectPool:
    """N worker PROCESSES (`python -m blindshadowremoval_amd._row_worker`: plain subprocesses over pipes — nothing is forked from a process
    that may hold a GPU context, and the workers do not re-import the parent's __main__), driven from the CALLING thread with
    non-blocking pipes and select() — no helper threads at all.  Round 2's thread-per-worker pool cost nothing at 40 elements per
    second; at a few hundred, two dozen threads waking up to read and unpickle results keep taking the GIL from the loop's own thread (measured on the GPU box: host halves alone 3 900 /s, the
    device half alone 2 300 /s, both together through the threaded pool 430 /s).  `imap(jobs, depth)` yields results in job order
    with at most `depth` jobs outstanding."""

    def __init__(self, n: int):
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env.setdefault(k, "1")
        env["HIP_VISIBLE_DEVICES"] = ""
        env["CUDA_VISIBLE_DEVICES"] = ""
        self.procs = [subprocess.Popen([sys.executable, "-m", "blindshadowremoval_amd._row_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                       env=env, bufsize=0) for _ in range(n)]
        for p in self.procs:
            os.set_blocking(p.stdout.fileno(), False)
            os.set_blocking(p.stdin.fileno(), False)
            try:                                 # a finished result should fit in the pipe, so the worker can start its next job before we read
                import fcntl
                fcntl.fcntl(p.stdout.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), 1 << 20)
            except Exception:
                pass
        self._buf = [bytearray() for _ in range(n)]
        self._out = [bytearray() for _ in range(n)]      # bytes of jobs not yet accepted by the worker's stdin pipe (large jobs: post-processing, PNG strips)
        self._load = [0] * n                     # jobs outstanding per worker
        self._owner = [[] for _ in range(n)]     # their sequence numbers, oldest first (a worker answers in order)
        self._done = {}
        self._seq = 0

    def _send(self, w: int, job) -> int:
        import pickle
        import struct
        payload = pickle.dumps(job, protocol=pickle.HIGHEST_PROTOCOL)
        self._out[w] += struct.pack("<Q", len(payload)) + payload
        self._flush_out(w)
        seq = self._seq
        self._seq += 1
        self._load[w] += 1
        self._owner[w].append(seq)
        return seq

    def _flush_out(self, w: int) -> None:
        out = self._out[w]
        while out:
            try:
                n = os.write(self.procs[w].stdin.fileno(), memoryview(out)[:1 << 20])
            except BlockingIOError:
                return
            del out[:n]

    def submit(self, job) -> int:
        """Queue a job on the least loaded worker; -> ticket for result()."""
        return self._send(min(range(len(self.procs)), key=self._load.__getitem__), job)

    def result(self, seq: int):
        """Value of ticket `seq`; a job that failed in its worker raises HERE (once, for its own ticket) — the worker's other tickets
        stay valid and keep their order."""
        while seq not in self._done:
            self._pump(block=True)
        value = self._done.pop(seq)
        if isinstance(value, _JobFailed):
            raise RuntimeError("loader worker failed: %s" % value.args[0])
        return value

    def _parse(self, w: int) -> int:
        """Move every COMPLETE reply in worker w's buffer to _done (a failed job as a _JobFailed value under its own ticket: the
        ticket queue and the load count advance for failures exactly as for results); -> replies parsed."""
        import pickle
        import struct
        buf, n_done = self._buf[w], 0
        while len(buf) >= 8:
            n = struct.unpack_from("<Q", buf)[0]
            if len(buf) < 8 + n:
                break
            status, value = pickle.loads(bytes(buf[8:8 + n]))
            del buf[:8 + n]
            seq = self._owner[w].pop(0)
            self._load[w] -= 1
            self._done[seq] = value if status == "ok" else _JobFailed(value)
            n_done += 1
        return n_done

    def _pump(self, block: bool) -> None:
        import select
        if sum(self._parse(w) for w in range(len(self.procs)) if self._buf[w]):
            block = False                        # replies were already sitting in a buffer: hand them out before waiting for more
        fds = {p.stdout.fileno(): i for i, p in enumerate(self.procs) if self._load[i] > 0}
        wfds = {p.stdin.fileno(): i for i, p in enumerate(self.procs) if self._out[i]}
        if not fds and not wfds:
            return
        ready, wready, _ = select.select(list(fds), list(wfds), [], None if block else 0)
        for fd in wready:
            self._flush_out(wfds[fd])
        for fd in ready:
            w = fds[fd]
            buf = self._buf[w]
            while True:
                try:
                    chunk = os.read(fd, 1 << 20)
                except BlockingIOError:
                    break
                if not chunk:
                    self._parse(w)
                    raise RuntimeError("loader worker exited (code %s)" % self.procs[w].poll())
                buf += chunk
            self._parse(w)

    def imap(self, jobs, depth: int):
        jobs = iter(jobs)
        nxt = sent = self._seq          # tickets are pool-wide sequence numbers: this iteration's run from here
        exhausted = False
        per_worker = max(1, (depth + len(self.procs) - 1) // len(self.procs))
        while True:
            while not exhausted and sent - nxt < depth:
                w = min(range(len(self.procs)), key=self._load.__getitem__)
                if self._load[w] >= per_worker:
                    break
                try:
                    job = next(jobs)
                except StopIteration:
                    exhausted = True
                    break
                self._send(w, job)
                sent += 1
            if nxt in self._done:
                value = self._done.pop(nxt)
                if isinstance(value, _JobFailed):
                    raise RuntimeError("loader worker failed: %s" % value.args[0])
                yield value
                nxt += 1
                self._pump(block=False)
                continue
            if exhausted and nxt >= sent:
                return
            self._pump(block=True)

    def warm(self, kind: str, ring_path: Optional[str] = None) -> None:
        for w in range(len(self.procs)):
            self._send(w, ("warm", kind) + ((ring_path,) if ring_path else ()))
        while any(self._load):
            self._pump(block=True)
        self._done.clear()
        self._seq = 0

    def shutdown(self) -> None:
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()                # the exact child we started
        self.procs = []
