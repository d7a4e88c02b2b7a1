"""Generate tests/golden/sfw_synth/ (a tiny synthetic SFW-style video folder) and tests/golden/sfw_elements.npz — what the
reference's OWN TSM loaders make of it.

Runs IN THE BUILD CONTAINER ONLY: imports /root/reference/dataset_with_TSM.py (with utils.py / warp.py) over the same stand-ins as
tools/make_sample_fixture.py (TensorFlow etc. stubbed; cv2.imread / cvtColor / resize / GaussianBlur / flip restated with OpenCV's
documented semantics; tf.numpy_function simply calls the function) and calls `Dataset.parse_fn_test_sfw`
(dataset_with_TSM.py:225-287) and `Dataset.parse_fn_test_sfw_video` (:289-583) on the synthetic folder.  The SFW dataset itself is
not shipped upstream, so the folder is made here from the reference's sample face: frames <n>.png / <n>.npy are the 02165 image
shifted by a few pixels per frame (landmarks shifted alike), `<n>_label.png` a 3-level mask (0 / 1 / 2) and `<n>_label_cmap.png`
a colour rendering of it.  The elements are large ([2,256,256,17] and [10,256,256,13] float32), so the fixture stores every 4th
pixel of each plus per-channel sums; tests/test_dataset.py rebuilds them with blindshadowremoval_amd.dataset and compares."""
import contextlib
import io
import os
import sys
import types

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT_DIR = os.path.join(ROOT, "tests", "golden", "sfw_synth", "vid0")
FRAMES = list(range(1, 20))             # frame 1 groups with 3,5,...,17 and 2; frame 10 with 11..19 and 8,6,4,2


def make_folder():
    os.makedirs(OUT_DIR, exist_ok=True)
    src = Image.open(os.path.join(REF, "sample_imgs/02165/02165.png")).convert("RGB")
    src = src.resize((128, 128), Image.BILINEAR)                     # small files: the crop is resized to 256 anyway
    base = np.asarray(src)
    lm = np.load(os.path.join(REF, "sample_imgs/02165/02165.npy")).astype(np.float32) * 0.5
    for n in FRAMES:
        dx, dy = (n % 5) - 2, (n % 3) - 1
        img = np.roll(np.roll(base, dx, axis=1), dy, axis=0)
        Image.fromarray(img).save(os.path.join(OUT_DIR, "%d.png" % n))
        np.save(os.path.join(OUT_DIR, "%d.npy" % n), lm + np.array([dx, dy], np.float32))
    for n in (1, 10):
        img = np.asarray(Image.open(os.path.join(OUT_DIR, "%d.png" % n)).convert("L"), np.float32)
        label = np.digitize(img, [90.0, 160.0]).astype(np.uint8)      # 0 / 1 / 2
        Image.fromarray(label).save(os.path.join(OUT_DIR, "%d_label.png" % n))
        cmap = np.stack([label * 100, 255 - label * 100, label * 40 + 30], axis=2).astype(np.uint8)
        Image.fromarray(cmap).save(os.path.join(OUT_DIR, "%d_label_cmap.png" % n))


def main():
    make_folder()
    import make_sample_fixture as msf
    msf._install_stubs()
    import cv2
    cv2.flip = lambda img, code: np.ascontiguousarray(img[:, ::-1]) if code == 1 else (_ for _ in ()).throw(NotImplementedError())
    rgb = cv2.imread

    def imread(path, flag=1):
        if not os.path.isfile(path):
            return None
        if flag == 0:
            return np.asarray(Image.open(path).convert("L")).copy()
        return rgb(path)
    cv2.imread = imread
    tf = sys.modules["tensorflow"]
    tf.numpy_function = lambda fn, inp, Tout: fn(*inp)
    tf.ensure_shape = lambda x, shape: x
    tf.float32, tf.string = "float32", "string"
    tf.data = types.SimpleNamespace(experimental=types.SimpleNamespace(AUTOTUNE=-1))
    sys.path.insert(0, REF)
    import dataset_with_TSM as ref
    me = types.SimpleNamespace(config=types.SimpleNamespace(IMG_SIZE=256))
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for n in (1, 10):
            label = os.path.join(OUT_DIR, "%d_label.png" % n)
            img, box, name = ref.Dataset.parse_fn_test_sfw(me, label.encode())
            out["pair%d" % n], out["pair%d_box" % n] = np.asarray(img, np.float32), np.asarray(box, np.float32)
        for n in (1, 10):
            # the video parser is fed "<n>.png"-style names (its own `_lm = _mask.split('.')[0] + '.npy'`)
            img, box, name = ref.Dataset.parse_fn_test_sfw_video(me, os.path.join(OUT_DIR, "%d.png" % n).encode())
            out["video%d" % n], out["video%d_box" % n] = np.asarray(img, np.float32), np.asarray(box, np.float32)
    small = {}
    for k, v in out.items():
        if v.ndim == 4:
            assert v.shape[1:3] == (256, 256)
            small[k] = v[:, ::4, ::4, :].copy()
            small[k + "_sum"] = v.astype(np.float64).sum(axis=(1, 2))
        else:
            small[k] = v
    dst = os.path.join(ROOT, "tests", "golden", "sfw_elements.npz")
    np.savez_compressed(dst, **small)
    print(dst, os.path.getsize(dst), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
