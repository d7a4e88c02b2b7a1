"""Generate tests/golden/ucb_post_9156.npz: inputs and outputs of the reference's OWN UCB post-processing.

The body of `FSRNet.test_step` (/root/reference/train_test_GSC.py:411-748) is taken from the reference file at run time
(ast -> compile; nothing of it is written to this repository) and executed with
  * `tf`  = a small numpy-backed stand-in for the ~25 TensorFlow calls the function makes (ndarray subclass with .numpy();
            tf.image.resize = bilinear half-pixel, tf.round = round-half-even, tf.image.ssim / psnr = the restatements in
            blindshadowremoval_amd.metrics — these two values are therefore pinned only to our own restatement, every
            threshold / region / component decision is the reference's code),
  * `cv2` = connectedComponentsWithStats over scipy.ndimage.label, imwrite = no-op,
  * `self.gen` = a stub returning prepared generator outputs.
Inputs (tests/ucb_cases.py, shared with the test): the two UCB items kept under tests/golden/UCB (rows built by
blindshadowremoval_amd.dataset, itself pinned to the reference's dataset code by tests/golden/sample_02165.npz) with their seven
mask PNGs (tests/golden/UCB_masks), and as generator outputs a realistic synthetic prediction: con_rgb = ground truth + noise,
dif = gray(gt) - gray(input) (the true shadow magnitude), plus scaled variants that push the heuristics through other branches.
Only the expected OUTPUTS are stored.

    python tools/make_ucb_post_fixture.py [--backend standin|tf] [--out PATH]       # needs /root/reference

--backend tf — the pin of `tf.image.resize` / `tf.image.ssim` / `tf.image.psnr` that is missing here (no TensorFlow in this image): on a
machine with TensorFlow 2.3 the reference's `test_step` runs over REAL `tf` (and real `cv2` when importable) on the same cases and the
same npz keys are written, plus `backend = "tf-<version>"`; tests/test_ucb_post.py and the GPU tests then compare against TensorFlow's
own resize rounding and SSIM / PSNR without any change.  BSR_MOCK_TF=1 exercises that code path here over the stand-in presented as a
`tensorflow` module (tests/test_ucb_post.py: a test of this tool's logic, not a pin).
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class T(np.ndarray):
    """eager-tensor stand-in: an ndarray with .numpy(), which COPIES as EagerTensor.numpy() does (the reference writes into
    `curr_mask_no_hair.numpy()` at :535-537 and must not alter the tensor)"""

    def numpy(self):
        return np.array(self, copy=True)


def _t(x, dtype=None):
    return np.asarray(x, dtype=dtype).view(T)


def make_tf():
    from blindshadowremoval_amd.metrics import psnr, ssim
    from blindshadowremoval_amd.ucb_post import resize_bilinear
    tf = types.SimpleNamespace()
    tf.float32, tf.uint8 = np.float32, np.uint8
    tf.reshape = lambda x, s: _t(np.reshape(np.asarray(x), [int(v) for v in s]))
    tf.split = lambda x, sizes, axis: [_t(p) for p in np.split(np.asarray(x), np.cumsum(sizes)[:-1], axis=axis)]
    tf.shape = lambda x: _t(np.asarray(np.asarray(x).shape))
    tf.reduce_max = lambda x: _t(np.max(np.asarray(x)))
    tf.reduce_min = lambda x: _t(np.min(np.asarray(x)))
    tf.reduce_sum = lambda x: _t(np.sum(np.asarray(x), dtype=np.asarray(x).dtype))
    tf.reduce_mean = lambda x, axis=None: _t(np.mean(np.asarray(x), axis=axis, dtype=np.asarray(x).dtype))
    tf.round = lambda x: _t(np.round(np.asarray(x)))
    tf.pad = lambda x, p: _t(np.pad(np.asarray(x), [[int(a), int(b)] for a, b in p]))
    tf.cast = lambda x, d: _t(np.asarray(x).astype(d))
    tf.convert_to_tensor = lambda x: _t(x)
    tf.logical_and = lambda a, b: _t(np.logical_and(np.asarray(a), np.asarray(b)))
    tf.logical_not = lambda a: _t(np.logical_not(np.asarray(a)))
    tf.greater = lambda a, b: _t(np.asarray(a) > np.asarray(b))
    tf.concat = lambda xs, axis: _t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.clip_by_value = lambda x, lo, hi: _t(np.clip(np.asarray(x), lo, hi))
    img = types.SimpleNamespace()
    img.resize = lambda x, size: _t(resize_bilinear(np.asarray(x, np.float32), int(size[0])))
    img.ssim = lambda a, b, max_val: _t(ssim(torch.from_numpy(np.asarray(a, np.float32))[None], torch.from_numpy(np.asarray(b, np.float32))[None], max_val).numpy())
    img.psnr = lambda a, b, max_val: _t(psnr(torch.from_numpy(np.asarray(a, np.float32))[None], torch.from_numpy(np.asarray(b, np.float32))[None], max_val).numpy())
    tf.image = img
    return tf


def make_cv2():
    from scipy import ndimage
    cv2 = types.SimpleNamespace()

    def cc(arr, connectivity=8):
        assert connectivity == 4
        labels, n = ndimage.label(arr, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        areas = np.bincount(labels.reshape(-1), minlength=n + 1)
        stats = np.zeros((n + 1, 5), np.int64)
        stats[:, -1] = areas
        return n + 1, labels, stats, None
    cv2.connectedComponentsWithStats = cc
    cv2.imwrite = lambda *a, **k: True
    return cv2


def load_backend(backend: str):
    """-> (tf module, cv2 module, label).  "standin": the numpy-backed stand-ins above.  "tf": real TensorFlow (+ real cv2 if present);
    with BSR_MOCK_TF=1 the stand-in dressed as the module (a dry run of this path)."""
    if backend == "standin":
        return make_tf(), make_cv2(), "standin"
    if os.environ.get("BSR_MOCK_TF") == "1":
        mock = make_tf()
        mock.__version__ = "mock"
        return mock, make_cv2(), "tf-mock"
    try:
        import tensorflow as tf
    except ImportError as e:
        raise SystemExit("--backend tf needs TensorFlow (the reference pins 2.3.0, README.md:11): %s" % e)
    try:
        import cv2
    except ImportError:
        cv2 = make_cv2()
    return tf, cv2, "tf-%s" % tf.__version__


def reference_test_step(tf_mod=None, cv2_mod=None):
    """`FSRNet.test_step` compiled from the reference's file, with tf / cv2 / np bound to the given modules (default: the stand-ins above)."""
    with open(os.path.join(REF, "train_test_GSC.py")) as fsrc:
        src = fsrc.read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "FSRNet")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "test_step")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"tf": tf_mod if tf_mod is not None else make_tf(), "cv2": cv2_mod if cv2_mod is not None else make_cv2(), "np": np, "print": lambda *a, **k: None}
    exec(compile(mod, "<reference test_step>", "exec"), ns)
    return ns["test_step"]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    backend, out_path = "standin", os.path.join(ROOT, "tests", "golden", "ucb_post_9156.npz")
    while argv:
        a = argv.pop(0)
        if a == "--backend" and argv:
            backend = argv.pop(0)
            if backend not in ("standin", "tf"):
                raise SystemExit("--backend must be standin or tf")
        elif a == "--out" and argv:
            out_path = argv.pop(0)
        else:
            raise SystemExit(__doc__)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ucb_cases import cases
    tf_mod, cv2_mod, label = load_backend(backend)
    step = reference_test_step(tf_mod, cv2_mod)
    real_tf = label.startswith("tf-") and label != "tf-mock"
    wrap = (lambda x, dtype=None: tf_mod.convert_to_tensor(np.asarray(x, dtype=dtype))) if real_tf else _t
    out = {"backend": np.array(label)}
    for key, row, box, m, con, dif in cases():
        fake = types.SimpleNamespace(config=types.SimpleNamespace(IMG_SIZE=256))
        fake.gen = lambda im, uv, reg, chuck, training: (None, wrap(np.repeat(con[None], 10, 0)), None, wrap(np.repeat(dif[None], 10, 0)))
        stack = np.repeat(row[None], 10, axis=0)
        losses, figs = step(fake, wrap(stack), wrap(np.asarray(box, np.int32)), wrap(m["face_hair"]), wrap(m["face"]), wrap(m["mouth"]), wrap(m["nose"]),
                            wrap(m["eyebrow"]), wrap(m["eye"]), wrap(m["glasses"]), False)
        out[key + "_ssim"] = np.float32(losses["ssim"])
        out[key + "_psnr"] = np.float32(losses["psnr"])
        out[key + "_detected"] = np.asarray(figs[4])[0, :, :, 0].astype(np.uint8)
        if key.endswith("a"):                              # the composite image itself for one case per item (size)
            out[key + "_out"] = np.asarray(figs[1])[0].astype(np.float16)
        print(key, "size", int(box[3] - box[1]), "ssim %.4f psnr %.2f detected %d px" % (losses["ssim"], losses["psnr"], int(out[key + "_detected"].sum())))
    np.savez_compressed(out_path, **out)
    return out


if __name__ == "__main__":
    main()
