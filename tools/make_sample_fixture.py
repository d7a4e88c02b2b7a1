"""Generate tests/golden/sample_02165.npz — the real-input fixture of BASELINE config 1.

Runs IN THE BUILD CONTAINER ONLY: it imports the reference's own host-side input preparation from
/root/reference (`utils.face_crop_and_resize`, `utils.generate_face_region`, `warp.generate_uv_map`,
`warp.generate_offset_map`, and the canonical `uv` / `lm_ref` tables of `dataset.py`) and applies it to
/root/reference/sample_imgs/02165 in the order of `parse_fn_test_FFHQ` (/root/reference/dataset.py:619-640),
producing one `[256,256,16]` row = [img3, gt3, uvm3, reg_in3, reg_out3, face1].

TensorFlow / tensorflow_addons / cv2 / skimage / natsort are not installed here, so they are stubbed in
sys.modules; the three cv2 functions the path really executes are restated below (documented OpenCV
semantics) and PNG decoding uses PIL:
  cv2.imread + cvtColor(BGR2RGB)  -> PIL RGB decode
  cv2.resize(img, (256,256))      -> INTER_LINEAR: half-pixel centres, edge clamp, no antialias
  cv2.GaussianBlur(m, (5,5), 0)   -> separable [1,4,6,4,1]/16 (OpenCV's fixed kernel for ksize 5, sigma<=0),
                                     BORDER_REFLECT_101
The fixture is DATA (inputs); the reference's outputs for it do not exist (no weights, no TF).
"""
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _resize_linear(img, dsize):
    w, h = dsize
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float64))
    squeeze = t.dim() == 2
    if squeeze:
        t = t[..., None]
    out = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None], size=(h, w), mode="bilinear", align_corners=False, antialias=False)
    out = out[0].permute(1, 2, 0).numpy()
    return out[..., 0] if squeeze else out


def _gaussian_blur(img, ksize, sigma):
    assert tuple(ksize) == (5, 5) and sigma == 0
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16.0
    a = np.asarray(img, np.float64)
    if a.ndim == 3 and a.shape[2] == 1:            # cv2 returns HxW for an HxWx1 input
        a = a[..., 0]
    p = np.pad(a, 2, mode="reflect")               # numpy 'reflect' == BORDER_REFLECT_101
    tmp = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(5))
    out = sum(k[i] * tmp[i:i + a.shape[0], :] for i in range(5))
    return out.astype(img.dtype)


def _install_stubs():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            m = _Any(self.__name__ + "." + name)
            setattr(self, name, m)
            return m

        def __call__(self, *a, **k):
            return _Any("call")
    for name in ("tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow_addons", "skimage", "skimage.draw",
                 "natsort", "scipy.misc"):
        sys.modules[name] = _Any(name)
    sys.modules["skimage.draw"].line_aa = None
    cv2 = types.ModuleType("cv2")
    cv2.resize = lambda img, dsize, **k: _resize_linear(img, dsize)
    cv2.GaussianBlur = _gaussian_blur
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path: np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()
    cv2.cvtColor = lambda img, code: img[:, :, ::-1].copy()
    sys.modules["cv2"] = cv2
    import scipy.ndimage
    sys.modules.setdefault("scipy.ndimage.interpolation", scipy.ndimage)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import cv2
    import dataset as ref_dataset          # defines uv, lm_ref (dataset.py:10-17)
    from utils import face_crop_and_resize, generate_face_region
    from warp import generate_offset_map, generate_uv_map

    lm_path = os.path.join(REF, "sample_imgs/02165/02165.npy")
    img_path = lm_path.split(".")[0] + ".png"
    img = cv2.cvtColor(cv2.imread(img_path), cv2.COLOR_BGR2RGB) / 255.
    gt = img
    img = np.concatenate([img, gt], axis=2)
    img, lm, lm_mirror, box = face_crop_and_resize(img, np.load(lm_path), 256)
    face = generate_face_region(lm, 256).reshape(256, 256, 1)
    uvm = generate_uv_map(lm, ref_dataset.uv, 256)
    reg_in = generate_offset_map(lm, ref_dataset.lm_ref, 256)
    reg_out = generate_offset_map(ref_dataset.lm_ref, lm, 256)
    row = np.concatenate([img, uvm, reg_in, reg_out, face], axis=2).astype(np.float32)
    assert row.shape == (256, 256, 16)
    print("img", row[..., :3].min(), row[..., :3].max(), "uv", row[..., 6:9].min(), row[..., 6:9].max(),
          "uv zeros %.2f" % float((row[..., 6:9] == 0).mean()), "reg", row[..., 9:15].min(), row[..., 9:15].max(),
          "face", row[..., 15].min(), row[..., 15].max(), "box", box)
    dst = os.path.join(ROOT, "tests", "golden", "sample_02165.npz")
    np.savez_compressed(dst, row=row, box=np.asarray(box, np.float32), lm=np.asarray(lm, np.float32),
                        name="sample_imgs/02165/02165.png")
    print(dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
