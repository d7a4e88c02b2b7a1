"""Extract the canonical face tables the reference's loaders embed as Python literals
(/root/reference/dataset.py:10-17: `uv` 68x3 canonical UVZ coordinates, `lm_ref` 68x2 reference landmarks / 256)
into blindshadowremoval_amd/data/face_model.npz.  DATA only; run in the build container:
    python tools/make_face_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_sample_fixture import _install_stubs  # noqa: E402

_install_stubs()
sys.path.insert(0, "/root/reference")
import dataset as ref_dataset  # noqa: E402

uv = np.asarray(ref_dataset.uv, np.float32)
lm_ref = np.asarray(ref_dataset.lm_ref, np.float32)
assert uv.shape == (68, 3) and lm_ref.shape == (68, 2)
dst = os.path.join(ROOT, "blindshadowremoval_amd", "data", "face_model.npz")
np.savez(dst, uv=uv, lm_ref=lm_ref)
print(dst, uv.min(), uv.max(), lm_ref.min(), lm_ref.max())
