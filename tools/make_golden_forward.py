"""Generate tests/golden/gsc_forward_seed1.npz: one 256x256 image pair (uint8-quantised like a decoded
PNG), the seeded synthetic weights' seed, and the ORACLE's four outputs + threshold probes.

The reference cannot run here (no TensorFlow; weights not shipped), so these vectors pin the oracle
against drift and give the GPU box a fixed target; they are NOT reference outputs (parity unpinned,
see oracle/gsc_oracle.py).  Run:  python tools/make_golden_forward.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blindshadowremoval_amd.weights import init_weights  # noqa: E402
from oracle.gsc_oracle import GeneratorOracle  # noqa: E402

SEED_W = 1
oracle = GeneratorOracle(init_weights(SEED_W))


def make(seed):
    rng = np.random.default_rng(seed)

    def field(c):      # smooth-ish synthetic image: low-frequency field + noise
        base = rng.random((1, 9, 9, c))
        t = torch.from_numpy(base).permute(0, 3, 1, 2)
        up = torch.nn.functional.interpolate(t, size=(256, 256), mode="bicubic", align_corners=True).permute(0, 2, 3, 1).numpy()
        return np.clip(up + 0.08 * rng.standard_normal((1, 256, 256, c)), 0, 1)
    img_u8 = np.round(field(3) * 255).astype(np.uint8)      # quantised to uint8 like a decoded PNG
    uv_u8 = np.round(field(3) * 255).astype(np.uint8)
    uv_u8[:, :40] = 0          # the real uv map is 0 outside the landmark hull (warp.py:231)
    uv_u8[:, :, 220:] = 0
    pr = {}
    out = oracle(img_u8.astype(np.float32) / 255.0, uv_u8.astype(np.float32) / 255.0, probes=pr)
    return img_u8, uv_u8, out, pr, float((pr["d32"] - 0.1).abs().min())


for seed in range(20221121, 20221160):      # first seed whose 1024 d32 cells all clear the 0.1 threshold by > 3e-4
    img_u8, uv_u8, (gs, con_rgb, mask22, dif), pr, margin = make(seed)
    print("seed %d: bmask mean %.3f, threshold margin %.2e" % (seed, float(pr["bmask"].mean()), margin))
    if margin > 3e-4:
        break
else:
    raise SystemExit("no seed found")
dst = os.path.join(ROOT, "tests", "golden", "gsc_forward_seed1.npz")
np.savez_compressed(dst, seed_w=SEED_W, seed_img=seed, img_u8=img_u8, uv_u8=uv_u8, gs=gs.numpy(), con_rgb=con_rgb.numpy(),
                    mask22_pos=mask22.numpy()[..., 0], mask22_neg=mask22.numpy()[..., 2], dif=dif.numpy(),
                    d32=pr["d32"].numpy(), bmask=pr["bmask"].numpy().astype(np.uint8), margin=margin)
print(dst, os.path.getsize(dst))
