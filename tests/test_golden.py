"""The committed golden forward (tests/golden/gsc_forward_seed1.npz, made by tools/make_golden_forward.py)
pins the oracle against drift on CPU; the -m gpu test compares the HIP path with the same file."""
import os

import numpy as np
import pytest
import torch

from blindshadowremoval_amd.weights import init_weights
from oracle.gsc_oracle import GeneratorOracle


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "gsc_forward_seed1.npz"))
    inp = z["img_u8"].astype(np.float32) / 255.0
    uv = z["uv_u8"].astype(np.float32) / 255.0
    mask22 = np.stack([z["mask22_pos"], np.zeros_like(z["mask22_pos"]), z["mask22_neg"]], -1)
    return z, inp, uv, (z["gs"], z["con_rgb"], mask22, z["dif"])


def test_oracle_reproduces_golden(golden_dir):
    z, inp, uv, want = load(golden_dir)
    pr = {}
    got = GeneratorOracle(init_weights(int(z["seed_w"])))(inp, uv, probes=pr)
    assert float(z["margin"]) > 3e-4
    np.testing.assert_array_equal(pr["bmask"].numpy().astype(np.uint8), z["bmask"])
    for a, b, name in zip(got, want, ("gs", "con_rgb", "mask22", "dif")):
        np.testing.assert_allclose(a.numpy(), b, atol=2e-5, err_msg=name)     # other BLAS / thread counts reorder sums


@pytest.mark.gpu
def test_hip_matches_golden(golden_dir):
    from blindshadowremoval_amd import Generator
    z, inp, uv, want = load(golden_dir)
    gen = Generator().load_weights(init_weights(int(z["seed_w"])))
    out = gen(torch.from_numpy(inp).cuda(), torch.from_numpy(uv).cuda())
    np.testing.assert_array_equal(gen.probe("bmask").cpu().numpy().astype(np.uint8), z["bmask"])
    assert float(np.abs(gen.probe("d32").cpu().numpy() - z["d32"]).max()) <= 1e-3
    for a, b, name in zip(out, want, ("gs", "con_rgb", "mask22", "dif")):
        err = float(np.abs(a.cpu().numpy() - b).max())
        assert err <= 1e-3, "%s: %g" % (name, err)       # north_star tolerance: 1e-3 per pixel, fp32
