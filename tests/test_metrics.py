"""PSNR / SSIM restatements (tf.image.psnr / tf.image.ssim defaults) and the loud failure of the HIP path without its library."""
import numpy as np
import pytest
import torch

from blindshadowremoval_amd import metrics as M


def test_psnr_known_values():
    a = torch.zeros(2, 8, 8, 3)
    b = torch.full((2, 8, 8, 3), 0.1)
    np.testing.assert_allclose(M.psnr(a, b).numpy(), [20.0, 20.0], rtol=1e-5)         # mse = 0.01 -> 20 dB
    np.testing.assert_allclose(M.psnr(a * 255, b * 255, 255.0).numpy(), [20.0, 20.0], rtol=1e-5)


def test_ssim_properties_and_constant_shift():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((2, 32, 32, 3)).astype(np.float32))
    assert torch.allclose(M.ssim(x, x), torch.ones(2), atol=1e-6)
    y = (x + 0.05 * torch.from_numpy(rng.standard_normal(x.shape).astype(np.float32))).clamp(0, 1)
    s = M.ssim(x, y)
    assert torch.all(s < 1) and torch.all(s > 0.5) and torch.allclose(s, M.ssim(y, x), atol=1e-6)
    # two constant images: variances vanish, SSIM = luminance term (2 m1 m2 + c1) / (m1^2 + m2^2 + c1)
    a, b = torch.full((1, 16, 16, 1), 0.4), torch.full((1, 16, 16, 1), 0.5)
    want = (2 * 0.4 * 0.5 + 1e-4) / (0.16 + 0.25 + 1e-4)
    assert abs(float(M.ssim(a, b)) - want) < 1e-6
    w = M._gauss_window()
    assert w.shape == (11, 11) and abs(float(w.sum()) - 1) < 1e-12 and float(w[5, 5]) == float(w.max())


def test_hip_path_fails_loudly_without_its_library(monkeypatch, tmp_path):
    from blindshadowremoval_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libbsr_hip.so"))
    with pytest.raises(RuntimeError, match="not built"):
        _lib.load()
    from blindshadowremoval_amd import Generator, init_weights
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            Generator().load_weights(init_weights(1))


def test_ssim_matches_an_independent_windowed_form():
    """tf.image.ssim's definition restated directly: for every 11x11 window (VALID) the Gaussian-weighted means, variances and
    covariance, the SSIM index per window and channel, averaged — explicit loops in float64, no shared code with metrics.ssim's
    banded-matrix filter.  (The reference's UCB step prints this metric: train_test_GSC.py:724.)"""
    rng = np.random.default_rng(4)
    a = rng.random((1, 20, 23, 3))
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    x = np.arange(11) - 5.0
    g = np.exp(-x ** 2 / (2 * 1.5 ** 2))
    g /= g.sum()
    w = np.outer(g, g)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    vals = []
    for c in range(3):
        for i in range(20 - 10):
            for j in range(23 - 10):
                pa, pb = a[0, i:i + 11, j:j + 11, c], b[0, i:i + 11, j:j + 11, c]
                ma, mb = (w * pa).sum(), (w * pb).sum()
                va, vb, cab = (w * pa * pa).sum() - ma * ma, (w * pb * pb).sum() - mb * mb, (w * pa * pb).sum() - ma * mb
                vals.append((2 * ma * mb + c1) / (ma * ma + mb * mb + c1) * (2 * cab + c2) / (va + vb + c2))
    got = float(M.ssim(torch.from_numpy(a), torch.from_numpy(b)))
    assert abs(got - float(np.mean(vals))) < 1e-6
