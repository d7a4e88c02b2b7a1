"""Image-quality metrics the reference prints at test time (host side):
`tf.image.psnr` / `tf.image.ssim` as used in /root/reference/train_test_GSC.py:724-725 (UCB) and
/root/reference/train_with_TSM.py:681-682, and the ROC AUC of train_with_TSM.py:701 (`fsrnet.roc_auc_score`).

`ssim` follows tf.image.ssim's defaults: 11x11 Gaussian window with sigma 1.5, k1 = 0.01, k2 = 0.03, 'VALID' filtering,
per-channel SSIM maps averaged over space and channels.  Inputs are NHWC torch tensors (any device)."""
from __future__ import annotations

import torch


def psnr(a: torch.Tensor, b: torch.Tensor, max_val: float = 1.0) -> torch.Tensor:
    """tf.image.psnr: 20 log10(max) - 10 log10(mean squared error) over the last three dims -> [B]."""
    mse = ((a.double() - b.double()) ** 2).mean(dim=(-3, -2, -1))
    return (20.0 * torch.log10(torch.tensor(float(max_val), dtype=torch.float64)) - 10.0 * torch.log10(mse)).float()


def _gauss_window(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    x = torch.arange(size, dtype=torch.float64) - (size - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum()
    return torch.outer(g, g)


def ssim(a: torch.Tensor, b: torch.Tensor, max_val: float = 1.0, filter_size: int = 11, filter_sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """tf.image.ssim -> [B]."""
    assert a.shape == b.shape and a.dim() == 4
    # tf.image.ssim computes in the inputs' float32.  The 11x11 window is the outer product of a 1-D Gaussian with itself, so 'VALID'
    # filtering is two depthwise 11-tap convolutions (vertical, horizontal) over the 5 x C stacked maps x, y, x^2, y^2, xy — 7 ms per
    # 256x256x3 pair on one core (round 2's pair of banded float64 matmuls: 25 ms, a 121-tap grouped convolution before that: 0.8 s;
    # this is the largest single term of the UCB loop's per-item host cost).  Agreement with the float64 form: ~1e-8.
    x = a.float().permute(0, 3, 1, 2)
    y = b.float().permute(0, 3, 1, 2)
    maps = torch.cat([x, y, x * x, y * y, x * y], dim=1)
    C = maps.shape[1]
    g1 = _gauss_window(filter_size, filter_sigma).sum(dim=1).to(device=x.device, dtype=torch.float32)      # rows of the normalised outer product sum to the 1-D window
    kv = g1.view(1, 1, filter_size, 1).expand(C, 1, filter_size, 1).contiguous()
    kh = g1.view(1, 1, 1, filter_size).expand(C, 1, 1, filter_size).contiguous()
    f = torch.nn.functional.conv2d(torch.nn.functional.conv2d(maps, kv, groups=C), kh, groups=C)
    mx, my, xx, yy, xy = f.split(x.shape[1], dim=1)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    sxx, syy, sxy = xx - mx * mx, yy - my * my, xy - mx * my
    lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    return (lum * cs).mean(dim=(1, 2, 3)).float()
