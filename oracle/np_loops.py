"""Second, independent CPU restatement (numpy, explicit loops over pixels and taps) — TEST
INFRASTRUCTURE ONLY; see the header of oracle/gsc_oracle.py (PARITY UNPINNED applies here too).

It shares no code with gsc_oracle.py: padding, the transposed-conv scatter, the bilinear resize and
the attention are written from the definitions in SURVEY.md Appendix A, so an indexing or padding
slip in one form shows up as a disagreement (tests/test_oracle_two_forms.py).  Accumulation is
float64; use only on small shapes.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

BN_EPS = 1e-3
ALPHA = 0.3


def conv2d_same(x, k, b, stride=1):
    """A.1 — x [B,H,W,Ci], k [kh,kw,Ci,Co] (model.py:119)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = k.shape
    Ho, Wo = math.ceil(H / stride), math.ceil(W / stride)
    pt = max((Ho - 1) * stride + kh - H, 0) // 2
    pl = max((Wo - 1) * stride + kw - W, 0) // 2
    y = np.zeros((B, Ho, Wo, Co), np.float64)
    for oy in range(Ho):
        for ox in range(Wo):
            acc = np.zeros((B, Co), np.float64)
            for a in range(kh):
                iy = oy * stride + a - pt
                if iy < 0 or iy >= H:
                    continue
                for c in range(kw):
                    ix = ox * stride + c - pl
                    if ix < 0 or ix >= W:
                        continue
                    acc += x[:, iy, ix, :].astype(np.float64) @ k[a, c].astype(np.float64)
            y[:, oy, ox, :] = acc + b
    return y


def conv2d_transpose_same(x, k, b):
    """A.2 — x [B,H,W,Ci], k [3,3,Co,Ci] (model.py:153), stride 2, output 2H x 2W."""
    B, H, W, Ci = x.shape
    kh, kw, Co, _ = k.shape
    full = np.zeros((B, 2 * H + 1, 2 * W + 1, Co), np.float64)
    for i in range(H):
        for j in range(W):
            v = x[:, i, j, :].astype(np.float64)
            for a in range(kh):
                for c in range(kw):
                    full[:, 2 * i + a, 2 * j + c, :] += v @ k[a, c].astype(np.float64).T
    return full[:, :2 * H, :2 * W, :] + b


def batchnorm(x, gamma, beta, mean, var):
    return (x - mean) / np.sqrt(var.astype(np.float64) + BN_EPS) * gamma + beta


def lrelu(x):
    return np.where(x >= 0, x, ALPHA * x)


def resize_bilinear(x, oh, ow):
    """A.5 — half-pixel-centre bilinear, no antialias, edge clamp (tf.image.resize TF2 default)."""
    B, H, W, C = x.shape
    y = np.zeros((B, oh, ow, C), np.float64)
    sy, sx = H / oh, W / ow
    for oy in range(oh):
        fy = (oy + 0.5) * sy - 0.5
        y0 = math.floor(fy)
        wy = fy - y0
        y0c, y1c = min(max(y0, 0), H - 1), min(max(y0 + 1, 0), H - 1)
        for ox in range(ow):
            fx = (ox + 0.5) * sx - 0.5
            x0 = math.floor(fx)
            wx = fx - x0
            x0c, x1c = min(max(x0, 0), W - 1), min(max(x0 + 1, 0), W - 1)
            top = x[:, y0c, x0c] * (1 - wx) + x[:, y0c, x1c] * wx
            bot = x[:, y1c, x0c] * (1 - wx) + x[:, y1c, x1c] * wx
            y[:, oy, ox] = top * (1 - wy) + bot * wy
    return y


def gray(x):
    return (0.2989 * x[..., 0:1].astype(np.float64) + 0.5870 * x[..., 1:2] + 0.1140 * x[..., 2:3])


def _bn(w, stem, x):
    return batchnorm(x, w[stem + "/gamma"], w[stem + "/beta"], w[stem + "/moving_mean"], w[stem + "/moving_variance"])


def non_local(w: Dict[str, np.ndarray], st: str, x):
    B, H, W, C = x.shape
    def c1(n):
        return conv2d_same(x, w[st + n + "/kernel"], w[st + n + "/bias"])
    g, phi, theta = (c1(n).reshape(B, H * W, -1) for n in ("g", "phi", "theta"))
    out = np.zeros_like(g)
    for b in range(B):
        for t in range(H * W):
            f = phi[b] @ theta[b, t]
            e = np.exp(f - f.max())
            out[b, t] = (e / e.sum()) @ g[b]
    y = out.reshape(B, H, W, -1)
    wy = _bn(w, st + "bnorm", conv2d_same(y, w[st + "w/kernel"], w[st + "w/bias"]))
    return x + wy


def res_bottleneck(w, i, x):
    st = "res_stack/%d/" % i
    y = lrelu(_bn(w, st + "bnorm1", conv2d_same(x, w[st + "conv1/kernel"], w[st + "conv1/bias"])))
    y = lrelu(_bn(w, st + "bnorm2", conv2d_same(y, w[st + "conv2/kernel"], w[st + "conv2/bias"])))
    y = _bn(w, st + "bnorm3", conv2d_same(y, w[st + "conv3/kernel"], w[st + "conv3/bias"]))
    y = non_local(w, st + "non_local/", y)
    cx, cy = x.shape[-1], y.shape[-1]
    if cx < cy:
        x = np.concatenate([x, np.zeros(x.shape[:3] + (cy - cx,))], -1)
    elif cy < cx:
        y = np.concatenate([y, np.zeros(y.shape[:3] + (cx - cy,))], -1)
    return lrelu(x + y)


def generator(w: Dict[str, np.ndarray], inputs, uv):
    """model.py:228-290, any H=W multiple of 8 (use <= 32 — this is O(pixels * taps) Python)."""
    def conv(x, stem, stride=1, bn=True, act=True):
        y = conv2d_same(x, w[stem + "/conv/kernel"], w[stem + "/conv/bias"], stride)
        if bn:
            y = _bn(w, stem + "/bnorm", y)
        return lrelu(y) if act else y

    def convt(x, stem):
        return lrelu(_bn(w, stem + "/bnorm", conv2d_transpose_same(x, w[stem + "/conv/kernel"], w[stem + "/conv/bias"])))

    x1 = conv(inputs, "conv1")
    x2 = conv(x1, "down1", 2)
    x3 = conv(x2, "down2", 2)
    x = conv(x3, "down3", 2)
    h, ww = x.shape[1], x.shape[2]
    uvs = resize_bilinear(uv, h, ww)
    x = np.concatenate([x, uvs], -1)
    for i in range(3):
        x = res_bottleneck(w, i, x)
    y = convt(x, "up1")
    y = convt(np.concatenate([y, x3], -1), "up2")
    y = convt(np.concatenate([y, x2], -1), "up3")
    mask = np.tanh(conv(y, "conv2", bn=False, act=False))
    con = conv(y, "conv3", bn=False, act=False)
    g0 = gray(inputs)
    gs = g0 * (1 + mask) + con
    dif = gs - g0
    mask22 = np.concatenate([np.maximum(mask, 0), mask * 0, np.maximum(-mask, 0)], -1)
    d32 = resize_bilinear(dif, h, ww)
    bmask = (d32.astype(np.float32) > np.float32(0.1)).astype(np.float64)
    x = np.concatenate([x * (1 - bmask), bmask, uvs], -1)
    for i in range(3, 6):
        x = res_bottleneck(w, i, x)
    f = convt(convt(convt(x, "clr_up1"), "clr_up2"), "clr_up3")
    c = conv(np.concatenate([gs, f], -1), "clr_conv1")
    c = conv(c, "clr_conv2")
    con_rgb = conv(c, "clr_conv3", bn=False, act=False)
    dif2 = gray(con_rgb) - gray(inputs)
    return gs, con_rgb, mask22, dif2, dict(d32=d32, bmask=bmask)
