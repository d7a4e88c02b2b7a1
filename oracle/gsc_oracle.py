"""CPU ORACLE for the GSC generator forward pass — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the shipped path (``blindshadowremoval_amd``) never does.

PARITY UNPINNED: the reference (/root/reference, TensorFlow 2.3) cannot be imported or run in the
build container (no tensorflow / tensorflow_addons / cv2), holds no tests and no golden outputs, and
its trained weights are absent (/root/reference/.MISSING_LARGE_BLOBS).  This file restates
/root/reference/model.py layer by layer with the documented TF-2.3 / Keras op semantics
(SURVEY.md Appendix A); what pins it is (1) hand-computable known-answer tests per semantic
(tests/test_oracle_kat.py), (2) agreement with an independent numpy direct-loop restatement
(oracle/np_loops.py) on small shapes, and (3) the variable inventory parsed from the reference's own
``ckpt-94.index`` (tests/golden/gsc_ckpt94_inventory.json) chaining shape-correctly through it.

Tensors are NHWC float32 torch CPU tensors, exactly as the reference passes them.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # tf.keras.layers.BatchNormalization default epsilon (model.py:16,87-89,122,155)
LRELU_ALPHA = 0.3    # tf.keras.layers.LeakyReLU default alpha (model.py:90-92,130,161)
GRAY = (0.2989, 0.5870, 0.1140)   # tf.image.rgb_to_grayscale weights (model.py:250,251,288)
BMASK_THRESHOLD = 0.1             # model.py:256


def _t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def same_pad(size: int, k: int, s: int) -> Tuple[int, int]:
    """TF 'SAME' padding for one axis: out = ceil(size/s); extra pad goes AFTER (A.1)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def conv2d_same(x: torch.Tensor, kernel_hwio, bias, stride: int = 1) -> torch.Tensor:
    """``layers.Conv2D(padding='same')`` (model.py:10-13,84-86,119): cross-correlation, HWIO kernel."""
    w = _t(kernel_hwio)
    kh, kw = w.shape[0], w.shape[1]
    xt = x.permute(0, 3, 1, 2)
    pt, pb = same_pad(x.shape[1], kh, stride)
    pl, pr = same_pad(x.shape[2], kw, stride)
    xt = F.pad(xt, (pl, pr, pt, pb))
    y = F.conv2d(xt, w.permute(3, 2, 0, 1).contiguous(), _t(bias), stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def conv2d_transpose_same(x: torch.Tensor, kernel_hwoi, bias, stride: int = 2) -> torch.Tensor:
    """``layers.Conv2DTranspose(k, strides=2, padding='same')`` (model.py:153): kernel
    ``[kh,kw,Cout,Cin]``; ``y[2i+a,2j+b,o] += x[i,j,c]*W[a,b,o,c]``, keep rows/cols [0, 2H) (A.2).

    For k=3, s=2 TF's SAME deconv pads (k-s)=1 in total, 0 before / 1 after, i.e. the full
    (2H+1)-long scatter result is cropped at the end."""
    w = _t(kernel_hwoi)
    kh = w.shape[0]
    assert stride == 2 and kh == 3, "only the reference's ConvT(3, stride 2) is restated"
    xt = x.permute(0, 3, 1, 2)
    y = F.conv_transpose2d(xt, w.permute(3, 2, 0, 1).contiguous(), _t(bias), stride=stride, padding=0)
    H, W = x.shape[1] * stride, x.shape[2] * stride
    return y[:, :, :H, :W].permute(0, 2, 3, 1).contiguous()


def batchnorm_infer(x: torch.Tensor, gamma, beta, mean, var) -> torch.Tensor:
    """``BatchNormalization()(x, training=False)`` (A.3)."""
    g, b, m, v = _t(gamma), _t(beta), _t(mean), _t(var)
    return (x - m) * torch.rsqrt(v + BN_EPS) * g + b


def leaky_relu(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x >= 0, x, x * LRELU_ALPHA)


def rgb_to_grayscale(x: torch.Tensor) -> torch.Tensor:
    """``tf.image.rgb_to_grayscale`` (A.6): tensordot with [0.2989, 0.5870, 0.1140], keeps a size-1 channel."""
    w = torch.tensor(GRAY, dtype=torch.float32)
    return (x * w).sum(-1, keepdim=True)


def resize_bilinear(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """``tf.image.resize(x, size)`` TF2 default: bilinear, half_pixel_centers, no antialias (A.5)."""
    y = F.interpolate(x.permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False, antialias=False)
    return y.permute(0, 2, 3, 1).contiguous()


class GeneratorOracle:
    """Restatement of ``Generator`` (/root/reference/model.py:198-290) at ``training=False``."""

    def __init__(self, weights: Dict[str, np.ndarray], n_res: int = 6):
        self.w = weights
        self.n_res = n_res

    # -- blocks -----------------------------------------------------------------------------
    def conv_block(self, x, stem: str, stride: int = 1, bn: bool = True, act: bool = True):
        """``Conv.call`` (model.py:139-147)."""
        y = conv2d_same(x, self.w[stem + "/conv/kernel"], self.w[stem + "/conv/bias"], stride)
        if bn:
            y = batchnorm_infer(y, *[self.w[stem + "/bnorm/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")])
        return leaky_relu(y) if act else y

    def convt_block(self, x, stem: str):
        """``ConvT.call`` (model.py:169-177)."""
        y = conv2d_transpose_same(x, self.w[stem + "/conv/kernel"], self.w[stem + "/conv/bias"], 2)
        y = batchnorm_infer(y, *[self.w[stem + "/bnorm/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")])
        return leaky_relu(y)

    def non_local(self, x, st: str, probes: Optional[dict] = None):
        """``NonLocalBlock.call`` (model.py:23-61) with pool=False."""
        b, h, w, c = x.shape
        def c1(name):
            return conv2d_same(x, self.w[st + name + "/kernel"], self.w[st + name + "/bias"], 1)
        g_x = c1("g").reshape(b, h * w, -1)                       # :33,36
        phi_x = c1("phi").reshape(b, h * w, -1).permute(0, 2, 1)  # :39-43
        theta_x = c1("theta").reshape(b, h * w, -1)               # :46-49
        f = torch.matmul(theta_x, phi_x)                          # :51
        f_softmax = torch.softmax(f, dim=-1)                      # :52
        y = torch.matmul(f_softmax, g_x).reshape(b, h, w, -1)     # :53-54
        if probes is not None:
            probes[st + "att"] = y
        w_y = conv2d_same(y, self.w[st + "w/kernel"], self.w[st + "w/bias"], 1)       # :56
        w_y = batchnorm_infer(w_y, *[self.w[st + "bnorm/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")])
        return x + w_y                                            # :59

    def res_bottleneck(self, x, i: int, probes: Optional[dict] = None):
        """``ResBottleneck.call`` (model.py:98-113), stride 1."""
        st = "res_stack/%d/" % i
        def cb(name, bn, inp):
            y = conv2d_same(inp, self.w[st + name + "/kernel"], self.w[st + name + "/bias"], 1)
            return batchnorm_infer(y, *[self.w[st + bn + "/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")])
        y = leaky_relu(cb("conv1", "bnorm1", x))
        y = leaky_relu(cb("conv2", "bnorm2", y))
        y = cb("conv3", "bnorm3", y)
        if probes is not None:
            probes[st + "y3"] = y
        y = self.non_local(y, st + "non_local/", probes)
        cx, cy = x.shape[-1], y.shape[-1]
        if cx < cy:                                               # :105-108
            x = torch.cat([x, torch.zeros(*x.shape[:3], cy - cx)], dim=3)
        elif cy < cx:                                             # :109-112
            y = torch.cat([y, torch.zeros(*y.shape[:3], cx - cy)], dim=3)
        return leaky_relu(x + y)                                  # :113

    # -- forward ----------------------------------------------------------------------------
    def forward(self, inputs, uv, reg=None, chuck=1, training=False, probes: Optional[dict] = None,
                bmask_override: Optional[torch.Tensor] = None):
        """``Generator.call`` (model.py:228-290).  ``reg`` / ``chuck`` are accepted and unused, as in
        the reference.  ``probes`` (dict) receives intermediates; ``bmask_override`` substitutes the
        thresholded mask (used by parity tests to separate threshold flips from arithmetic error)."""
        assert not training, "the oracle restates the inference path only"
        inputs, uv = _t(inputs), _t(uv)
        x1 = self.conv_block(inputs, "conv1")                     # :230
        x2 = self.conv_block(x1, "down1", 2)                      # :231
        x3 = self.conv_block(x2, "down2", 2)                      # :232
        x = self.conv_block(x3, "down3", 2)                       # :233
        h, w = x.shape[1], x.shape[2]
        uv_s = resize_bilinear(uv, (h, w))                        # :237
        x = torch.cat([x, uv_s], dim=3)                           # :238
        if probes is not None:
            probes.update(x1=x1, x2=x2, x3=x3, x0=x)
        for i in range(self.n_res // 2):                          # :239-240
            x = self.res_bottleneck(x, i, probes)
            if probes is not None:
                probes["res%d" % i] = x
        y = self.convt_block(x, "up1")                            # :243
        if probes is not None:
            probes["up1"] = y
        y = self.convt_block(torch.cat([y, x3], dim=3), "up2")    # :244
        if probes is not None:
            probes["up2"] = y
        y = self.convt_block(torch.cat([y, x2], dim=3), "up3")    # :245
        mask = torch.tanh(self.conv_block(y, "conv2", 1, bn=False, act=False))   # :246
        con = self.conv_block(y, "conv3", 1, bn=False, act=False)                # :247
        g0 = rgb_to_grayscale(inputs)
        gs = g0 * (1 + mask) + con                                # :250
        dif = gs - g0                                             # :251
        mask22 = torch.cat([torch.relu(mask), mask * 0, torch.relu(-mask)], dim=3)   # :252
        d32 = resize_bilinear(dif, (h, w))
        bmask = (d32 > BMASK_THRESHOLD).to(torch.float32)         # :256 (strict >)
        if probes is not None:
            probes.update(y=y, mask=mask, con=con, d32=d32, bmask=bmask)
        if bmask_override is not None:
            bmask = _t(bmask_override).reshape(bmask.shape)
        x_hole = x * (1 - bmask)                                  # :258
        x = torch.cat([x_hole, bmask, uv_s], dim=3)               # :259
        for i in range(self.n_res // 2, self.n_res):              # :261-262
            x = self.res_bottleneck(x, i, probes)
            if probes is not None:
                probes["res%d" % i] = x
        f = self.convt_block(x, "clr_up1")                        # :264
        f = self.convt_block(f, "clr_up2")                        # :265
        f = self.convt_block(f, "clr_up3")                        # :266
        if probes is not None:
            probes["f"] = f
        c = self.conv_block(torch.cat([gs, f], dim=3), "clr_conv1")              # :267
        c = self.conv_block(c, "clr_conv2")                       # :268
        con_rgb = self.conv_block(c, "clr_conv3", bn=False, act=False)           # :269
        dif2 = rgb_to_grayscale(con_rgb) - rgb_to_grayscale(inputs)              # :288
        return gs, con_rgb, mask22, dif2                          # :290

    __call__ = forward


# ---------------------------------------------------------------------------------------------------
# TSM variant (BASELINE config 5): ShareLayer + UV/offset warp.  /root/reference/model_with_TSM.py:199-325, warp.py:71-165
# ---------------------------------------------------------------------------------------------------
def batch_map_coordinates(x: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """``tf_batch_map_coordinates`` (warp.py:71-115): x [B,S,S,C], coords [B,N,2] (axis-0, axis-1) -> [B,N,C].
    Clamp to [0,S-1]; corners lt = floor, rb = ceil; lerp along axis 0 first (``vals_t``/``vals_b``), then along axis 1."""
    B, S = x.shape[0], x.shape[1]
    coords = coords.clamp(0, S - 1)
    lt = torch.floor(coords).long()
    rb = torch.ceil(coords).long()
    bi = torch.arange(B).reshape(B, 1).expand(B, coords.shape[1])

    def g(i0, i1):
        return x[bi, i0, i1]                                  # [B,N,C]
    v_lt, v_rb = g(lt[..., 0], lt[..., 1]), g(rb[..., 0], rb[..., 1])
    v_lb, v_rt = g(lt[..., 0], rb[..., 1]), g(rb[..., 0], lt[..., 1])
    off = coords - lt.to(coords.dtype)
    o0, o1 = off[..., 0:1], off[..., 1:2]
    vals_t = v_lt + (v_rt - v_lt) * o0
    vals_b = v_lb + (v_rb - v_lb) * o0
    return vals_t + (vals_b - vals_t) * o1


def batch_map_offsets(x: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """``tf_batch_map_offsets`` (warp.py:134-165): offsets [B,H,W,>=2] are resized to SxS, scaled by S, channels 0:2 kept,
    added to the 'ij' grid and sampled bilinearly."""
    B, S = x.shape[0], x.shape[1]
    off = resize_bilinear(offsets, (S, S)) * S
    off = off[..., 0:2].reshape(B, -1, 2)
    ii, jj = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    grid = torch.stack([ii, jj], dim=-1).to(torch.float32).reshape(1, -1, 2)
    return batch_map_coordinates(x, off + grid).reshape(B, S, S, -1)


def share_layer(x: torch.Tensor, reg: torch.Tensor, frame: int, share: bool = True) -> torch.Tensor:
    """``ShareLayer.call`` (model_with_TSM.py:204-229).  The reference reshapes to [1, frame, ...] (batch == frame); groups of
    ``frame`` consecutive images generalise that to batch = k * frame."""
    if not share:
        return torch.cat([x, x], dim=3)
    reg_in, reg_out = torch.split(reg, reg.shape[3] // 2, dim=3)
    x_reg = batch_map_offsets(x, reg_in)
    B, w, h, ch = x_reg.shape
    xr = x_reg.reshape(B // frame, frame, w, h, ch)
    sh = torch.cat([xr.max(dim=1).values, xr.mean(dim=1)], dim=3)           # [G,w,h,2ch]
    sh = sh.unsqueeze(1).expand(B // frame, frame, w, h, 2 * ch).reshape(B, w, h, 2 * ch)
    return batch_map_offsets(sh, reg_out)


class GeneratorTSMOracle(GeneratorOracle):
    """``Generator`` of /root/reference/model_with_TSM.py:231-325 at ``training=False``."""

    def forward(self, inputs, uv, reg, frame, share=True, chuck=1, training=False, probes: Optional[dict] = None,
                bmask_override: Optional[torch.Tensor] = None):
        assert not training
        inputs, uv, reg = _t(inputs), _t(uv), _t(reg)
        x1 = self.conv_block(inputs, "conv1")
        x2 = self.conv_block(x1, "down1", 2)
        x3 = self.conv_block(x2, "down2", 2)
        x = self.conv_block(x3, "down3", 2)
        h, w = x.shape[1], x.shape[2]
        uv_s = resize_bilinear(uv, (h, w))                                     # :269
        x_share = share_layer(x, reg, frame, share)                           # :271
        x = torch.cat([x, x_share, uv_s], dim=3)                              # :272
        if probes is not None:
            probes.update(x_share1=x_share, x0=x)
        for i in range(self.n_res // 2):
            x = self.res_bottleneck(x, i, probes)
        if probes is not None:
            probes["res2"] = x
        y = self.convt_block(x, "up1")
        y = self.convt_block(torch.cat([y, x3], dim=3), "up2")
        y = self.convt_block(torch.cat([y, x2], dim=3), "up3")
        mask = torch.tanh(self.conv_block(y, "conv2", 1, bn=False, act=False))
        con = self.conv_block(y, "conv3", 1, bn=False, act=False)
        g0 = rgb_to_grayscale(inputs)
        gs = g0 * (1 + mask) + con
        dif = gs - g0
        mask22 = torch.cat([torch.relu(mask), mask * 0, torch.relu(-mask)], dim=3)
        d32 = resize_bilinear(dif, (h, w))
        bmask = (d32 > BMASK_THRESHOLD).to(torch.float32)                     # :289
        if probes is not None:
            probes.update(d32=d32, bmask=bmask)
        if bmask_override is not None:
            bmask = _t(bmask_override).reshape(bmask.shape)
        x_hole = x * (1 - bmask)                                              # :291
        x_share = share_layer(x_hole, reg, frame, share)                      # :292
        x = torch.cat([x_hole, bmask, x_share, uv_s], dim=3)                  # :293
        if probes is not None:
            probes.update(x_share2=x_share)
        for i in range(self.n_res // 2, self.n_res):
            x = self.res_bottleneck(x, i, probes)
        if probes is not None:
            probes["res5"] = x
        f = self.convt_block(x, "clr_up1")
        f = self.convt_block(f, "clr_up2")
        f = self.convt_block(f, "clr_up3")
        c = self.conv_block(torch.cat([gs, f], dim=3), "clr_conv1")
        c = self.conv_block(c, "clr_conv2")
        con_rgb = self.conv_block(c, "clr_conv3", bn=False, act=False)
        dif2 = rgb_to_grayscale(con_rgb) - rgb_to_grayscale(inputs)
        return gs, con_rgb, mask22, dif2

    __call__ = forward


def test_step_ffhq(gen, img16: torch.Tensor):
    """Restatement of ``FSRNet.test_step_FFHQ`` (/root/reference/train_test_GSC.py:863-890) for a
    ``[N,256,256,16]`` packed tensor: split [3,3,3,6,1], generator, ``mask_pred*face``,
    ``clip(con_rgb,0,1)``; returns figs ``[img, deshadow_img_c, mask_pred*2]``."""
    img16 = _t(img16)
    img, gt, uv, reg, face = torch.split(img16, [3, 3, 3, 6, 1], dim=3)
    _, con_rgb, _, mask_pred = gen(img, uv, reg, chuck=1, training=False)
    mask_pred = mask_pred * face
    con_rgb = torch.clamp(con_rgb, 0, 1)
    return [img, con_rgb, mask_pred * 2]
