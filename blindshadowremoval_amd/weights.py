"""Variable inventory of the GSC generator and a seeded synthetic initialiser.

The variable names and shapes are the ones ``tf.train.Checkpoint(generator=Generator())`` produces
for /root/reference/model.py:198-226 (layer table) — ``tests/golden/gsc_ckpt94_inventory.json`` holds
the same table parsed from the reference's own ``ckpt-94.index`` and the CPU test-suite checks the
two agree.  Kernel layouts are TensorFlow's: ``Conv2D`` HWIO ``[kh,kw,Cin,Cout]``
(/root/reference/model.py:119), ``Conv2DTranspose`` ``[kh,kw,Cout,Cin]`` (/root/reference/model.py:153).

The trained weights are not shipped with the reference (/root/reference/.MISSING_LARGE_BLOBS), so
benchmarks and parity tests use ``init_weights(seed)``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple  # noqa: F401 (Tuple: string annotations)

import numpy as np

N_RES = 6                       # /root/reference/model.py:199
N_CH = [32, 64, 64, 96, 128, 256, 256]   # /root/reference/model.py:201
RES_CH = N_CH[5] + 1            # 257, /root/reference/model.py:226
BN_EPS = 1e-3                   # Keras BatchNormalization default epsilon
LRELU_ALPHA = 0.3               # Keras LeakyReLU default alpha


def _conv(spec, stem, kh, cin, cout, bn, transpose=False):
    spec[stem + "/conv/kernel"] = (kh, kh, cout, cin) if transpose else (kh, kh, cin, cout)
    spec[stem + "/conv/bias"] = (cout,)
    if bn:
        for p in ("gamma", "beta", "moving_mean", "moving_variance"):
            spec[stem + "/bnorm/" + p] = (cout,)


def generator_variable_shapes(variant: str = "gsc") -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for the 258 float32 variables of the generator.  ``variant``: "gsc" (/root/reference/model.py) or
    "tsm" (/root/reference/model_with_TSM.py: ShareLayer widens the bottleneck inputs to 291 / 877 channels)."""
    if variant not in ("gsc", "tsm"):
        raise ValueError("variant must be 'gsc' or 'tsm'")
    tsm = variant == "tsm"
    c0 = N_CH[3] + 3 + (2 * N_CH[3] if tsm else 0)            # cat[x, (x_share,) uv]: 99 | 291
    c12 = max(c0, RES_CH)                                       # output width of res blocks 0-2: 257 | 291
    c3 = c12 + 1 + 3 + (2 * c12 if tsm else 0)                  # cat[x_hole, bmask, (x_share,) uv]: 261 | 877
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _conv(s, "conv1", 7, 3, N_CH[0], True)                 # model.py:203
    _conv(s, "conv2", 7, N_CH[1], 1, False)                # model.py:204 (mask head)
    _conv(s, "conv3", 7, N_CH[1], 1, False)                # model.py:205 (con head)
    _conv(s, "down1", 3, N_CH[0], N_CH[1], True)           # model.py:207
    _conv(s, "down2", 3, N_CH[1], N_CH[2], True)
    _conv(s, "down3", 3, N_CH[2], N_CH[3], True)
    _conv(s, "up1", 3, c12, N_CH[3], True, transpose=True)               # model.py:210,243
    _conv(s, "up2", 3, N_CH[3] + N_CH[2], N_CH[2], True, transpose=True)  # cat[y,x3] model.py:244
    _conv(s, "up3", 3, N_CH[2] + N_CH[1], N_CH[1], True, transpose=True)  # cat[y,x2] model.py:245
    _conv(s, "clr_up1", 3, c3, N_CH[4], True, transpose=True)             # 261 -> 128 model.py:214,264
    _conv(s, "clr_up2", 3, N_CH[4], N_CH[3], True, transpose=True)
    _conv(s, "clr_up3", 3, N_CH[3], N_CH[2], True, transpose=True)
    _conv(s, "clr_conv1", 3, N_CH[2] + 1, 16, True)        # cat[gs,f] model.py:217,267
    _conv(s, "clr_conv2", 1, 16, 16, True)
    _conv(s, "clr_conv3", 1, 16, 3, False)
    half = RES_CH // 2                                      # 128
    for i in range(N_RES):
        cin = c0 if i == 0 else (c12 if i < N_RES // 2 else c3)   # GSC 99 / 257 / 261, TSM 291 / 291 / 877
        st = "res_stack/%d/" % i
        for name, shp in (("conv1", (1, 1, cin, half)), ("conv2", (3, 3, half, half)), ("conv3", (1, 1, half, RES_CH))):
            s[st + name + "/kernel"] = shp
            s[st + name + "/bias"] = (shp[3],)
        for j, c in ((1, half), (2, half), (3, RES_CH)):
            for p in ("gamma", "beta", "moving_mean", "moving_variance"):
                s[st + "bnorm%d/%s" % (j, p)] = (c,)
        for name in ("g", "phi", "theta"):                  # model.py:10-12
            s[st + "non_local/%s/kernel" % name] = (1, 1, RES_CH, half)
            s[st + "non_local/%s/bias" % name] = (half,)
        s[st + "non_local/w/kernel"] = (1, 1, half, RES_CH)  # model.py:13
        s[st + "non_local/w/bias"] = (RES_CH,)
        for p in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[st + "non_local/bnorm/" + p] = (RES_CH,)
    return s


# Kernel variance gains (x 1/fan_in).  1.6 roughly preserves variance through LeakyReLU(0.3); the
# residual branches (``conv3``, ``non_local/w``) and the attention projections are damped so the six
# bottleneck blocks neither blow activations up nor saturate the 1024-wide softmax — a trained
# network keeps both O(1), and parity to 1e-3 absolute is only meaningful at that scale.
_GAINS = (("non_local/theta", 4.0), ("non_local/phi", 4.0), ("non_local/w", 0.15), ("/conv3/kernel", 0.45),
          ("clr_conv3", 0.25), ("clr_conv2", 1.0))


def _gain(name: str) -> float:
    if name.startswith("res_stack") or name.startswith("clr_conv"):
        for key, g in _GAINS:
            if key in name:
                return g
    return 1.6


def init_weights(seed: int = 1, con_bias_shift: float = 0.25, variant: str = "gsc") -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (SURVEY.md §8d recipe): kernels N(0, 1.6/fan_in), biases / beta /
    moving_mean N(0, 0.05^2), gamma U[0.8,1.2], moving_variance U[0.75,1.25].

    ``con_bias_shift`` is added to the ``conv3`` (con head) bias; +0.1 pushes ``dif`` across the
    in-network 0.1 threshold (/root/reference/model.py:256) so ``bmask`` is non-degenerate."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shp in generator_variable_shapes(variant).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            transpose = name.split("/")[0] in ("up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3")
            cin = shp[3] if transpose else shp[2]
            fan_in = shp[0] * shp[1] * cin
            if transpose:
                fan_in = fan_in / 4.0       # stride-2 transposed conv: 9/4 taps reach one output on average
            w = rng.standard_normal(shp) * np.sqrt(_gain(name) / fan_in)
        elif leaf == "gamma":
            w = rng.uniform(0.8, 1.2, shp)
        elif leaf == "moving_variance":
            w = rng.uniform(0.75, 1.25, shp)
        else:                               # bias, beta, moving_mean
            w = rng.standard_normal(shp) * 0.05
        out[name] = w.astype(np.float32)
    if con_bias_shift:
        out["conv3/conv/bias"] = (out["conv3/conv/bias"] + np.float32(con_bias_shift)).astype(np.float32)
    return out


def detect_variant(weights: Dict[str, np.ndarray]) -> str:
    k = weights.get("res_stack/0/conv1/kernel")
    return "tsm" if k is not None and k.shape[2] == 291 else "gsc"


def check_weights(weights: Dict[str, np.ndarray], variant: str = "gsc") -> None:
    """Raise ValueError if ``weights`` is not exactly the generator's variable set."""
    spec = generator_variable_shapes(variant)
    missing = [k for k in spec if k not in weights]
    if missing:
        raise ValueError("missing generator variables: %s%s" % (missing[:4], " ..." if len(missing) > 4 else ""))
    for k, shp in spec.items():
        if tuple(weights[k].shape) != tuple(shp):
            raise ValueError("variable %s has shape %s, expected %s" % (k, tuple(weights[k].shape), shp))
