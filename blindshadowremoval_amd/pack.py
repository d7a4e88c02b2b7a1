"""Offline weight preparation for libbsr_hip: BatchNorm folding + MFMA-friendly packing.

Input: the reference's generator variables by their checkpoint names (HWIO ``Conv2D`` kernels,
``[kh,kw,Cout,Cin]`` ``Conv2DTranspose`` kernels, BatchNormalization gamma/beta/moving stats —
/root/reference/model.py:115-177, 81-113, 6-61; names per blindshadowremoval_amd/weights.py).

Output: one blob (bytes) that ``bsr_create`` uploads as is.  Per MFMA conv layer the weights become
``[chunk][tap][n_pad][CC+4]`` float32 — the exact LDS image the kernel stages per (chunk, tap) step,
including the 4-float bank pad — and a ``[n_pad]`` bias.  Every BatchNormalization on the path
follows a conv and runs with ``training=False`` (/root/reference/train_test_GSC.py:404,856), so it
folds exactly:  s = gamma * rsqrt(var + 1e-3);  W' = W * s;  b' = (b - mean) * s + beta.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from .weights import BN_EPS, N_RES, check_weights, detect_variant

BLOB_MAGIC = 0x57525342   # "BSRW"
BLOB_VERSION = 1
_ENTRY = struct.Struct("<40sQQ4i")
_HEADER = struct.Struct("<4I")

TRANSPOSED = ("up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3")

DTYPES = {"f32": 0, "f16": 1, "f32x3": 2}     # BSR_DTYPE_* of include/bsr_hip.h
# layers the 16-bit modes run on igemm_h16_kernel (csrc/igemm_h16.h): every 3x3 / stride-2 3x3 / transposed 3x3 igemm layer
H16_LAYERS = ("down1", "down2", "down3", "up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3") + tuple("res%d.conv2" % i for i in range(6))
# f16 mode: the stride-1 3x3 and the transposed 3x3 layers run on conv3_f16_kernel (csrc/conv3_f16.h) from their `w3` images
W3_LAYERS = ("up1", "up2", "up3", "clr_up1", "clr_up2", "clr_up3") + tuple("res%d.conv2" % i for i in range(6))
# 1x1 layers (igemm_h16_kernel<1,1> / gemm_nloop_kernel<.., H = 2>): split-precision (hi + lo planes) in BOTH 16-bit modes
X3_LAYERS = tuple("res%d.%s" % (i, n) for i in range(6) for n in ("conv1", "c3q", "w")) + ("heads", "clr_conv1", "conv1")      # + conv_n16_kernel / stem7_kernel<.., H = 2>


def geometry(variant: str = "gsc", dtype: str = "f32") -> Dict[str, Tuple[int, int, int]]:
    """name -> (CC, k_pad, n_pad): must match the launch table in csrc/bsr_api.hip.  The TSM variant
    (/root/reference/model_with_TSM.py) only widens the K of the layers fed by the ShareLayer concats: 291 -> 312, 877 -> 888
    (320 / 896 in the 16-bit modes).
    The 16-bit modes use 32-channel K chunks, so the 257 / 261-wide trunk tensors get stride 288 instead of 264 (kGSC16)."""
    tsm = variant == "tsm"
    h16 = dtype != "f32"
    if tsm:
        k_a, k_r, k_h = (320, 320, 896) if h16 else (312, 312, 888)
    else:
        k_a, k_r, k_h = (128, 288, 288) if h16 else (120, 264, 264)
    cu = 32 if h16 else 24
    g: Dict[str, Tuple[int, int, int]] = {
        "conv1": (32, 32, 32) if h16 else (24, 24, 32), "down1": (16, 32, 64), "down2": (16, 64, 64), "down3": (16, 64, 96),
        "up1": (cu, k_r, 96), "up2": (32, 160, 64), "up3": (32, 128, 64), "heads": (32, 64, 16),
        "clr_up1": (cu, k_h, 128), "clr_up2": (32, 128, 96), "clr_up3": (32, 96, 64), "clr_conv1": (32, 64, 16),
    }
    for i in range(N_RES):
        g["res%d.conv1" % i] = (cu, k_a if i == 0 else (k_r if i < N_RES // 2 else k_h), 128)
        g["res%d.conv2" % i] = (32, 128, 128)
        g["res%d.c3q" % i] = (32, 128, 768)       # [y3: 257 real of 288 | theta|phi|g: 384 | 3 zero tiles of slack]
        g["res%d.w" % i] = (32, 128, 384)         # 257 real of 288 + 3 zero tiles of slack (gemm_nloop group reads)
    return g


GEOMETRY = geometry("gsc")


def fold_bn(kernel_tkn: np.ndarray, bias: np.ndarray, bn: Dict[str, np.ndarray] | None):
    """kernel_tkn: [taps, K, N] float64.  Returns folded (kernel, bias) in float64."""
    k = kernel_tkn.astype(np.float64)
    b = bias.astype(np.float64)
    if bn is None:
        return k, b
    s = bn["gamma"].astype(np.float64) / np.sqrt(bn["moving_variance"].astype(np.float64) + BN_EPS)
    return k * s[None, None, :], (b - bn["moving_mean"].astype(np.float64)) * s + bn["beta"].astype(np.float64)


def _bn(w: Dict[str, np.ndarray], stem: str) -> Dict[str, np.ndarray]:
    return {p: w[stem + "/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")}


def pack_taps(kernel_tkn: np.ndarray, bias: np.ndarray, cc: int, k_pad: int, n_pad: int):
    """[taps, K, N] -> ([k_pad/cc, taps, n_pad, cc+4] float32, [n_pad] float32), zero padded."""
    taps, k, n = kernel_tkn.shape
    assert k <= k_pad and n <= n_pad and k_pad % cc == 0
    full = np.zeros((taps, k_pad, n_pad), np.float64)
    full[:, :k, :n] = kernel_tkn
    arr = np.zeros((k_pad // cc, taps, n_pad, cc + 4), np.float32)
    arr[..., :cc] = full.reshape(taps, k_pad // cc, cc, n_pad).transpose(1, 0, 3, 2)
    b = np.zeros(n_pad, np.float32)
    b[:n] = bias
    return arr, b


def pack_taps_h16(kernel_tkn: np.ndarray, bias: np.ndarray, cc: int, k_pad: int, n_pad: int, nsplit: int, swizzle: bool = False):
    """[taps, K, N] -> the fp16 LDS image of csrc/igemm_h16.h as float32 words: [k_pad/cc, taps, n_pad, (nsplit*cc + 8) / 2].
    Row of output channel n: cc halves hi = fp16(w) | (nsplit == 2) cc halves lo = fp16(w - hi) | 8 halves of zero pad, where
    w is the fp32 value the fp32 path uses (so hi + lo reproduces it to ~2^-22)."""
    taps, k, n = kernel_tkn.shape
    assert k <= k_pad and n <= n_pad and k_pad % cc == 0 and cc % 16 == 0 and nsplit in (1, 2)
    full = np.zeros((taps, k_pad, n_pad), np.float64)
    full[:, :k, :n] = kernel_tkn.astype(np.float32)
    rows = full.reshape(taps, k_pad // cc, cc, n_pad).transpose(1, 0, 3, 2)          # [chunk, tap, n, cc]
    hi = rows.astype(np.float16)
    if not np.all(np.isfinite(hi)):
        raise ValueError("a folded weight exceeds the fp16 range (65504): this layer cannot run in the 16-bit modes")
    planes = [hi]
    if nsplit == 2:
        planes.append((rows - hi.astype(np.float64)).astype(np.float16))
    if swizzle:
        # unpadded 128-byte rows for the LDS-DMA-fed f32x3 layers (H16Cfg::SWZ in csrc/igemm_h16.h): the eight 16-byte slots
        # [hi k0-7, hi k8-15, hi k16-23, hi k24-31, lo ...] of row n are stored at slot ^ ((n >> 1) & 7)
        assert nsplit == 2 and cc == 32
        lin = np.concatenate(planes, axis=3).reshape(rows.shape[:3] + (8, 8))          # [chunk, tap, n, slot, 8 halves]
        img = np.empty_like(lin)
        for row in range(n_pad):
            sw = (row >> 1) & 7
            for slot in range(8):
                img[:, :, row, slot ^ sw] = lin[:, :, row, slot]
        img = np.ascontiguousarray(img.reshape(rows.shape[:3] + (64,)))
    else:
        planes.append(np.zeros(rows.shape[:3] + (8,), np.float16))
        img = np.ascontiguousarray(np.concatenate(planes, axis=3))                    # [chunk, tap, n, nsplit*cc + 8] halves
    arr = img.view(np.float32)                                                         # two halves per 32-bit word, little endian
    b = np.zeros(n_pad, np.float32)
    b[:n] = bias
    return arr, b


def pack_w3(kernel_tkn: np.ndarray, bias: np.ndarray, k_pad: int):
    """A 3x3 / transposed 3x3 layer ([9, K, N] folded) as the weight stream of csrc/conv3_f16.h (f16 mode): per 64-channel output block and
    32-channel K chunk one 36-KB run of nine tap images [64 rows n][32 halves k] — three 12-KB trios, each moved by LDS-DMA as it lies.
    Rows are unpadded (64 bytes); the 16-byte unit u = k / 8 of row n sits at position u ^ ((n >> 2) & 3), which makes a ds_read_b128 over
    16 different rows conflict-free.  Returns ([nblk * nchunk, 9, 64, 16] float32 words, [nblk * 64] bias)."""
    taps, k, n = kernel_tkn.shape
    assert taps == 9 and k <= k_pad and k_pad % 32 == 0
    nblk, nchunk = (n + 63) // 64, k_pad // 32
    full = np.zeros((9, k_pad, nblk * 64), np.float64)
    full[:, :k, :n] = kernel_tkn.astype(np.float32)
    hi = full.astype(np.float16)
    if not np.all(np.isfinite(hi)):
        raise ValueError("a folded weight exceeds the fp16 range (65504): this layer cannot run in the 16-bit modes")
    rows = hi.reshape(9, nchunk, 4, 8, nblk, 64).transpose(4, 1, 0, 5, 2, 3)          # [blk, chunk, tap, row, unit, 8 halves]
    img = np.empty_like(rows)
    for row in range(64):
        sw = (row >> 2) & 3
        for u in range(4):
            img[:, :, :, row, u ^ sw] = rows[:, :, :, row, u]
    arr = np.ascontiguousarray(img.reshape(nblk * nchunk, 9, 64, 32)).view(np.float32)     # [.., 16] words
    b = np.zeros(nblk * 64, np.float32)
    b[:n] = bias
    return arr, b


def pack_w4(kernel_tkn: np.ndarray, bias: np.ndarray):
    """The NonLocalBlock's `w` conv ([1, 128, N <= 288] folded) as the LDS image of the attention kernel's fused tail
    (csrc/attention_h16.h): 9 tiles of 32 output channels, row n = 512 bytes = 16 chunks of 8 halves hi | 16 chunks lo.  Chunk
    c = 4 dt + 2 p + h holds k = 32 dt + 16 p + 4 h + (j & 3) + 8 (j >> 2), j = 0..7 — the order in which registers 8p .. 8p + 7 of the
    O^T accumulator tile dt present the attention output as the A operand — and sits at chunk position c ^ (n & 15) (conflict-free
    ds_read_b128 over 16 different rows; the image goes to LDS by DMA as it lies).  Returns ([9, 1, 32, 128] float32 words, [288] bias)."""
    taps, k, n = kernel_tkn.shape
    assert taps == 1 and k == 128 and n <= 288
    full = np.zeros((128, 288), np.float64)
    full[:, :n] = kernel_tkn[0].astype(np.float32)
    hi = full.astype(np.float16)
    if not np.all(np.isfinite(hi)):
        raise ValueError("a folded weight exceeds the fp16 range (65504): this layer cannot run in the 16-bit modes")
    lo = (full - hi.astype(np.float64)).astype(np.float16)
    korder = np.array([[32 * dt + 16 * p + 4 * h + (j & 3) + 8 * (j >> 2) for j in range(8)]
                       for dt in range(4) for p in range(2) for h in range(2)])                      # [16 chunks, 8]
    img = np.zeros((288, 32, 8), np.float16)
    for row in range(288):
        sw = row & 15
        for c in range(16):
            img[row, c ^ sw] = hi[korder[c], row]
            img[row, 16 + (c ^ sw)] = lo[korder[c], row]
    arr = np.ascontiguousarray(img.reshape(9, 1, 32, 256)).view(np.float32)                          # [9, 1, 32, 128] words
    b = np.zeros(288, np.float32)
    b[:n] = bias
    return arr, b


def layer_matrices(w: Dict[str, np.ndarray]) -> "Dict[str, Tuple[np.ndarray, np.ndarray]]":
    """Folded [taps, K, N] kernels + biases (float64) of every MFMA layer, in kernel K/N order."""
    out: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}

    def hwio(stem):            # Conv2D kernel [kh,kw,ci,co] -> [kh*kw, ci, co]
        k = w[stem + "/kernel"]
        return k.reshape(k.shape[0] * k.shape[1], k.shape[2], k.shape[3])

    # stem 7x7x3: taps = ky, K = kx*3 + c   (im2row7_kernel layout)
    k = w["conv1/conv/kernel"]                                    # [7,7,3,32]
    out["conv1"] = fold_bn(k.reshape(7, 21, 32), w["conv1/conv/bias"], _bn(w, "conv1/bnorm"))
    for nm in ("down1", "down2", "down3"):
        out[nm] = fold_bn(hwio(nm + "/conv"), w[nm + "/conv/bias"], _bn(w, nm + "/bnorm"))
    for nm in TRANSPOSED:                                         # [3,3,co,ci] -> [9, ci, co]
        k = w[nm + "/conv/kernel"]
        out[nm] = fold_bn(k.reshape(9, k.shape[2], k.shape[3]).transpose(0, 2, 1), w[nm + "/conv/bias"], _bn(w, nm + "/bnorm"))
    # heads: taps = ky, K = c, N = kx*2 + head (head 0 = conv2/mask, 1 = conv3/con); bias applied in heads_post
    k2, k3 = w["conv2/conv/kernel"][..., 0], w["conv3/conv/kernel"][..., 0]      # [7,7,64]
    hk = np.stack([k2, k3], axis=-1)                              # [ky,kx,c,head]
    out["heads"] = (hk.transpose(0, 2, 1, 3).reshape(7, 64, 14).astype(np.float64), np.zeros(14))
    # clr_conv1: reference input cat[gs, f] (model.py:267): the 64 f channels go through the K = 64 MFMA loop,
    # the gs channel (reference channel 0) is a separate 9-tap K group ("clr_conv1.gs", see clr_gs_weights)
    k, b = fold_bn(hwio("clr_conv1/conv"), w["clr_conv1/conv/bias"], _bn(w, "clr_conv1/bnorm"))   # [9,65,16]
    out["clr_conv1"] = (k[:, 1:, :], b)
    for i in range(N_RES):
        st = "res_stack/%d/" % i
        out["res%d.conv1" % i] = fold_bn(hwio(st + "conv1"), w[st + "conv1/bias"], _bn(w, st + "bnorm1"))
        out["res%d.conv2" % i] = fold_bn(hwio(st + "conv2"), w[st + "conv2/bias"], _bn(w, st + "bnorm2"))
        # conv3 + bnorm3 -> y3 (257), then theta | phi | g = 1x1 convs of y3 with NO nonlinearity in between
        # (model.py:101-102, 33-46): composed offline into one K = 128 GEMM, N = [y3 (257, padded to 288) | q k v (3 x 128)].
        # theta | phi | g keep the query / key / value order of the attention kernel.
        k3, b3 = fold_bn(hwio(st + "conv3"), w[st + "conv3/bias"], _bn(w, st + "bnorm3"))            # [1,128,257], [257]
        qkv = np.concatenate([hwio(st + "non_local/" + n) for n in ("theta", "phi", "g")], axis=2).astype(np.float64)   # [1,257,384]
        qb = np.concatenate([w[st + "non_local/%s/bias" % n] for n in ("theta", "phi", "g")]).astype(np.float64)
        kc = np.zeros((1, 128, 672))
        bc = np.zeros(672)
        kc[0, :, :257] = k3[0]
        bc[:257] = b3
        kc[0, :, 288:] = k3[0] @ qkv[0]
        bc[288:] = b3 @ qkv[0] + qb
        out["res%d.c3q" % i] = (kc, bc)
        out["res%d.w" % i] = fold_bn(hwio(st + "non_local/w"), w[st + "non_local/w/bias"], _bn(w, st + "non_local/bnorm"))
    return out


def clr_gs_weights(w: Dict[str, np.ndarray]) -> np.ndarray:
    """[16 n][16 k] float32: BN-folded clr_conv1 weights of the gs input channel, k = 3x3 tap index (k >= 9 zero)."""
    k, _ = fold_bn(w["clr_conv1/conv/kernel"].reshape(9, 65, 16), w["clr_conv1/conv/bias"], _bn(w, "clr_conv1/bnorm"))
    out = np.zeros((16, 16), np.float32)
    out[:, :9] = k[:, 0, :].T
    return out


def tail_weights(w: Dict[str, np.ndarray]) -> np.ndarray:
    """clr_conv2 (folded, k-major [16][16]) | bias[16] | clr_conv3 [16][3] | bias[3] for the fused tail of conv_n16_kernel (csrc/conv_n16.h)."""
    k2, b2 = fold_bn(w["clr_conv2/conv/kernel"].reshape(1, 16, 16), w["clr_conv2/conv/bias"], _bn(w, "clr_conv2/bnorm"))
    k3 = w["clr_conv3/conv/kernel"].reshape(16, 3).astype(np.float64)
    return np.concatenate([k2.reshape(-1), b2, k3.reshape(-1), w["clr_conv3/conv/bias"].astype(np.float64)]).astype(np.float32)


def pack_generator(weights: Dict[str, np.ndarray], dtype: str = "f32") -> bytes:
    """reference-named variables -> blob for ``bsr_create(..., dtype)`` (the blob header records the dtype it was packed for)."""
    if dtype not in DTYPES:
        raise ValueError("dtype must be one of %s" % sorted(DTYPES))
    variant = detect_variant(weights)
    check_weights(weights, variant)
    geo = geometry(variant, dtype)
    entries: List[Tuple[str, np.ndarray, Tuple[int, int, int, int]]] = []
    for name, (k, b) in layer_matrices(weights).items():
        cc, k_pad, n_pad = geo[name]
        if dtype != "f32" and name in H16_LAYERS:
            arr, bias = pack_taps_h16(k, b, cc, k_pad, n_pad, 2 if dtype == "f32x3" else 1, swizzle=(dtype == "f32x3" and cc == 32))
        elif dtype != "f32" and name in X3_LAYERS:
            arr, bias = pack_taps_h16(k, b, cc, k_pad, n_pad, 2)
        else:
            arr, bias = pack_taps(k, b, cc, k_pad, n_pad)
        entries.append((name + ".w", arr, tuple(arr.shape)))
        entries.append((name + ".b", bias, (n_pad, 0, 0, 0)))
        if dtype == "f16" and name in W3_LAYERS:
            arr3, bias3 = pack_w3(k, b, k_pad)
            entries.append((name + ".w3.w", arr3, tuple(arr3.shape)))
            entries.append((name + ".w3.b", bias3, (bias3.shape[0], 0, 0, 0)))
        if dtype != "f32" and name.startswith("res") and name.endswith(".w"):
            arr4, bias4 = pack_w4(k, b)
            entries.append((name + "4.w", arr4, tuple(arr4.shape)))
            entries.append((name + "4.b", bias4, (288, 0, 0, 0)))
    entries.append(("heads.bias", np.array([weights["conv2/conv/bias"][0], weights["conv3/conv/bias"][0]], np.float32), (2, 0, 0, 0)))
    entries.append(("clr_conv1.gs", clr_gs_weights(weights), (16, 16, 0, 0)))
    entries.append(("tail.w", tail_weights(weights), (323, 0, 0, 0)))

    off = _HEADER.size + _ENTRY.size * len(entries)
    off = (off + 255) & ~255
    table = bytearray()
    chunks = []
    for name, arr, dims in entries:
        raw = np.ascontiguousarray(arr, dtype="<f4").tobytes()
        table += _ENTRY.pack(name.encode(), off, arr.size, *dims)
        chunks.append((off, raw))
        off = (off + len(raw) + 255) & ~255
    blob = bytearray(off)
    blob[:_HEADER.size] = _HEADER.pack(BLOB_MAGIC, BLOB_VERSION, len(entries), DTYPES[dtype])
    blob[_HEADER.size:_HEADER.size + len(table)] = table
    for o, raw in chunks:
        blob[o:o + len(raw)] = raw
    return bytes(blob)
