"""Host-side weight preparation: BN folding and the MFMA packing are checked on CPU by emulating, in
numpy, exactly what the kernels compute from the packed arrays, against the oracle's unfolded
conv + BatchNorm on the same inputs."""
import struct

import numpy as np
import pytest
import torch

from blindshadowremoval_amd import pack
from blindshadowremoval_amd.weights import init_weights
from oracle import gsc_oracle as O


@pytest.fixture(scope="module")
def w():
    return init_weights(5)


def unpack(arr):
    """[chunk][tap][n_pad][cc+4] -> dense [tap][k_pad][n_pad] (what the kernel multiplies by)."""
    nch, taps, n_pad, ldp = arr.shape
    assert np.all(arr[..., ldp - 4:] == 0)
    return arr[..., :ldp - 4].transpose(1, 0, 3, 2).reshape(taps, nch * (ldp - 4), n_pad)


def packed(w, name):
    k, b = pack.layer_matrices(w)[name]
    cc, k_pad, n_pad = pack.GEOMETRY[name]
    arr, bias = pack.pack_taps(k, b, cc, k_pad, n_pad)
    return unpack(arr), bias


def bn_args(w, stem):
    return [w[stem + "/" + p] for p in ("gamma", "beta", "moving_mean", "moving_variance")]


def test_bn_fold_matches_conv_then_bn(w):
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((1, 8, 8, 32)).astype(np.float32))
    ref = O.batchnorm_infer(O.conv2d_same(x, w["down1/conv/kernel"], w["down1/conv/bias"], 2), *bn_args(w, "down1/bnorm"))
    k, bias = packed(w, "down1")                                     # [9, 32, 64]
    got = O.conv2d_same(x, k[:, :32, :].reshape(3, 3, 32, 64), bias[:64], 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)


def test_stem_im2row_equivalence(w):
    """conv1 7x7x3 == 7 row taps over 24-wide windows of the raw NHWC row (21 contiguous floats + 3 zero-weight columns):
    the address pattern stem7_kernel reads from its LDS tile, emulated here with an explicit row expansion."""
    rng = np.random.default_rng(1)
    x = rng.random((1, 12, 16, 3)).astype(np.float32)
    ref = O.GeneratorOracle(w).conv_block(torch.from_numpy(x), "conv1").numpy()
    xr = np.zeros((1, 12, 16, 24), np.float32)
    for kx in range(7):
        for c in range(3):
            sx = np.arange(16) + kx - 3
            ok = (sx >= 0) & (sx < 16)
            xr[0, :, ok, kx * 3 + c] = x[0, :, sx[ok], c]
    k, bias = packed(w, "conv1")                                     # [7, 24, 32]
    got = O.leaky_relu(O.conv2d_same(torch.from_numpy(xr), k.reshape(7, 1, 24, 32), bias, 1)).numpy()
    np.testing.assert_allclose(got, ref, atol=2e-5)


def test_heads_decomposition(w):
    """conv2/conv3 (7x7, 64->1 each) == 7x1 conv to N=(kx,head) followed by a 7-tap horizontal sum."""
    rng = np.random.default_rng(2)
    y = rng.standard_normal((1, 10, 12, 64)).astype(np.float32)
    g = O.GeneratorOracle(w)
    ref_m = g.conv_block(torch.from_numpy(y), "conv2", bn=False, act=False).numpy()[0, ..., 0]
    ref_c = g.conv_block(torch.from_numpy(y), "conv3", bn=False, act=False).numpy()[0, ..., 0]
    k, bias = packed(w, "heads")                                     # [7, 64, 16]
    assert np.all(bias == 0)
    assert np.all(k[:, :, 14:] == 0)
    q = O.conv2d_same(torch.from_numpy(y), k.reshape(7, 1, 64, 16), bias, 1).numpy()
    m = np.zeros((10, 12)); c = np.zeros((10, 12))
    for kx in range(7):
        for x in range(12):
            sx = x + kx - 3
            if 0 <= sx < 12:
                m[:, x] += q[0, :, sx, kx * 2]
                c[:, x] += q[0, :, sx, kx * 2 + 1]
    np.testing.assert_allclose(m + w["conv2/conv/bias"][0], ref_m, atol=2e-5)
    np.testing.assert_allclose(c + w["conv3/conv/bias"][0], ref_c, atol=2e-5)


def test_transposed_phase_decomposition(w):
    """ConvT(3, s2) == 4 output-parity phases; tap (a,b) feeds phase (a==1, b==1) from x[i-(a==2), j-(b==2)]."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 4, 5, 96)).astype(np.float32)
    ref = O.GeneratorOracle(w).convt_block(torch.from_numpy(x), "clr_up3").numpy()
    k, bias = packed(w, "clr_up3")                                   # [9, 96, 64]
    out = np.zeros((1, 8, 10, 64))
    xp = np.pad(x, ((0, 0), (1, 0), (1, 0), (0, 0)))
    for a in range(3):
        for b in range(3):
            di, dj = (0 if a == 2 else 1), (0 if b == 2 else 1)
            out[0, (a == 1)::2, (b == 1)::2] += xp[0, di:di + 4, dj:dj + 5] @ k[a * 3 + b, :96, :64]
    got = O.leaky_relu(torch.from_numpy(out + bias[:64])).numpy()
    np.testing.assert_allclose(got, ref, atol=2e-5)


def test_clr_conv1_gs_split_and_qkv_order(w):
    """clr_conv1 over cat[gs, f] == 64-channel conv over f + a 9-tap im2col group over gs (conv_n16_kernel GS path)."""
    rng = np.random.default_rng(4)
    gs = rng.standard_normal((1, 6, 6, 1)).astype(np.float32)
    f = rng.standard_normal((1, 6, 6, 64)).astype(np.float32)
    ref = O.GeneratorOracle(w).conv_block(torch.from_numpy(np.concatenate([gs, f], -1)), "clr_conv1").numpy()
    k, bias = packed(w, "clr_conv1")                                 # [9, 64, 16]
    wg = pack.clr_gs_weights(w)                                      # [16 n][16 k]
    assert k.shape == (9, 64, 16) and wg.shape == (16, 16) and np.all(wg[:, 9:] == 0)
    main = O.conv2d_same(torch.from_numpy(f), k.reshape(3, 3, 64, 16), bias, 1).numpy()
    gsp = np.pad(gs[0, :, :, 0], 1)
    extra = np.zeros((6, 6, 16))
    for t in range(9):
        extra += gsp[t // 3:t // 3 + 6, t % 3:t % 3 + 6, None] * wg[None, None, :, t]
    got = O.leaky_relu(torch.from_numpy(main[0] + extra)).numpy()
    np.testing.assert_allclose(got, ref[0], atol=2e-5)
    # conv3 + theta|phi|g composed into one K = 128 GEMM: N = [y3 (257 of 288) | q k v (384)]
    k, bias = packed(w, "res2.c3q")                                  # [1, 128, 672]
    st = "res_stack/2/"
    rng = np.random.default_rng(6)
    t2 = torch.from_numpy(rng.standard_normal((1, 4, 4, 128)).astype(np.float32))
    y3 = O.batchnorm_infer(O.conv2d_same(t2, w[st + "conv3/kernel"], w[st + "conv3/bias"]), *bn_args(w, st + "bnorm3"))
    got = O.conv2d_same(t2, k[:, :, :672].reshape(1, 1, 128, 672), bias[:672], 1).numpy()
    assert np.all(k[:, :, 672:] == 0) and np.all(bias[672:] == 0)
    np.testing.assert_allclose(got[..., :257], y3.numpy(), atol=2e-5)
    assert np.all(got[..., 257:288] == 0)
    for j, n in enumerate(("theta", "phi", "g")):                    # query, key, value order of the attention kernel
        ref = O.conv2d_same(y3, w[st + "non_local/" + n + "/kernel"], w[st + "non_local/" + n + "/bias"]).numpy()
        np.testing.assert_allclose(got[..., 288 + 128 * j:288 + 128 * (j + 1)], ref, atol=2e-5)


def test_blob_layout(w):
    blob = pack.pack_generator(w)
    magic, ver, n, _ = struct.unpack_from("<4I", blob, 0)
    assert magic == pack.BLOB_MAGIC and ver == pack.BLOB_VERSION
    names = {}
    for i in range(n):
        name, off, nfl, d0, d1, d2, d3 = struct.unpack_from("<40sQQ4i", blob, 16 + 72 * i)
        name = name.rstrip(b"\0").decode()
        assert off % 16 == 0 and off + 4 * nfl <= len(blob)
        names[name] = (off, nfl, (d0, d1, d2, d3))
    assert len(names) == n == 2 * len(pack.GEOMETRY) + 3
    off, nfl, dims = names["res0.conv1.w"]
    assert dims == (5, 1, 128, 28) and nfl == 5 * 128 * 28
    off, nfl, dims = names["up1.w"]
    assert dims == (11, 9, 96, 28)
    hb = np.frombuffer(blob, "<f4", 2, names["heads.bias"][0])
    assert hb[0] == w["conv2/conv/bias"][0] and hb[1] == w["conv3/conv/bias"][0]
    with pytest.raises(ValueError):
        bad = dict(w); bad.pop("up1/conv/kernel"); pack.pack_generator(bad)
    with pytest.raises(ValueError):
        bad = dict(w); bad["up1/conv/kernel"] = bad["up1/conv/kernel"].transpose(0, 1, 3, 2); pack.pack_generator(bad)


def test_h16_pack_planes_reproduce_the_fp32_weights():
    """16-bit modes (csrc/igemm_h16.h): row = [cc halves hi | cc halves lo | 8 halves pad]; hi + lo must give back the fp32 weight
    to ~2^-22 relative (f32x3), hi alone to 2^-11 (f16); the blob header records the dtype."""
    import struct
    from blindshadowremoval_amd import pack
    rng = np.random.default_rng(3)
    k = rng.standard_normal((9, 40, 70)) * 0.05
    b = rng.standard_normal(70)
    arr, bias = pack.pack_taps_h16(k, b, 32, 64, 96, 2)
    assert arr.shape == (2, 9, 96, 36) and arr.dtype == np.float32 and bias.shape == (96,)
    halves = arr.view(np.float16).reshape(2, 9, 96, 72)
    hi, lo, pad = halves[..., :32].astype(np.float64), halves[..., 32:64].astype(np.float64), halves[..., 64:]
    assert not pad.any()
    full = np.zeros((9, 64, 96))
    full[:, :40, :70] = k.astype(np.float32)
    want = full.reshape(9, 2, 32, 96).transpose(1, 0, 3, 2)
    assert np.abs(hi + lo - want).max() <= 2.0 ** -21 * np.abs(want).max()
    assert np.abs(hi - want).max() <= 2.0 ** -11 * np.abs(want).max()
    arr1, _ = pack.pack_taps_h16(k, b, 32, 64, 96, 1)
    assert arr1.shape == (2, 9, 96, 20)
    assert np.array_equal(arr1.view(np.float16).reshape(2, 9, 96, 40)[..., :32], halves[..., :32])
    # swizzled unpadded rows of the DMA-fed f32x3 layers: slot s of row n sits at slot s ^ ((n >> 1) & 7), nothing else changes
    arrs, _ = pack.pack_taps_h16(k, b, 32, 64, 96, 2, swizzle=True)
    assert arrs.shape == (2, 9, 96, 32)
    hs = arrs.view(np.float16).reshape(2, 9, 96, 8, 8)
    for n in (0, 1, 2, 37, 95):
        for slot in range(8):
            assert np.array_equal(hs[:, :, n, slot ^ ((n >> 1) & 7)], halves[:, :, n, 8 * slot:8 * slot + 8])
    w = init_weights(1)
    for dtype, code in pack.DTYPES.items():
        blob = pack.pack_generator(w, dtype)
        assert struct.unpack_from("<4I", blob)[3] == code
    assert struct.unpack_from("<4I", pack.pack_generator(init_weights(1, variant="tsm"), "f32x3"))[3] == pack.DTYPES["f32x3"]      # TSM packs in every dtype
    with pytest.raises(ValueError):
        pack.pack_taps_h16(k * 1e7, b, 32, 64, 96, 2)


def test_w4_image_is_the_attention_tails_operand(w):
    """pack_w4 (round 6): the NonLocalBlock's `w` conv as the LDS image of csrc/attention_h16.h's fused tail.  Emulate the kernel's reads: lane
    (r = n & 31, h) of K step ks reads the 16-byte chunk at position (2 ks + h) ^ (n & 15) of row n (hi plane; lo plane 16 chunks further) and
    multiplies it with registers 8p .. 8p + 7 of O^T tile dt (ks = 2 dt + p), i.e. channels 32 dt + 16 p + 4 h + (j & 3) + 8 (j >> 2): summed over
    ks, h, j that must be the plain GEMM with the folded weights, to the 2^-22 of the hi + lo split."""
    k, b = pack.layer_matrices(w)["res2.w"]
    arr, bias = pack.pack_w4(k, b)
    assert arr.shape == (9, 1, 32, 128) and bias.shape == (288,) and np.array_equal(bias[:257], b.astype(np.float32)) and not bias[257:].any()
    img = np.ascontiguousarray(arr).view(np.float16).reshape(288, 32, 8).astype(np.float64)          # [row n][chunk position][8 halves]
    rng = np.random.default_rng(3)
    att = rng.standard_normal(128)
    got = np.zeros(288)
    for n in range(288):
        for ks in range(8):
            dt, p = ks >> 1, ks & 1
            for h in range(2):
                pos = (2 * ks + h) ^ (n & 15)
                wk = img[n, pos] + img[n, 16 + pos]                                                   # hi + lo
                ch = [32 * dt + 16 * p + 4 * h + (j & 3) + 8 * (j >> 2) for j in range(8)]
                got[n] += float(np.dot(wk, att[ch]))
    want = np.zeros(288)
    want[:257] = att @ k[0].astype(np.float32).astype(np.float64)
    assert np.abs(got - want).max() < 1e-5 * max(1.0, np.abs(want).max())
    # every 16-lane group of a ds_read_b128 (16 different rows, one K step) touches 16 different chunk positions mod 16: conflict-free
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for c in range(16):
            assert len({(c ^ (r & 15)) & 15 for r in grp}) == 16


def test_16_bit_blobs_carry_the_w4_images(w):
    blob = pack.pack_generator(w, "f32x3")
    _, _, n_entries, _ = struct.unpack_from("<4I", blob, 0)
    names = [struct.unpack_from("<40sQQ4i", blob, 16 + 72 * i)[0].split(b"\0")[0].decode() for i in range(n_entries)]
    for i in range(6):
        assert "res%d.w4.w" % i in names and "res%d.w4.b" % i in names and "res%d.w.w" % i in names
    names32 = pack.pack_generator(w, "f32")
    assert b"res0.w4.w" not in names32


def test_w3_stream_is_the_f16_conv_kernels_operand(w):
    """pack_w3 (round 6): a 3x3 / transposed 3x3 layer as the weight stream of csrc/conv3_f16.h.  Emulate the kernel's B-fragment read — lane
    (r, h) of block row ni * 32 + r reads the 16-byte unit at position (2 g + h) ^ ((r >> 2) & 3) of its 64-byte row and uses it as
    k = 16 g + 8 h + j — and check it returns fp16(W[tap][32 chunk + k][64 blk + row]) for every (block, chunk, tap, row, k); the padded
    block of a 96-channel layer is zero."""
    k, b = pack.layer_matrices(w)["clr_up2"]                      # [9, 128, 96]
    arr, bias = pack.pack_w3(k, b, 128)
    assert arr.shape == (2 * 4, 9, 64, 16) and bias.shape == (128,) and not bias[96:].any()
    img = np.ascontiguousarray(arr).view(np.float16).reshape(2, 4, 9, 64, 4, 8)
    want = np.zeros((9, 128, 128), np.float16)
    want[:, :, :96] = k.astype(np.float32).astype(np.float16)
    for blk in range(2):
        for row in (0, 1, 5, 31, 32, 47, 63):
            sw = (row >> 2) & 3
            for g in range(2):
                for h in range(2):
                    got = img[blk, :, :, row, (2 * g + h) ^ sw]                              # [chunk, tap, 8]
                    ks = 16 * g + 8 * h + np.arange(8)
                    exp = np.stack([want[:, 32 * c + ks, 64 * blk + row] for c in range(4)])  # [chunk, tap, 8]
                    assert np.array_equal(got, exp)
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for u in range(4):                                             # 16 rows of 64 bytes, one unit each: 16 different 16-byte bank groups
            assert len({(r * 4 + (u ^ ((r >> 2) & 3))) & 15 for r in grp}) == 16
