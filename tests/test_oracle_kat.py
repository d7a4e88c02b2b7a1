"""Known-answer tests for every TF semantic the oracle asserts.

Each expected value is hand-computable from the definition quoted in the test."""
import numpy as np
import torch

from oracle import gsc_oracle as O
from oracle import np_loops as L


def _nhwc(a):
    return torch.tensor(np.asarray(a, np.float32))


def test_a1_same_padding_amounts():
    # k=7,s=1 -> 3/3 ; k=3,s=1 -> 1/1 ; k=3,s=2 even H -> 0 before / 1 after ; k=1 -> none
    assert O.same_pad(256, 7, 1) == (3, 3)
    assert O.same_pad(32, 3, 1) == (1, 1)
    assert O.same_pad(256, 3, 2) == (0, 1)
    assert O.same_pad(32, 1, 1) == (0, 0)


def test_a1_conv_stride2_pads_after_only():
    # 4x4 ramp, 3x3 all-ones kernel, stride 2: out[0,0] sums rows 0..2, cols 0..2 (NO top/left pad),
    # out[1,1] sums rows 2..4, cols 2..4 with row/col 4 being the zero pad.
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    k = np.ones((3, 3, 1, 1), np.float32)
    y = O.conv2d_same(_nhwc(x), k, np.zeros(1, np.float32), 2).numpy()[0, :, :, 0]
    img = x[0, :, :, 0]
    assert y.shape == (2, 2)
    assert y[0, 0] == img[0:3, 0:3].sum()
    assert y[0, 1] == img[0:3, 2:4].sum()
    assert y[1, 0] == img[2:4, 0:3].sum()
    assert y[1, 1] == img[2:4, 2:4].sum()


def test_a1_conv_is_cross_correlation_hwio():
    # a single 1 at (1,2) with an asymmetric 3x3 kernel: out[y,x] = k[1-y+1, 2-x+1] (no flip)
    x = np.zeros((1, 4, 4, 1), np.float32)
    x[0, 1, 2, 0] = 1
    k = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1)
    y = O.conv2d_same(_nhwc(x), k, np.zeros(1, np.float32), 1).numpy()[0, :, :, 0]
    for oy in range(4):
        for ox in range(4):
            a, b = 1 - oy + 1, 2 - ox + 1
            want = k[a, b, 0, 0] if 0 <= a < 3 and 0 <= b < 3 else 0
            assert y[oy, ox] == want
    # channels: HWIO means k[..., ci, co]
    x2 = np.zeros((1, 2, 2, 2), np.float32)
    x2[0, 0, 0] = [1, 10]
    k2 = np.zeros((1, 1, 2, 3), np.float32)
    k2[0, 0, 0] = [1, 2, 3]
    k2[0, 0, 1] = [4, 5, 6]
    y2 = O.conv2d_same(_nhwc(x2), k2, np.array([0.5, 0, 0], np.float32), 1).numpy()
    np.testing.assert_allclose(y2[0, 0, 0], [41.5, 52, 63])


def test_a2_transposed_conv_scatter_and_crop():
    # y[2i+a, 2j+b, o] += x[i,j,c] * W[a,b,o,c], rows/cols [0, 2H) kept
    x = np.zeros((1, 2, 2, 1), np.float32)
    x[0, 0, 0, 0] = 1
    x[0, 1, 1, 0] = 10
    k = (np.arange(9, dtype=np.float32) + 1).reshape(3, 3, 1, 1)
    y = O.conv2d_transpose_same(_nhwc(x), k, np.zeros(1, np.float32)).numpy()[0, :, :, 0]
    want = np.zeros((5, 5), np.float32)
    want[0:3, 0:3] += k[:, :, 0, 0]
    want[2:5, 2:5] += 10 * k[:, :, 0, 0]
    np.testing.assert_array_equal(y, want[:4, :4])
    # kernel layout is [kh,kw,Cout,Cin]
    x2 = np.zeros((1, 1, 1, 2), np.float32)
    x2[0, 0, 0] = [1, 10]
    k2 = np.zeros((3, 3, 3, 2), np.float32)
    k2[1, 1, :, 0] = [1, 2, 3]
    k2[1, 1, :, 1] = [4, 5, 6]
    y2 = O.conv2d_transpose_same(_nhwc(x2), k2, np.zeros(3, np.float32)).numpy()
    np.testing.assert_allclose(y2[0, 1, 1], [41, 52, 63])


def test_a2_transposed_conv_is_gradient_of_a1_conv():
    # TF defines conv2d_transpose as the input-gradient of conv2d with the same kernel/stride/padding
    torch.manual_seed(0)
    xin = torch.randn(1, 8, 8, 3, requires_grad=True)
    kern = torch.randn(3, 3, 3, 5)                       # HWIO for the forward conv: in=3, out=5
    yf = O.conv2d_same(xin, kern.numpy(), np.zeros(5, np.float32), 2)
    gy = torch.randn_like(yf)
    (gx,) = torch.autograd.grad(yf, xin, gy)
    # deconv input gy [1,4,4,5] -> output [1,8,8,3]; its kernel layout [kh,kw,Cout=3,Cin=5] == HWIO of the conv
    yt = O.conv2d_transpose_same(gy, kern.numpy(), np.zeros(3, np.float32))
    np.testing.assert_allclose(yt.numpy(), gx.numpy(), atol=1e-5)


def test_a3_batchnorm_inference_eps_1e3():
    x = _nhwc(np.array([[[[2.0, -1.0]]]]))
    y = O.batchnorm_infer(x, np.array([2.0, 0.5], np.float32), np.array([0.1, -0.2], np.float32),
                          np.array([1.0, 1.0], np.float32), np.array([0.999, 3.999], np.float32)).numpy()
    np.testing.assert_allclose(y[0, 0, 0], [(2 - 1) / 1.0 * 2 + 0.1, (-1 - 1) / 2.0 * 0.5 - 0.2], rtol=1e-6)


def test_a4_leaky_relu_alpha_03():
    y = O.leaky_relu(torch.tensor([-2.0, 0.0, 3.0])).numpy()
    np.testing.assert_allclose(y, [-0.6, 0.0, 3.0], rtol=1e-6)


def test_a5_resize_8x_is_centre_2x2_mean():
    rng = np.random.default_rng(0)
    x = rng.random((1, 64, 64, 2), dtype=np.float32)
    y = O.resize_bilinear(_nhwc(x), (8, 8)).numpy()
    want = np.zeros((1, 8, 8, 2), np.float32)
    for o in range(8):
        for p in range(8):
            want[0, o, p] = x[0, 8 * o + 3:8 * o + 5, 8 * p + 3:8 * p + 5].mean(axis=(0, 1))
    np.testing.assert_allclose(y, want, atol=1e-6)
    np.testing.assert_allclose(L.resize_bilinear(x, 8, 8), want, atol=1e-6)


def test_a6_grayscale_weights():
    x = _nhwc(np.array([[[[1.0, 0, 0], [0, 1.0, 0]], [[0, 0, 1.0], [1.0, 1.0, 1.0]]]]))
    y = O.rgb_to_grayscale(x).numpy()[0, :, :, 0]
    np.testing.assert_allclose(y, [[0.2989, 0.5870], [0.1140, 0.9999]], rtol=1e-6)
    assert O.rgb_to_grayscale(x).shape == (1, 2, 2, 1)


def test_a7_a8_nonlocal_token_order_and_no_scale():
    # 1x2 map, C=2 -> C/2=1.  theta=phi=g=identity-on-channel-0; logits f[t,s] = x_t0 * x_s0 (NO 1/sqrt(d))
    w = {}
    for n in ("g", "phi", "theta"):
        k = np.zeros((1, 1, 2, 1), np.float32)
        k[0, 0, 0, 0] = 1
        w["nl/" + n + "/kernel"] = k
        w["nl/" + n + "/bias"] = np.zeros(1, np.float32)
    kw = np.zeros((1, 1, 1, 2), np.float32)
    kw[0, 0, 0] = [1, 2]
    w["nl/w/kernel"] = kw
    w["nl/w/bias"] = np.zeros(2, np.float32)
    w["nl/bnorm/gamma"] = np.ones(2, np.float32)
    w["nl/bnorm/beta"] = np.zeros(2, np.float32)
    w["nl/bnorm/moving_mean"] = np.zeros(2, np.float32)
    w["nl/bnorm/moving_variance"] = np.full(2, 1 - 1e-3, np.float32)
    x = np.array([[[[1.0, 5.0], [3.0, 7.0]]]], np.float32)      # tokens: t0=(1,5), t1=(3,7)
    z = O.GeneratorOracle(w).non_local(_nhwc(x), "nl/").numpy()
    out = []
    for t, xt in enumerate([1.0, 3.0]):
        f = np.array([xt * 1.0, xt * 3.0])
        p = np.exp(f - f.max())
        p /= p.sum()
        y = p[0] * 1.0 + p[1] * 3.0
        out.append([x[0, 0, t, 0] + y * 1, x[0, 0, t, 1] + y * 2])
    np.testing.assert_allclose(z[0, 0], out, rtol=1e-5)
    np.testing.assert_allclose(L.non_local(w, "nl/", x)[0, 0], out, rtol=1e-5)


def test_a9_channel_zero_pad_at_tail_and_a10_strict_threshold():
    # ResBottleneck pads the NARROWER of x / y at the channel tail: checked through shapes+values of the
    # full generator in test_oracle_two_forms; here the strict '>' of the bmask threshold
    d = torch.tensor([0.1, np.nextafter(np.float32(0.1), np.float32(1.0)), 0.09])
    assert ((d > O.BMASK_THRESHOLD).float().numpy() == [0, 1, 0]).all()
