"""-m gpu parity tests: the HIP path, called through the C ABI (ctypes -> libbsr_hip.so), against the CPU
oracle on the same seeded inputs.  Tolerance: 1e-3 absolute per pixel in fp32 (BASELINE.json north_star);
measured agreement is ~1e-5."""

import pytest
import torch

from blindshadowremoval_amd.weights import init_weights

pytestmark = pytest.mark.gpu

# Tolerance of the f16 mode (BASELINE configs[3]) against the fp32 oracle, and its measured margin PER TEST (round 5: the review asked for
# the measured value beside every use; profiles/r5_f16_margins.txt holds the lines these tests wrote on the MI355X):
#   test_f16_mfma_mode_tracks_the_fp32_oracle      B = 2               max abs err 1.264e-3  (63 % of F16_TOL)
#   test_config3_rank_shape_f16_batch32            B = 32              max abs err 1.373e-3  (69 %)
#   test_config3_full_size_f16_batch256            8 rows of B = 256   max abs err 1.309e-3  (65 %)
#   test_tsm_f16_mode_tracks_the_oracle            TSM, B = 4          max abs err 1.542e-3  (77 %)
# A margin below 20 % of the tolerance fails the test: a kernel change that eats the headroom is noticed before it eats the tolerance.
F16_TOL = 2e-3
F16_MIN_MARGIN = 0.20


def _note_f16_margin(test: str, err: float) -> None:
    import os
    line = "%s max_abs_err %.3e tol %.1e used %.0f%%" % (test, err, F16_TOL, 100 * err / F16_TOL)
    print(line)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "f16_margins.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert err <= (1.0 - F16_MIN_MARGIN) * F16_TOL, "f16 mode: %s — less than %.0f %% of the tolerance is left" % (line, 100 * F16_MIN_MARGIN)

PROBES = ("x1", "x2", "x3", "x0", "res0", "res1", "res2", "up1", "up2", "y", "res3", "res4", "res5", "f")


@pytest.fixture(scope="module", params=["f32", "f32x3"])
def gen_w(request):
    """Every test that takes this fixture runs twice: on the fp32 matrix-core path (the measured one) and on the split-precision
    "f32x3" path (3x3-conv layers on the fp16 matrix cores with hi/lo operand planes, csrc/igemm_h16.h) — the SAME tolerances
    (TOL 1e-3, FLIP_TOL 2e-5 of parity_util.py) apply to both."""
    from blindshadowremoval_amd import Generator
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    w = init_weights(1)
    gen = Generator(dtype=request.param).load_weights(w)
    yield gen, w
    gen.close()


def test_library_is_the_in_tree_hip_extension():
    from blindshadowremoval_amd import _lib
    from blindshadowremoval_amd.build import LIB_PATH
    lib = _lib.load()
    assert lib._name == LIB_PATH and lib.bsr_abi_version() == _lib.ABI_VERSION == 8


@pytest.mark.parametrize("seed,B", [(0, 2), (7, 3)])
def test_forward_matches_oracle_with_probes(gen_w, seed, B):
    from parity_util import run_and_compare
    gen, w = gen_w
    torch.manual_seed(seed)
    inp, uv = torch.rand(B, 256, 256, 3), torch.rand(B, 256, 256, 3)
    uv[:, :, :30] = 0                      # real uv maps are ~60 % zeros outside the landmark hull
    out, ref, errs, nflip = run_and_compare(gen, w, inp, uv, want_probes=PROBES)
    assert max(errs.values()) <= 1e-3
    assert out[0].shape == (B, 256, 256, 1) and out[1].shape == (B, 256, 256, 3)
    assert out[2].shape == (B, 256, 256, 3) and out[3].shape == (B, 256, 256, 1)


def test_other_weights_and_degenerate_masks(gen_w):
    """bmask all-zero (no bias shift) and all-one (large shift): both branches of x*(1-bmask)."""
    from blindshadowremoval_amd import Generator
    from parity_util import run_and_compare
    torch.manual_seed(11)
    inp, uv = torch.rand(1, 256, 256, 3), torch.rand(1, 256, 256, 3)
    for shift, want_mean in ((-1.0, 0.0), (2.0, 1.0)):
        w = init_weights(4, con_bias_shift=shift)
        gen = Generator(dtype=gen_w[0].dtype).load_weights(w)
        run_and_compare(gen, w, inp, uv)
        assert float(gen.probe("bmask").mean()) == want_mean


def test_rows_are_independent_and_deterministic(gen_w):
    """no cross-sample op in GSC inference, so an image's outputs do not depend on what
    else is in the batch — bit for bit — and repeated runs are bit-identical."""
    gen, _ = gen_w
    torch.manual_seed(5)
    inp, uv = torch.rand(4, 256, 256, 3).cuda(), torch.rand(4, 256, 256, 3).cuda()
    full = [t.clone() for t in gen(inp, uv)]
    again = gen(inp, uv)
    for a, b in zip(full, again):
        assert torch.equal(a, b)
    solo = gen(inp[2:3].contiguous(), uv[2:3].contiguous())
    for a, b in zip(full, solo):
        assert torch.equal(a[2:3], b)
    # the reference feeds 10 copies/siblings and keeps row 0 (train_test_GSC.py:866-871, utils.py:231)
    ten = gen(inp[:1].repeat(10, 1, 1, 1), uv[:1].repeat(10, 1, 1, 1))
    for a, b in zip(full, ten):
        assert torch.equal(a[0], b[0]) and torch.equal(b[0], b[9])


def test_small_batches_equal_the_rows_of_the_full_batch(gen_w):
    """Round 4: below B = 16 the forward picks smaller workgroup shapes for the 1/8-resolution trunk (2x32-pixel conv tiles, 64- / 32-query
    attention blocks, finer N ranges in the bottleneck GEMMs) so that the grid still covers the chip.  Every shape accumulates each output
    element in the same order: B = 1, 2, 8, 10 (the reference's literal element, train_test_GSC.py:866-871) and 16 (BASELINE configs[2])
    reproduce the rows of the B = 32 forward bit for bit."""
    gen, _ = gen_w
    torch.manual_seed(15)
    inp, uv = torch.rand(32, 256, 256, 3).cuda(), torch.rand(32, 256, 256, 3).cuda()
    full = [t.clone() for t in gen(inp, uv)]
    for B, lo in ((1, 31), (2, 5), (8, 16), (10, 0), (16, 16)):
        part = gen(inp[lo:lo + B].contiguous(), uv[lo:lo + B].contiguous())
        for a, b, name in zip(full, part, ("gs", "con_rgb", "mask22", "dif")):
            assert torch.equal(a[lo:lo + B], b), (B, name)


def test_attention_workgroup_shapes_are_bit_identical():
    """nonlocal_attention_kernel<QW>: 128 / 64 / 32 queries per workgroup (4 / 2 / 1 query waves x two key streams) — a wave's work is the
    same 32 queries x one key stream in every shape, so all of them, and the automatic choice, give the same bits."""
    from blindshadowremoval_amd import _lib
    lib = _lib.load()
    torch.manual_seed(21)
    B, T, D = 3, 1024, 128
    x = (torch.randn(B, T, 3 * D) * 0.5).cuda()
    outs = []
    for qw in (4, 2, 1, 0):
        y = torch.empty(B, T, D, device="cuda")
        _lib.check(lib.bsr_debug_attention_qw(x.data_ptr(), y.data_ptr(), B, T, qw, None), "bsr_debug_attention_qw")
        torch.cuda.synchronize()
        outs.append(y)
    for y in outs[1:]:
        assert torch.equal(outs[0], y)
    q, k, v = (t.double().cpu() for t in x.split(D, dim=2))
    ref = torch.softmax(q @ k.transpose(1, 2), -1) @ v
    assert float((outs[0].cpu().double() - ref).abs().max()) < 2e-5
    assert lib.bsr_debug_attention_qw(x.data_ptr(), y.data_ptr(), B, T, 3, None) == 1


def test_full_batch_properties(gen_w):
    """BASELINE config 2 size (B=32): size-independent properties of the four outputs."""
    gen, _ = gen_w
    torch.manual_seed(6)
    inp, uv = torch.rand(32, 256, 256, 3).cuda(), torch.rand(32, 256, 256, 3).cuda()
    gs, con_rgb, mask22, dif = gen(inp, uv)
    assert torch.isfinite(gs).all() and torch.isfinite(con_rgb).all()
    assert float(mask22[..., 1].abs().max()) == 0.0                      # mask*0 (model.py:252)
    assert float((mask22[..., 0] * mask22[..., 2]).abs().max()) == 0.0   # relu(m) * relu(-m) == 0
    assert float(mask22.min()) >= 0.0 and float(mask22.max()) < 1.0      # tanh range
    gw = torch.tensor([0.2989, 0.5870, 0.1140], device="cuda")
    recomputed = (con_rgb * gw).sum(-1, keepdim=True) - (inp * gw).sum(-1, keepdim=True)   # model.py:288
    assert float((dif - recomputed).abs().max()) < 1e-5
    bm = gen.probe("bmask")
    xh, r2 = gen.probe("xh"), gen.probe("res2")
    assert set(bm.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(xh[..., :257], r2 * (1 - bm)) and torch.equal(xh[..., 257:258], bm)   # model.py:258-259
    # all 32 rows against the oracle (F7 protocol: d32 to 1e-3, mask flips only on the threshold, outputs given the same mask)
    from oracle.gsc_oracle import GeneratorOracle
    from parity_util import FLIP_TOL
    oracle, pr = GeneratorOracle(gen_w[1]), {}
    oracle(inp.cpu(), uv.cpu(), probes=pr)
    assert float((gen.probe("d32").cpu() - pr["d32"]).abs().max()) <= 1e-3
    flips = bm.cpu() != pr["bmask"]
    assert not flips.any() or float((pr["d32"][flips] - 0.1).abs().max()) < FLIP_TOL
    ref = oracle(inp.cpu(), uv.cpu(), bmask_override=bm.cpu())
    for a, b in zip((gs, con_rgb, mask22, dif), ref):
        assert float((a.cpu() - b).abs().max()) <= 1e-3


@pytest.mark.parametrize("dtype_code", [0, 2])
def test_attention_kernel_forced_rescale(dtype_code):
    """Online-softmax rescale branch (cdna guide rule 26): spike late keys so the running max jumps in the
    last tiles; compare with an fp64 softmax on the full tensor.  dtype_code 0 = the fp32 matrix-core kernel (attention.h),
    2 = the split-precision fp16 matrix-core kernel of the f32x3 / f16 modes (attention_h16.h, fed through bsr_debug_split_qkv) — same tolerance."""
    from blindshadowremoval_amd import _lib
    lib = _lib.load()
    torch.manual_seed(2)
    B, T, D = 2, 1024, 128
    qkv = torch.randn(B, T, 3 * D) * 0.5
    qkv[:, 1000, D:2 * D] = qkv[:, 17, :D] * 6.0        # key 1000 aligned with query 17: large late logit
    qkv[:, 500:520, D:2 * D] *= 8.0                      # a mid-sequence burst of big keys
    qkv[0, 3, :D] = 0.0                                  # a query with all-zero logits (uniform softmax)
    x = qkv.cuda()
    y = torch.empty(B, T, D, device="cuda")
    _lib.check(lib.bsr_debug_attention_dtype(x.data_ptr(), y.data_ptr(), B, T, dtype_code, None), "bsr_debug_attention")
    torch.cuda.synchronize()
    q, k, v = (t.double() for t in qkv.split(D, dim=2))
    ref = torch.softmax(q @ k.transpose(1, 2), -1) @ v
    assert float((q @ k.transpose(1, 2)).max()) > 50.0          # the test really exercises large logits
    assert float((y.cpu().double() - ref).abs().max()) < 2e-5
    print("attention dtype", dtype_code, "max abs err vs fp64", float((y.cpu().double() - ref).abs().max()))
    rc = lib.bsr_debug_attention_dtype(x.data_ptr(), y.data_ptr(), B, 1000, dtype_code, None)
    assert rc == 1 and b"multiple of 128" in lib.bsr_last_error()


def test_argument_errors(gen_w):
    from blindshadowremoval_amd import Generator
    gen, w = gen_w
    ok = torch.rand(1, 256, 256, 3)
    with pytest.raises(ValueError):
        gen(torch.rand(1, 256, 256, 4), ok)
    with pytest.raises(ValueError):
        gen(torch.rand(1, 250, 256, 3), torch.rand(1, 250, 256, 3))
    with pytest.raises(ValueError):
        gen(ok, torch.rand(2, 256, 256, 3))
    with pytest.raises(TypeError):
        gen(ok.double(), ok)
    with pytest.raises(NotImplementedError):
        gen(ok, ok, training=True)
    with pytest.raises(RuntimeError, match="no weights"):
        Generator()(ok, ok)
    with pytest.raises(RuntimeError, match="unknown probe"):
        gen(ok, ok); gen.probe("nope")
    # C ABI level: bad H/W are rejected with BSR_ERR_ARG
    t = ok.cuda()
    o = torch.empty(1, 256, 256, 3, device="cuda")
    rc = gen._lib.bsr_forward(gen._handle, t.data_ptr(), t.data_ptr(), 1, 100, 256, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr(), None)
    assert rc == 1
    # a blob packed for one dtype is refused by bsr_create of another
    import ctypes
    from blindshadowremoval_amd.pack import pack_generator
    blob = pack_generator(w, "f32")
    hnd = ctypes.c_void_p()
    buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
    assert gen._lib.bsr_create(ctypes.byref(hnd), 0, ctypes.cast(buf, ctypes.c_void_p), len(blob), 2) == 2
    assert b"another dtype" in gen._lib.bsr_last_error()


def test_wider_input_512(gen_w):
    """The kernels are size-generic (BASELINE config 5 runs 512x512 frames): one 256x512 image vs oracle."""
    from parity_util import run_and_compare
    gen, w = gen_w
    torch.manual_seed(8)
    inp, uv = torch.rand(1, 256, 512, 3), torch.rand(1, 256, 512, 3)
    run_and_compare(gen, w, inp, uv)


@pytest.mark.parametrize("frame,share,dtype", [(2, True, "f32"), (4, True, "f32"), (2, False, "f32"), (2, True, "f32x3"), (4, True, "f32x3")])
def test_tsm_variant_matches_oracle(frame, share, dtype):
    """BASELINE config 5 path: TSM generator = GSC net + ShareLayer (offset warp -> group max|mean -> inverse warp),
    /root/reference/model_with_TSM.py:199-325, against the TSM oracle whose warp is pinned to the reference's scipy form."""
    from blindshadowremoval_amd import GeneratorTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    w = init_weights(1, variant="tsm")
    gen = GeneratorTSM(dtype=dtype).load_weights(w)
    torch.manual_seed(13 + frame)
    B = 4
    inp, uv = torch.rand(B, 256, 256, 3), torch.rand(B, 256, 256, 3)
    # smooth offset fields of a few cells amplitude (real fields are |.| <~ 0.11 image fractions), some leaving the map
    reg = torch.nn.functional.interpolate((torch.rand(B, 6, 9, 9) - 0.5) * 0.3, size=(256, 256), mode="bicubic", align_corners=True).permute(0, 2, 3, 1).contiguous()
    reg[..., 2] = 0
    reg[..., 5] = 0
    out = [t.cpu() for t in gen(inp.cuda(), uv.cuda(), reg.cuda(), frame, share)]
    bmask = gen.probe("bmask").cpu()
    oracle = GeneratorTSMOracle(w)
    pr = {}
    ref = oracle(inp, uv, reg, frame, share, probes=pr)
    assert float((gen.probe("d32").cpu() - pr["d32"]).abs().max()) <= 1e-3
    flips = bmask != pr["bmask"]
    if int(flips.sum()):
        assert float((pr["d32"][flips] - 0.1).abs().max()) < 2e-5
        pr = {}
        ref = oracle(inp, uv, reg, frame, share, probes=pr, bmask_override=bmask)
    assert float((gen.probe("x0").cpu() - pr["x0"]).abs().max()) <= 1e-4         # cat[x, x_share, uv]: first ShareLayer
    assert float((gen.probe("res2").cpu() - pr["res2"]).abs().max()) <= 1e-3    # 291-wide blocks
    assert float((gen.probe("res5").cpu() - pr["res5"]).abs().max()) <= 1e-3    # 877-wide blocks
    for a, b, name in zip(out, ref, ("gs", "con_rgb", "mask22", "dif")):
        assert float((a - b).abs().max()) <= 1e-3, name
    # API misuse: GSC call on a TSM handle, bad frame
    with pytest.raises(RuntimeError, match="TSM weights"):
        gen.__class__.__mro__[1].__call__(gen, inp[:1], uv[:1])
    with pytest.raises(ValueError):
        gen(inp, uv, reg, 3, True)


def test_edge_inputs(gen_w):
    """Empty batch is rejected; non-contiguous / channel-sliced inputs give the same bits as contiguous ones; a batch
    larger than the bench's (B = 48) reproduces the rows of smaller batches bit for bit (workspace regrowth included)."""
    gen, _ = gen_w
    with pytest.raises((ValueError, RuntimeError)):
        gen(torch.rand(0, 256, 256, 3), torch.rand(0, 256, 256, 3))
    torch.manual_seed(17)
    packed = torch.rand(3, 256, 256, 16).cuda()                      # the loader's 16-channel layout: img = [..., 0:3], uv = [..., 6:9]
    img_v, uv_v = packed[..., 0:3], packed[..., 6:9]                 # strided views, not contiguous
    assert not img_v.is_contiguous()
    a = gen(img_v, uv_v)
    b = gen(img_v.contiguous(), uv_v.contiguous())
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    big_in, big_uv = torch.rand(48, 256, 256, 3).cuda(), torch.rand(48, 256, 256, 3).cuda()
    big = [t.clone() for t in gen(big_in, big_uv)]
    small = gen(big_in[40:44].contiguous(), big_uv[40:44].contiguous())
    for x, y in zip(big, small):
        assert torch.equal(x[40:44], y)


def test_f16_mfma_mode_tracks_the_fp32_oracle():
    """BASELINE configs[3] (fp16 MFMA conv path, opt-in BSR_DTYPE_F16): operands of the 3x3-conv layers are rounded to fp16
    (11-bit significand, half-width LDS tiles, v_mfma_f32_32x32x16_f16) and accumulated in fp32, so parity is NOT the 1e-3 fp32
    bar: the tolerance here is F16_TOL on every output and on d32, and bmask cells may flip only where d32 is within F16_TOL of
    the 0.1 threshold (same protocol as fp32, wider band).  The fp32 mode stays the measured/default path; the fp32-accurate
    fast path is dtype "f32x3" (run by every gen_w test above under the unchanged fp32 tolerances)."""
    from blindshadowremoval_amd import Generator
    from parity_util import run_and_compare
    weights = init_weights(1)
    gen = Generator(dtype="f16").load_weights(weights)
    g = torch.Generator().manual_seed(5)
    inp, uv = torch.rand(2, 256, 256, 3, generator=g), torch.rand(2, 256, 256, 3, generator=g)
    out, ref, errs, nflip = run_and_compare(gen, weights, inp, uv, tol=F16_TOL, flip_tol=F16_TOL)
    print("f16 mode: max abs err", errs, "bmask flips", nflip)
    _note_f16_margin("test_f16_mfma_mode_tracks_the_fp32_oracle", max(errs.values()))
    assert max(errs.values()) > 1e-6          # it really is a different arithmetic (guards against silently running fp32)
    gen.close()


def test_config3_rank_shape_f16_batch32():
    """BASELINE configs[3] = 256 images over 8 GPUs with the fp16 MFMA conv path: the PER-RANK shape (B = 32, dtype f16) on one
    GPU against the fp32 oracle, all 32 rows, same tolerance as the B = 2 test above."""
    from blindshadowremoval_amd import Generator
    from parity_util import run_and_compare
    weights = init_weights(1)
    gen = Generator(dtype="f16").load_weights(weights)
    g = torch.Generator().manual_seed(21)
    inp, uv = torch.rand(32, 256, 256, 3, generator=g), torch.rand(32, 256, 256, 3, generator=g)
    out, ref, errs, nflip = run_and_compare(gen, weights, inp, uv, tol=F16_TOL, flip_tol=F16_TOL)
    print("f16 B=32: max abs err", errs, "bmask flips", nflip)
    _note_f16_margin("test_config3_rank_shape_f16_batch32", max(errs.values()))
    assert out[1].shape == (32, 256, 256, 3)
    gen.close()


@pytest.mark.parametrize("dtype", ["f32", "f32x3"])
def test_config4_rank_shape_tsm_512_batch8(dtype):
    """BASELINE configs[4] = 64 frames of 512x512 over 8 GPUs through the TSM generator: the PER-RANK shape (B = 8, 512x512,
    frame = 2) on one GPU against the TSM oracle (4096-token attention, 64x64 ShareLayer warp) — fp32 and split-precision paths."""
    from blindshadowremoval_amd import GeneratorTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    from parity_util import FLIP_TOL, TOL
    w = init_weights(1, variant="tsm")
    gen = GeneratorTSM(dtype=dtype).load_weights(w)
    g = torch.Generator().manual_seed(33)
    B, S = 8, 512
    inp, uv = torch.rand(B, S, S, 3, generator=g), torch.rand(B, S, S, 3, generator=g)
    reg = torch.nn.functional.interpolate((torch.rand(B, 6, 9, 9, generator=g) - 0.5) * 0.2, size=(S, S), mode="bicubic",
                                          align_corners=True).permute(0, 2, 3, 1).contiguous()
    reg[..., 2] = 0
    reg[..., 5] = 0
    out = [t.cpu() for t in gen(inp.cuda(), uv.cuda(), reg.cuda(), 2, True)]
    bmask, d32 = gen.probe("bmask").cpu(), gen.probe("d32").cpu()
    oracle, pr = GeneratorTSMOracle(w), {}
    ref = oracle(inp, uv, reg, 2, True, probes=pr)
    assert float((d32 - pr["d32"]).abs().max()) <= TOL
    flips = bmask != pr["bmask"]
    if int(flips.sum()):
        assert float((pr["d32"][flips] - 0.1).abs().max()) < FLIP_TOL
        ref = oracle(inp, uv, reg, 2, True, bmask_override=bmask)
    for a, b, name in zip(out, ref, ("gs", "con_rgb", "mask22", "dif")):
        assert a.shape == b.shape == (B, S, S, a.shape[3])
        assert float((a - b).abs().max()) <= TOL, name
    gen.close()


def test_out_buffers_are_validated_and_device_is_restored(gen_w):
    """Caller-supplied `out=` buffers reach the library as raw pointers: wrong shape / dtype / layout / device must be refused
    before the call; the C ABI restores the caller's current device."""
    gen, _ = gen_w
    t = torch.rand(2, 256, 256, 3).cuda()
    good = tuple(torch.empty((2, 256, 256, c), device="cuda") for c in (1, 3, 3, 1))
    a = [x.clone() for x in gen(t, t, out=good)]
    b = gen(t, t)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        gen(t, t, out=good[:3])
    with pytest.raises(ValueError):
        gen(t, t, out=(good[0], good[1], good[2], torch.empty((1, 256, 256, 1), device="cuda")))
    with pytest.raises(TypeError):
        gen(t, t, out=(good[0], good[1].double(), good[2], good[3]))
    with pytest.raises(ValueError):
        gen(t, t, out=(good[0], torch.empty((2, 256, 256, 6), device="cuda")[..., ::2], good[2], good[3]))
    with pytest.raises(ValueError):
        gen(t, t, out=(good[0].cpu(), good[1], good[2], good[3]))
    assert torch.cuda.current_device() == 0


def test_per_launch_timing_names_every_layer(gen_w):
    """bsr_timing_entry: one entry per launch, named by the reference layer(s) it computes; bench.py's roofline table keys on them."""
    import importlib.util
    import os
    gen, _ = gen_w
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = torch.rand(1, 256, 256, 3).cuda()
    gen.set_timing(True)
    gen(t, t)
    torch.cuda.synchronize()
    entries = gen.get_launch_timing()
    gen.set_timing(False)
    names = [n for n, _, _ in entries]
    covered1 = []
    for n in names:                      # 16-bit modes: attention + w are one launch at every batch (round 5), fp32 only at full batches
        covered1 += list(bench.FUSED_LAUNCHES.get(n, (n,) if n in bench.LAYER_MMAC else ()))
    assert sorted(covered1) == sorted(bench.LAYER_MMAC), sorted(set(bench.LAYER_MMAC) ^ set(covered1))
    assert ("res0.attw" in names) == (gen.dtype != "f32")
    assert all(ms > 0 for _, ms, _ in entries)
    assert abs(sum(bench.LAYER_MMAC.values()) - 9052.06) < 1.0          # SURVEY Appendix C total
    grouped = [n for layers in bench.KERNEL_GROUPS.values() for n in layers]
    assert sorted(grouped) == sorted(list(bench.LAYER_MMAC) + list(bench.FUSED_LAUNCHES))
    # a full batch: some layers run as the tail of another launch (round 4) — every reference layer is still covered exactly once,
    # directly or through the fused launch that names it, and bench.py can price every launch
    t32 = torch.rand(32, 256, 256, 3).cuda()
    gen.set_timing(True)
    gen(t32, t32)
    torch.cuda.synchronize()
    names32 = [n for n, _, _ in gen.get_launch_timing()]
    gen.set_timing(False)
    covered = []
    for n in names32:
        covered += list(bench.FUSED_LAUNCHES.get(n, (n,) if n in bench.LAYER_MMAC else ()))
    assert sorted(covered) == sorted(bench.LAYER_MMAC), sorted(set(bench.LAYER_MMAC) ^ set(covered))
    assert "res0.attw" in names32


def test_handle_lifecycle_returns_its_memory():
    """bsr_create / forward / bsr_destroy in a loop: the handle's weights and workspace go back to the device (the library owns them
    with hipMalloc, outside torch's caching allocator), so free memory after 8 cycles is what it was after the first."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    inp, uv = torch.rand(4, 256, 256, 3).cuda(), torch.rand(4, 256, 256, 3).cuda()
    free = []
    for i in range(8):
        gen = Generator(dtype="f32x3" if i & 1 else "f32").load_weights(w)
        gen(inp, uv)
        torch.cuda.synchronize()
        gen.close()
        free.append(torch.cuda.mem_get_info()[0])
    assert free[-1] >= free[0] - (16 << 20), free
    with pytest.raises(RuntimeError):
        gen(inp, uv)                                  # a closed generator refuses work instead of touching freed memory


def test_two_handles_on_two_streams_match_serial_runs():
    """One handle per stream (include/bsr_hip.h): two generators driven concurrently from two HIP streams give the bits of the
    same calls made one after the other."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    torch.manual_seed(23)
    ins = [(torch.rand(6, 256, 256, 3).cuda(), torch.rand(6, 256, 256, 3).cuda()) for _ in range(2)]
    gens = [Generator().load_weights(w) for _ in range(2)]
    serial = [[t.clone() for t in g(*x)] for g, x in zip(gens, ins)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [None, None]
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i] = [t.clone() for t in gens[i](*ins[i])]
    torch.cuda.synchronize()
    for a, b in zip(serial, outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    for g in gens:
        g.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] / configs[4] at their STATED sizes on one GPU (the 8-way shard of these batches is bench.py's RCCL path; rows
# (GSC) / frame groups (TSM) are independent, so the single-GPU result is what the 8 ranks' all-gather must re-assemble).

def _slices_equal(big, small, lo, hi):
    for a, b in zip(big, small):
        if not torch.equal(a[lo:hi], b):
            return False
    return True


def test_config3_full_size_f16_batch256():
    """configs[3] "Batch=256 synthetic 256x256x3, fp16 MFMA conv path": ONE B = 256 forward (21 GB of workspace) must equal, bit for
    bit, the eight B = 32 per-rank forwards of the same rows; 8 sampled rows against the fp32 oracle at the f16 tolerance; the
    size-independent output properties on all 256."""
    from blindshadowremoval_amd import Generator
    from oracle.gsc_oracle import GeneratorOracle
    weights = init_weights(1)
    gen = Generator(dtype="f16").load_weights(weights)
    g = torch.Generator().manual_seed(41)
    B = 256
    inp, uv = torch.rand(B, 256, 256, 3, generator=g), torch.rand(B, 256, 256, 3, generator=g)
    uv[:, :, :40] = 0
    inp_d, uv_d = inp.cuda(), uv.cuda()
    big = [t.clone() for t in gen(inp_d, uv_d)]
    bm_big, d32_big = gen.probe("bmask").clone(), gen.probe("d32").clone()
    for r in range(8):                                       # the per-rank shards of the 8-GPU run
        lo, hi = 32 * r, 32 * r + 32
        small = gen(inp_d[lo:hi].contiguous(), uv_d[lo:hi].contiguous())
        assert _slices_equal(big, small, lo, hi), "rank %d shard differs from the B = 256 forward" % r
    gs, con_rgb, mask22, dif = big
    assert all(bool(torch.isfinite(t).all()) for t in big)
    assert float(mask22[..., 1].abs().max()) == 0.0 and float((mask22[..., 0] * mask22[..., 2]).abs().max()) == 0.0
    gw = torch.tensor([0.2989, 0.5870, 0.1140], device="cuda")
    assert float((dif - ((con_rgb * gw).sum(-1, keepdim=True) - (inp_d * gw).sum(-1, keepdim=True))).abs().max()) < 1e-5
    rows = [0, 31, 32, 100, 127, 128, 200, 255]
    oracle, pr = GeneratorOracle(weights), {}
    oracle(inp[rows], uv[rows], probes=pr)
    assert float((d32_big[rows].cpu() - pr["d32"]).abs().max()) <= F16_TOL
    flips = bm_big[rows].cpu() != pr["bmask"]
    assert not flips.any() or float((pr["d32"][flips] - 0.1).abs().max()) < F16_TOL
    ref = oracle(inp[rows], uv[rows], bmask_override=bm_big[rows].cpu())
    worst = 0.0
    for a, b, name in zip(big, ref, ("gs", "con_rgb", "mask22", "dif")):
        err = float((a[rows].cpu() - b).abs().max())
        print("configs[3] B=256 f16 %s max abs err %.3e" % (name, err))
        assert err <= F16_TOL, name
        worst = max(worst, err)
    _note_f16_margin("test_config3_full_size_f16_batch256", worst)
    gen.close()


@pytest.mark.parametrize("dtype", ["f32"])
def test_config4_full_size_tsm_512_batch64(dtype):
    """configs[4] "512x512 frames batch=64" through the TSM generator, frame = 2 (train_with_TSM.py:676): ONE B = 64 forward must
    equal bit for bit the eight B = 8 per-rank forwards (shards on frame-group boundaries); two frame groups (4 images) from both
    ends of the batch against the TSM oracle at 1e-3."""
    from blindshadowremoval_amd import GeneratorTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    from parity_util import FLIP_TOL, TOL
    w = init_weights(1, variant="tsm")
    gen = GeneratorTSM(dtype=dtype).load_weights(w)
    g = torch.Generator().manual_seed(43)
    B, S = 64, 512
    inp, uv = torch.rand(B, S, S, 3, generator=g), torch.rand(B, S, S, 3, generator=g)
    reg = torch.nn.functional.interpolate((torch.rand(B, 6, 9, 9, generator=g) - 0.5) * 0.2, size=(S, S), mode="bicubic",
                                          align_corners=True).permute(0, 2, 3, 1).contiguous()
    reg[..., 2] = 0
    reg[..., 5] = 0
    inp_d, uv_d, reg_d = inp.cuda(), uv.cuda(), reg.cuda()
    big = [t.clone() for t in gen(inp_d, uv_d, reg_d, 2, True)]
    bm_big, d32_big = gen.probe("bmask").clone(), gen.probe("d32").clone()
    for r in range(8):
        lo, hi = 8 * r, 8 * r + 8
        small = gen(inp_d[lo:hi].contiguous(), uv_d[lo:hi].contiguous(), reg_d[lo:hi].contiguous(), 2, True)
        assert _slices_equal(big, small, lo, hi), "rank %d shard differs from the B = 64 forward" % r
    assert all(bool(torch.isfinite(t).all()) for t in big)
    rows = [0, 1, 62, 63]                                   # frame groups 0 and 31
    oracle, pr = GeneratorTSMOracle(w), {}
    oracle(inp[rows], uv[rows], reg[rows], 2, True, probes=pr)
    assert float((d32_big[rows].cpu() - pr["d32"]).abs().max()) <= TOL
    bm = bm_big[rows].cpu()
    flips = bm != pr["bmask"]
    assert not flips.any() or float((pr["d32"][flips] - 0.1).abs().max()) < FLIP_TOL
    ref = oracle(inp[rows], uv[rows], reg[rows], 2, True, bmask_override=bm)
    for a, b, name in zip(big, ref, ("gs", "con_rgb", "mask22", "dif")):
        err = float((a[rows].cpu() - b).abs().max())
        print("configs[4] B=64 512x512 TSM %s %s max abs err %.3e" % (dtype, name, err))
        assert err <= TOL, name
    gen.close()


@pytest.mark.parametrize("H,W", [(288, 256), (320, 512)])
def test_heights_that_are_not_powers_of_two(gen_w, H, W):
    """The C ABI accepts H % 32 == 0, W % 256 == 0: 288x256 (1152 tokens) and 320x512 (2560 tokens) exercise the attention
    kernel's XCD remap with a block count that is not a power of two and every conv grid with a ragged tile count."""
    from parity_util import run_and_compare
    gen, w = gen_w
    torch.manual_seed(H)
    inp, uv = torch.rand(2, H, W, 3), torch.rand(2, H, W, 3)
    out, ref, errs, nflip = run_and_compare(gen, w, inp, uv, want_probes=("x0", "res2", "res5", "y", "f"))
    assert out[1].shape == (2, H, W, 3)


def test_f16_handle_is_reusable_across_shapes():
    """One f16 handle across B / H / W changes (the workspace keeps fp16 data in half of each slot and a shape change clears only the
    channel-pad lanes): every forward equals, bit for bit, the same call on a fresh handle."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    gen = Generator(dtype="f16").load_weights(w)
    g = torch.Generator().manual_seed(51)
    for (B, H, W) in ((8, 256, 256), (2, 256, 256), (1, 256, 512), (3, 288, 256), (8, 256, 256)):
        inp, uv = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        a = [t.clone() for t in gen(inp, uv)]
        fresh = Generator(dtype="f16").load_weights(w)
        b = fresh(inp, uv)
        for x, y in zip(a, b):
            assert torch.equal(x, y), (B, H, W)
        fresh.close()
    gen.close()


def test_tsm_f16_mode_tracks_the_oracle():
    """GeneratorTSM(dtype="f16") (the kTSM16 channel plan with fp16 intermediate tensors) against the fp32 TSM oracle at the f16
    tolerance — the combination FSRNetTSM(dtype="f16") would run."""
    from blindshadowremoval_amd import GeneratorTSM
    from oracle.gsc_oracle import GeneratorTSMOracle
    w = init_weights(1, variant="tsm")
    gen = GeneratorTSM(dtype="f16").load_weights(w)
    g = torch.Generator().manual_seed(53)
    B = 4
    inp, uv = torch.rand(B, 256, 256, 3, generator=g), torch.rand(B, 256, 256, 3, generator=g)
    reg = torch.nn.functional.interpolate((torch.rand(B, 6, 9, 9, generator=g) - 0.5) * 0.3, size=(256, 256), mode="bicubic",
                                          align_corners=True).permute(0, 2, 3, 1).contiguous()
    reg[..., 2] = 0
    reg[..., 5] = 0
    out = [t.cpu() for t in gen(inp.cuda(), uv.cuda(), reg.cuda(), 2, True)]
    bmask, d32 = gen.probe("bmask").cpu(), gen.probe("d32").cpu()
    oracle, pr = GeneratorTSMOracle(w), {}
    oracle(inp, uv, reg, 2, True, probes=pr)
    assert float((d32 - pr["d32"]).abs().max()) <= F16_TOL
    flips = bmask != pr["bmask"]
    assert not flips.any() or float((pr["d32"][flips] - 0.1).abs().max()) < F16_TOL
    ref = oracle(inp, uv, reg, 2, True, bmask_override=bmask)
    worst = 0.0
    for a, b, name in zip(out, ref, ("gs", "con_rgb", "mask22", "dif")):
        err = float((a - b).abs().max())
        print("TSM f16 %s max abs err %.3e" % (name, err))
        assert err <= F16_TOL, name
        worst = max(worst, err)
    _note_f16_margin("test_tsm_f16_mode_tracks_the_oracle", worst)
    gen.close()


def test_bench_rccl_allgather_world1():
    """bench.py's RCCL path under the driver's `pytest -m gpu`: BSR_BENCH_FORCE_DIST=1 makes the single rank create an RCCL
    communicator and run the double-buffered all_gather_into_tensor of the packed con_rgb|dif payload every step; bench.py itself
    checks the gathered buffer against the packed outputs (`allgather.verified`)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))                 # a port that is free right now (a fixed one may be taken on a shared box)
    port = sock.getsockname()[1]
    sock.close()
    env.update(BSR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "1",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["dtype"] == "f32" and j["value"] > 0
    assert j["config"]["collective"].startswith("all_gather")
    ag = j["config"]["allgather"]
    assert ag["bytes_per_rank"] == 33554432                 # 32 x 256 x 256 x 4 channels x 4 B
    assert ag["verified"] is True and ag["backend"] == "nccl"


def test_peer_gather_world2_on_one_gpu_and_in_bench():
    """Round 5: the copy-engine alternative to the RCCL all-gather (blindshadowremoval_amd/peer_gather.py: every rank writes its shard into
    every peer's IPC-mapped gather buffer, a gloo barrier completes the step).  (1) its self-test with TWO ranks sharing GPU 0: handles
    exchanged, buffers mapped across processes, six double-buffered steps, every shard checked on every rank; (2) bench.py --gather
    peer on one rank (forced process group): the same `allgather.verified` bookkeeping as the RCCL path."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def free_port():
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        return port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        "-m", "blindshadowremoval_amd.peer_gather", "--device", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j == {"peer_gather_selftest": True, "world": 2, "steps": 6, "device_of_every_rank": 0}
    env.update(BSR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "1", "--gather", "peer",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ag = j["config"]["allgather"]
    assert ag["verified"] is True and ag["form"].startswith("peer") and ag["bytes_per_rank"] == 33554432 and j["value"] > 0


def test_bench_gpus2_real_generator_on_one_gpu():
    """Round 6: `bench.py --gpus 2 --device 0` — the N > 1 step of the headline bench with the REAL generator where only one GPU exists: two
    rank processes (started by bench.py itself, before anything touches the GPU) share GPU 0, gloo is the control plane, every step is
    bsr_forward_packed -> the double-buffered peer-copy gather of con_rgb | dif into every rank's IPC-mapped buffer -> the next step's
    forward beside it; at the end every rank checks its own shard bit for bit and the other rank's by checksum (`allgather.verified`).
    The rate it prints is two forwards sharing a chip, not a scaling figure: the line says so.  (The reference's loop is single-device:
    /root/reference/train_test_GSC.py:854-858.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BSR_BENCH_FORCE_DIST")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--device", "0", "--steps", "4", "--warmup", "2", "--repeats", "1",
                        "--batch", "8", "--no-cpu-baseline", "--no-secondary", "--no-sustained", "--streams", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "stub" not in j and j["roofline"] is not None
    cfg = j["config"]
    assert cfg["global_batch"] == 16 and cfg["all_ranks_on_device"] == 0 and "ONE GPU" in cfg["parallelism"]
    ag = cfg["allgather"]
    assert ag["verified"] is True and ag["form"].startswith("peer") and ag["backend"] == "gloo" and ag["bytes_per_rank"] == 8 * 256 * 256 * 4 * 4
    assert len(cfg["per_rank_images_per_sec"]) == 2 and all(v > 0 for v in cfg["per_rank_images_per_sec"])


# ---------------------------------------------------------------------------------------------------------------------------------
# Range guard of the 16-bit modes (include/bsr_hip.h: BSR_ERR_RANGE, bsr_check_range)

_RANGE_PROBES = ("x1", "x2", "x3", "x0", "res0", "res1", "res2", "up1", "up2", "y", "res3", "res4", "res5", "f1", "f2", "f")


def _max_activation(gen):
    return max(float(gen.probe(k).abs().max()) for k in _RANGE_PROBES)


@pytest.mark.parametrize("dtype", ["f32x3", "f16"])
def test_range_guard_large_activations_and_overflow(dtype):
    """Inputs scaled until the largest activation of the net sits at ~4e4 (fp16 max 65504): the 16-bit mode must still track the fp32
    path (1e-3 RELATIVE to the output scale for f32x3) and report no range error.  Scaled 4x further, activations pass 65520: the
    forward must be REPORTED — check_range() raises RangeError, and so does the next forward until the condition is acknowledged —
    instead of silently returning inf / NaN.  The fp32 handle never reports."""
    from blindshadowremoval_amd import Generator
    from blindshadowremoval_amd._lib import RangeError
    w = init_weights(1)
    g32, g16 = Generator(dtype="f32").load_weights(w), Generator(dtype=dtype).load_weights(w)
    torch.manual_seed(61)
    inp, uv = torch.rand(2, 256, 256, 3).cuda(), torch.rand(2, 256, 256, 3).cuda()
    g32(inp, uv)
    a1 = _max_activation(g32)
    k = 4.0e4 / a1                                     # first guess; refine once (the net is only piecewise linear)
    g32(inp * k, uv * k)
    k *= 4.0e4 / _max_activation(g32)
    ref = [t.clone() for t in g32(inp * k, uv * k)]
    amax = _max_activation(g32)
    assert 2.0e4 < amax < 6.0e4, amax
    ref_enc = {n: g32.probe(n).clone() for n in ("x1", "x2", "x3")}
    out = g16(inp * k, uv * k)
    g16.check_range()                                  # in range: no error
    assert all(bool(torch.isfinite(t).all()) for t in out)
    # The comparison is made where it is well-posed: the encoder's conv + LeakyReLU chain (x1, x2, x3 — values up to ~1e4).  Past the
    # first NonLocalBlock the logits theta.phi grow with the SQUARE of the scale (~1e10 here, softmax = a hard argmax), so two
    # correct fp32 implementations already differ by O(1e-3) of the output scale there; the outputs are only required to be finite.
    rel_tol = 1e-3 if dtype == "f32x3" else 4e-3
    for n, b in ref_enc.items():
        a = g16.probe(n)
        err = float((a - b).abs().max()) / float(b.abs().max())
        print("range guard %s %s: max |act| %.3g (net %.3g), rel err %.2e" % (dtype, n, float(b.abs().max()), amax, err))
        assert float(b.abs().max()) > 1.0e3 and err <= rel_tol, n
    # 4x further: overflow
    big = [t.clone() for t in g32(inp * (4 * k), uv * (4 * k))]
    assert _max_activation(g32) > 7.0e4 and all(bool(torch.isfinite(t).all()) for t in big)     # the fp32 path stays finite
    g32.check_range()
    g16.peek_range()                                   # the pipelined loops' form (no stream sync, no clear): nothing raised so far
    g16(inp * (4 * k), uv * (4 * k))
    torch.cuda.synchronize()
    for _ in range(2):                                 # peek reports and does NOT clear
        with pytest.raises(RangeError, match="fp16 range"):
            g16.peek_range()
    g32.peek_range()                                   # an fp32 handle never reports
    with pytest.raises(RangeError, match="fp16 range"):
        g16(inp, uv)                                   # the completed overflowing forward is not silent: the next call refuses
    with pytest.raises(RangeError):
        g16.check_range()                              # ... and check_range reports and CLEARS
    out2 = g16(inp, uv)                                # acknowledged: the handle works again
    g16.check_range()
    ref2 = g32(inp, uv)
    for a, b in zip(out2, ref2):
        assert float((a - b).abs().max()) <= (1e-3 if dtype == "f32x3" else 4e-3)
    g32.close()
    g16.close()


def test_f32x3_tiny_operands_bound_the_absolute_error():
    """Operands below 2^-14 make the lo half of the split an fp16 subnormal (igemm_h16.h): the ERROR is then absolute (<= 2^-25 per
    operand), not relative.  Inputs of 1e-6: the f32x3 outputs stay within 2e-5 absolute of the fp32 path."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    g32, g16 = Generator(dtype="f32").load_weights(w), Generator(dtype="f32x3").load_weights(w)
    torch.manual_seed(62)
    inp, uv = (torch.rand(2, 256, 256, 3) * 1e-6).cuda(), (torch.rand(2, 256, 256, 3) * 1e-6).cuda()
    ref = [t.clone() for t in g32(inp, uv)]
    out = g16(inp, uv)
    g16.check_range()
    assert torch.equal(g16.probe("bmask"), g32.probe("bmask"))
    for a, b, name in zip(out, ref, ("gs", "con_rgb", "mask22", "dif")):
        err = float((a - b).abs().max())
        print("tiny operands %s abs err %.2e" % (name, err))
        assert err <= 2e-5, name
    g32.close()
    g16.close()


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "f16"])
def test_fused_heads_epilogue_is_bit_identical_to_the_two_launch_form(dtype, monkeypatch):
    """Round 3: when the batch has enough row strips to fill the chip, the heads kernel sums its 7 horizontal taps and emits gs /
    mask22 itself (csrc/conv_n16.h FUSE) instead of writing a [B,H,W,16] scratch tensor for heads_post_kernel.  Same operation order
    => the same bits, on every output (gs feeds the threshold and everything after it).  BSR_FUSE_HEADS=0 at handle creation forces
    the two-launch form."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    fused = Generator(dtype=dtype).load_weights(w)
    monkeypatch.setenv("BSR_FUSE_HEADS", "0")
    plain = Generator(dtype=dtype).load_weights(w)
    monkeypatch.delenv("BSR_FUSE_HEADS")
    g = torch.Generator().manual_seed(71)
    for (B, H, W) in ((32, 256, 256), (17, 256, 256), (16, 288, 256), (16, 256, 512)):
        inp, uv = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        fused.set_timing(True)
        a = [t.clone() for t in fused(inp, uv)]
        torch.cuda.synchronize()
        names = [n for n, _, _ in fused.get_launch_timing()]
        fused.set_timing(False)
        assert "heads" in names and "heads_post" not in names, (B, H, W)          # the fused form really ran
        b = plain(inp, uv)
        for x, y, name in zip(a, b, ("gs", "con_rgb", "mask22", "dif")):
            assert torch.equal(x, y), (dtype, B, H, W, name)
    fused.close()
    plain.close()


def test_fused_gemm_tails_are_bit_identical_to_the_separate_launches(monkeypatch):
    """Round 4: at full fp32 batches the NonLocalBlock's `w` conv + residual + LeakyReLU runs as the TAIL of the attention kernel (its 128
    queries = 128 pixels of the K = 128 GEMM; csrc/gemm_tail.h) — the same MFMA order per output element as the separate gemm_nloop
    launch => the same bits on every probe and output.  BSR_FUSE_ATTW=0 at handle creation forces the two launches small batches use.
    (The same tail behind res*.conv2 was built too and retired in round 6: no faster; profiles/HISTORY.md.)"""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    fused = Generator().load_weights(w)
    monkeypatch.setenv("BSR_FUSE_ATTW", "0")
    plain = Generator().load_weights(w)
    monkeypatch.delenv("BSR_FUSE_ATTW")
    g = torch.Generator().manual_seed(72)
    # (33, ...): 264 query blocks of 128 = two rounds of the 8-wave shape — the launcher prefers three rounds of the 4-wave one there,
    # and that shape keeps the two launches: still the same bits
    for (B, H, W, want_fused) in ((32, 256, 256, True), (64, 256, 256, True), (8, 512, 512, True), (33, 256, 256, False)):
        inp, uv = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        fused.set_timing(True)
        a = [t.clone() for t in fused(inp, uv)]
        torch.cuda.synchronize()
        names = [n for n, _, _ in fused.get_launch_timing()]
        fused.set_timing(False)
        if want_fused:
            assert "res0.attw" in names and "res5.attw" in names and "res0.w" not in names and "res0.attention" not in names, (B, H, W)
            assert "res0.conv2" in names and "res0.c3q" in names, (B, H, W)
        else:
            assert "res0.attention" in names and "res0.w" in names, (B, H, W)
        b = plain(inp, uv)
        for x, y, name in zip(a, b, ("gs", "con_rgb", "mask22", "dif")):
            assert torch.equal(x, y), (B, H, W, name)
        for pr in ("res0", "res2", "res3", "res5", "y3x0", "y3x4"):
            assert torch.equal(fused.probe(pr), plain.probe(pr)), (B, H, W, pr)
    fused.close()
    plain.close()


def test_packed_output_is_bit_identical(gen_w):
    """bsr_forward_packed: con_rgb | dif written as one [B,H,W,4] tensor by the tail kernel (the all-gather payload of bench.py /
    dist.py) — the same bits as the two separate outputs."""
    gen, _ = gen_w
    torch.manual_seed(81)
    inp, uv = torch.rand(5, 256, 256, 3).cuda(), torch.rand(5, 256, 256, 3).cuda()
    a = [t.clone() for t in gen(inp, uv)]
    packed = torch.full((5, 256, 256, 4), float("nan"), device="cuda")
    b = gen(inp, uv, packed_out=packed)
    assert b[1].data_ptr() == packed.data_ptr()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(packed, torch.cat((a[1], a[3]), dim=3))
    with pytest.raises(ValueError):
        gen(inp, uv, packed_out=packed[:4])


def test_smallest_accepted_image(gen_w):
    """H = 32 is the smallest height the C ABI accepts with W = 256 (128 attention tokens, one 4x32 tile row at 1/8 resolution)."""
    from parity_util import run_and_compare
    gen, w = gen_w
    torch.manual_seed(91)
    inp, uv = torch.rand(3, 32, 256, 3), torch.rand(3, 32, 256, 3)
    out, ref, errs, nflip = run_and_compare(gen, w, inp, uv, want_probes=("x0", "res2", "res5"))
    assert out[0].shape == (3, 32, 256, 1)


@pytest.mark.parametrize("dtype", ["f32x3", "f16"])
def test_fused_attention_w_tail_agrees_with_the_two_launch_form_in_the_16_bit_modes(dtype, monkeypatch):
    """Round 6: the one-wave-per-SIMD attention kernel of the 16-bit modes (attention_h16.h) runs the `w` GEMM as its tail with the normalised
    O^T accumulators as the A operand — no LDS round trip, the K order of the accumulator registers (pack.py `w4`), the residual as the C
    operand of each tile's first matrix instruction.  That is the arithmetic of the two-launch form (attention output to HBM, then
    gemm_nloop_kernel<3, 4, 2>) in another summation order, so the two agree to rounding, not to the bit: block outputs to 2e-5, the
    forward's outputs to 1e-4 whenever both forms take the same threshold decisions (model.py:256) — at every batch (this kernel has one
    workgroup shape), GSC and the wider TSM trunk alike.  In the f16 mode a last-bit difference of a block output can flip the fp16
    rounding of an activation the next 3x3 layer reads (2^-11 relative), so the bounds there are that mode's own scale: 5e-4 on the block
    outputs, F16_TOL on the outputs.  The round-5 kernel's test here was bit-identity; it went with that kernel."""
    from blindshadowremoval_amd import Generator, GeneratorTSM
    w = init_weights(1)
    fused = Generator(dtype=dtype).load_weights(w)
    monkeypatch.setenv("BSR_FUSE_ATTW", "0")
    plain = Generator(dtype=dtype).load_weights(w)
    monkeypatch.delenv("BSR_FUSE_ATTW")
    g = torch.Generator().manual_seed(73)

    def close(x, y, tol):
        return float((x.double() - y.double()).abs().max()) <= tol
    tol_blk, tol_blk2, tol_out = (2e-5, 5e-5, 1e-4) if dtype == "f32x3" else (5e-4, 1e-3, F16_TOL)

    for (B, H, W) in ((32, 256, 256), (3, 256, 256), (2, 512, 512), (2, 32, 256)):
        inp, uv = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        fused.set_timing(True)
        a = [t.clone() for t in fused(inp, uv)]
        torch.cuda.synchronize()
        names = [n for n, _, _ in fused.get_launch_timing()]
        fused.set_timing(False)
        assert "res0.attw" in names and "res5.attw" in names and "res0.w" not in names and "res0.attention" not in names, (B, H, W)
        fa = {pr: fused.probe(pr).clone() for pr in ("res0", "res2", "res3", "res5", "bmask")}
        plain.set_timing(True)
        b = plain(inp, uv)
        torch.cuda.synchronize()
        names_p = [n for n, _, _ in plain.get_launch_timing()]
        plain.set_timing(False)
        assert "res0.attention" in names_p and "res0.w" in names_p and "res0.attw" not in names_p
        fused.check_range()
        for pr in ("res0", "res2"):
            assert close(fa[pr], plain.probe(pr), tol_blk), (dtype, B, H, W, pr, float((fa[pr] - plain.probe(pr)).abs().max()))
        if torch.equal(fa["bmask"], plain.probe("bmask")):
            for pr in ("res3", "res5"):
                assert close(fa[pr], plain.probe(pr), tol_blk2), (dtype, B, H, W, pr, float((fa[pr] - plain.probe(pr)).abs().max()))
            for x, y, name in zip(a, b, ("gs", "con_rgb", "mask22", "dif")):
                assert close(x, y, tol_out), (dtype, B, H, W, name, float((x - y).abs().max()))
        with pytest.raises(RuntimeError, match="never left LDS"):
            fused.probe("att0")
        assert plain.probe("att0").shape == (B, H // 8, W // 8, 128)
    fused.close()
    plain.close()
    if dtype == "f32x3":
        wt = init_weights(1, variant="tsm")
        ft = GeneratorTSM(dtype=dtype).load_weights(wt)
        monkeypatch.setenv("BSR_FUSE_ATTW", "0")
        pt = GeneratorTSM(dtype=dtype).load_weights(wt)
        monkeypatch.delenv("BSR_FUSE_ATTW")
        inp, uv = torch.rand(4, 256, 256, 3, generator=g).cuda(), torch.rand(4, 256, 256, 3, generator=g).cuda()
        reg = ((torch.rand(4, 256, 256, 6, generator=g) - 0.5) * 0.2).cuda()
        a = [t.clone() for t in ft(inp, uv, reg, 2, True)]
        fa = {pr: ft.probe(pr).clone() for pr in ("res2", "res5", "bmask")}
        b = pt(inp, uv, reg, 2, True)
        assert close(fa["res2"], pt.probe("res2"), 2e-5), ("tsm", "res2")
        if torch.equal(fa["bmask"], pt.probe("bmask")):
            assert close(fa["res5"], pt.probe("res5"), 5e-5), ("tsm", "res5")
            for x, y, name in zip(a, b, ("gs", "con_rgb", "mask22", "dif")):
                assert close(x, y, 1e-4), ("tsm", name)
        ft.close()
        pt.close()


@pytest.mark.parametrize("dtype", ["f32x3", "f16"])
def test_conv1_resident_gemm_is_bit_identical_to_the_implicit_gemm_form(dtype, monkeypatch):
    """Round 5: at full batches res*.conv1 (1x1, 99 | 257 | 261 -> 128) runs as gemm_nloop_kernel<4, NCH, H, MINW = 1> — one workgroup per
    CU, the whole input tile resident in registers, all of N per workgroup — instead of the implicit-GEMM kernels (igemm_conv_kernel<1,1,1,
    CC = 24> / igemm_h16_kernel<1,1,1>) in the 16-bit modes.  Same operands (the same hi / lo split) and the same matrix-instruction order
    per output element => the same bits; small batches keep the implicit-GEMM form.  (On the fp32 path the resident form was built too,
    bit-identical and no faster: retired in round 6.)"""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    new = Generator(dtype=dtype).load_weights(w)
    monkeypatch.setenv("BSR_CONV1_GEMM", "0")
    old = Generator(dtype=dtype).load_weights(w)
    monkeypatch.delenv("BSR_CONV1_GEMM")
    g = torch.Generator().manual_seed(74)
    for (B, H, W) in ((32, 256, 256), (16, 256, 256), (8, 512, 512), (3, 256, 256)):
        inp, uv = torch.rand(B, H, W, 3, generator=g).cuda(), torch.rand(B, H, W, 3, generator=g).cuda()
        a = [t.clone() for t in new(inp, uv)]
        b = old(inp, uv)
        for x, y, name in zip(a, b, ("gs", "con_rgb", "mask22", "dif")):
            assert torch.equal(x, y), (dtype, B, H, W, name)
        for pr in ("res0", "res1", "res3", "res5"):
            assert torch.equal(new.probe(pr), old.probe(pr)), (dtype, B, H, W, pr)
    new.close()
    old.close()


@pytest.mark.gpu
def test_two_handles_on_two_streams_give_the_serial_result():
    """bench.py --streams 2 (and any serving loop) keeps two forwards in flight on two handles / two HIP streams: the kernels of one
    run beside the other's.  Every output must be bit-identical to the same forward running alone."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    torch.manual_seed(5)
    xs = [torch.rand(8, 256, 256, 3).cuda() for _ in range(4)]
    us = [torch.rand(8, 256, 256, 3).cuda() for _ in range(4)]
    gens = [Generator(device=0).load_weights(w) for _ in range(2)]
    serial = [[t.clone() for t in gens[0](x, u)] for x, u in zip(xs, us)]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        outs = []
        for i, (x, u) in enumerate(zip(xs, us)):
            with torch.cuda.stream(lanes[i & 1]):
                outs.append(gens[i & 1](x, u))
        torch.cuda.synchronize()
        for o, r in zip(outs, serial):
            for a, b in zip(o, r):
                assert torch.equal(a, b)
    for g in gens:
        g.close()


def test_att_probe_refuses_after_a_fused_forward(monkeypatch):
    """At full fp32 batches attention and the `w` GEMM are ONE launch and the attention output never leaves LDS: bsr_probe("att<i>")
    must say so (BSR_ERR_STATE) instead of copying whatever an earlier forward left in the workspace slot.  Forms that do write it
    (small batches -> the 2-wave attention shape; BSR_FUSE_ATTW=0) still serve the probe, and both agree on the block outputs."""
    from blindshadowremoval_amd import Generator
    w = init_weights(1)
    torch.manual_seed(63)
    inp, uv = torch.rand(32, 256, 256, 3).cuda(), torch.rand(32, 256, 256, 3).cuda()
    g = Generator(dtype="f32").load_weights(w)
    g(inp[:2], uv[:2])                                 # B = 2: separate launches, the slot is written
    att_small = g.probe("att0").clone()
    assert att_small.shape == (2, 32, 32, 128) and float(att_small.abs().max()) > 0
    g(inp, uv)                                         # B = 32: fused
    with pytest.raises(RuntimeError, match="never left LDS"):
        g.probe("att0")
    res0 = g.probe("res0").clone()
    g.close()
    monkeypatch.setenv("BSR_FUSE_ATTW", "0")
    g2 = Generator(dtype="f32").load_weights(w)
    g2(inp, uv)
    att = g2.probe("att0")
    assert att.shape == (32, 32, 32, 128) and torch.equal(att[:2], att_small)
    assert torch.equal(g2.probe("res0"), res0)
    g2.close()
