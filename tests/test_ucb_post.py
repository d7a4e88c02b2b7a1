"""the UCB post-processing restatement (blindshadowremoval_amd/ucb_post.py) against the outputs of the
reference's own `FSRNet.test_step` code (tests/golden/ucb_post_9156.npz, tools/make_ucb_post_fixture.py) on the same inputs."""
import os

import numpy as np
import pytest

from blindshadowremoval_amd.ucb_post import resize_bilinear, ucb_postprocess
from ucb_cases import GOLDEN, cases

FIX = np.load(os.path.join(GOLDEN, "ucb_post_9156.npz"))


def test_fixture_covers_small_and_large_detections():
    assert str(FIX["backend"]) in ("standin",) or str(FIX["backend"]).startswith("tf-")          # which arithmetic made the expected values
    det = {k: int(FIX[k].sum()) for k in FIX.files if k.endswith("_detected")}
    assert len(det) == 10 and any(v < 2000 for v in det.values()) and any(v > 20000 for v in det.values())


@pytest.mark.parametrize("case", list(cases()), ids=lambda c: c[0])
def test_matches_reference_code(case):
    key, row, box, masks, con, dif = case
    with np.errstate(invalid="ignore", divide="ignore"):      # the reference divides by an empty mask's area in the all-rejected case
        losses, figs = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, masks)
    assert len(figs) == 7 and all(f.shape == (1, 256, 256, 3) and f.dtype == np.float32 for f in figs)
    detected = figs[4][0, :, :, 0]
    assert set(np.unique(detected)) <= {0.0, 1.0}
    np.testing.assert_array_equal(detected.astype(np.uint8), FIX[key + "_detected"])      # every threshold / component decision
    if key + "_out" in FIX.files:
        np.testing.assert_allclose(figs[1][0], FIX[key + "_out"].astype(np.float32), atol=1e-3)   # fixture stored as fp16
    assert abs(losses["ssim"] - float(FIX[key + "_ssim"])) < 1e-4
    assert abs(losses["psnr"] - float(FIX[key + "_psnr"])) < 1e-3
    # composite only inside the detected mask (train_test_GSC.py:714)
    outside = detected == 0
    np.testing.assert_allclose(figs[1][0][outside], np.clip(figs[0][0][outside], 0, 1), atol=1e-6)


def test_resize_is_tf_half_pixel_bilinear():
    x = np.arange(16, dtype=np.float32).reshape(4, 4, 1)
    y = resize_bilinear(x, 2)                        # 2x down-sampling: sample points at 0.5, 2.5 -> plain 2x2 means
    np.testing.assert_allclose(y[:, :, 0], [[2.5, 4.5], [10.5, 12.5]])
    z = resize_bilinear(x, 8)                        # up-sampling clamps at the border
    assert z.shape == (8, 8, 1) and z[0, 0, 0] == 0.0 and z[-1, -1, 0] == 15.0


def test_single_channel_masks_give_the_same_result():
    """run_post_job reads the masks as ONE channel (read_masks(grey=True)); ucb_postprocess must not care."""
    key, row, box, masks, con, dif = next(iter(cases()))
    grey = {k: v[:, :, 0:1] for k, v in masks.items()}
    with np.errstate(invalid="ignore", divide="ignore"):
        l3, f3 = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, masks)
        l1, f1 = ucb_postprocess(row[..., 0:3], row[..., 3:6], con, dif, box, grey)
    assert l1 == l3
    for a, b in zip(f1, f3):
        np.testing.assert_array_equal(a, b)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="runs the reference's test_step source (build container only)")
def test_tf_backend_of_the_fixture_tool_dry_run(tmp_path, monkeypatch):
    """`tools/make_ucb_post_fixture.py --backend tf` is the one command that would pin tf.image.resize / ssim / psnr (and every decision
    that follows from them) to real TensorFlow; it cannot run here.  Its code path can: with BSR_MOCK_TF=1 the stand-in is presented as
    the `tensorflow` module and the tool must write the committed fixture's values under `backend = "tf-mock"`.  A test of the tool, not a pin."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_ucb_post_fixture", os.path.join(root, "tools", "make_ucb_post_fixture.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    monkeypatch.setenv("BSR_MOCK_TF", "1")
    out = tmp_path / "fix.npz"
    with np.errstate(invalid="ignore", divide="ignore"):
        tool.main(["--backend", "tf", "--out", str(out)])
    z = np.load(out)
    assert str(z["backend"]) == "tf-mock" and set(z.files) == set(FIX.files)
    for k in FIX.files:
        if k != "backend":
            np.testing.assert_array_equal(z[k], FIX[k], err_msg=k)
    with pytest.raises(SystemExit):
        tool.main(["--backend", "jax"])
