"""Harness counterpart of FSRNet.testFFHQ / test (train_test_GSC.py:360-422, 840-890) on the real-input fixture
built from /root/reference/sample_imgs/02165 by the reference's own input preparation (tools/make_sample_fixture.py)."""
import os

import numpy as np
import pytest
import torch

from blindshadowremoval_amd.weights import init_weights


class OneSampleDataset:
    """What dataset.py hands the loops: .name_list and .feed yielding (img[1,10,256,256,16], box[1,4], name)."""

    def __init__(self, golden_dir, copies=1):
        z = np.load(os.path.join(golden_dir, "sample_02165.npz"))
        self.row = z["row"]
        chunk = np.stack([self.row] * 10, axis=0)[None]                 # dataset.py:764-768, batch(1)
        self.name_list = ["sample_imgs/02165/02165.npy"] * copies
        self.feed = iter([(chunk, z["box"][None], np.array([b"sample_imgs/02165/02165.png"]))] * copies)


def test_fixture_layout(golden_dir):
    ds = OneSampleDataset(golden_dir)
    row = ds.row
    assert row.shape == (256, 256, 16) and row.dtype == np.float32
    np.testing.assert_array_equal(row[..., 0:3], row[..., 3:6])          # gt == img for FFHQ (dataset.py:625,638)
    assert float((row[..., 6:9] == 0).mean()) > 0.5                      # uv is 0 outside the landmark hull (warp.py:231)
    assert float(np.abs(row[..., 11]).max()) == 0 and float(np.abs(row[..., 14]).max()) == 0   # 3rd reg channels are x*0
    assert 0.0 <= row[..., 15].min() and row[..., 15].max() <= 1.0       # blurred face mask


def test_logging_strip(tmp_path):
    from blindshadowremoval_amd.fsrnet import Config, Logging
    cfg = Config()
    cfg.CHECKPOINT_DIR = str(tmp_path)
    log = Logging(cfg)
    a = torch.rand(2, 8, 8, 3) * 1.5 - 0.2
    m = torch.rand(2, 8, 8, 1)
    strip = log.get_imgs([a, m])
    assert strip.shape == (8, 16, 3) and strip.dtype == np.uint8
    np.testing.assert_array_equal(strip[:, :8], np.rint(np.clip(a[0].numpy(), 0, 1) * 255).astype(np.uint8))
    np.testing.assert_array_equal(strip[:, 8:, 0], strip[:, 8:, 2])      # grey replicated to 3 channels
    out = log.save_img([a, m], "sample_imgs/02165/02165.png")
    assert out.endswith("02165_02165-result.png") and os.path.isfile(out)


@pytest.mark.gpu
def test_testFFHQ_on_sample_matches_oracle(golden_dir, tmp_path):
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    from oracle.gsc_oracle import GeneratorOracle, test_step_ffhq
    w = init_weights(1)
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    fsr = FSRNet(cfg, weights=w)
    results = fsr.testFFHQ(OneSampleDataset(golden_dir, copies=3), batch=2)     # 3 elements, batched 2 + 1
    assert len(results) == 3 and len(fsr.log.saved) == 3 and all(os.path.isfile(p) for p in fsr.log.saved)
    row = torch.from_numpy(OneSampleDataset(golden_dir).row)[None]
    ref = test_step_ffhq(GeneratorOracle(w), row)
    for name, figs in results:
        assert name == "sample_imgs/02165/02165.npy"
        for a, b in zip(figs, ref):
            assert float((a.cpu() - b).abs().max()) <= 1e-3
    # the reference's own 10-row element forward keeps row 0: identical to the row-0-only forward
    ds = OneSampleDataset(golden_dir)
    element = next(ds.feed)
    _, figs10 = fsr.test_step_FFHQ(element[0], element[1], training=False, all_rows=True)
    _, figs1 = fsr.test_step_FFHQ(element[0], element[1], training=False)
    assert figs10[1].shape[0] == 10 and figs1[1].shape[0] == 1
    for a, b in zip(figs10, figs1):
        assert torch.equal(a[:1], b)
    # UCB loop head (generator outputs)
    res = fsr.test(OneSampleDataset(golden_dir), batch=4)
    assert len(res) == 1 and res[0][1][1].shape == (1, 256, 256, 1)


@pytest.mark.gpu
def test_restore_from_tensor_bundle(tmp_path):
    """FSRNet restores generator weights from a TF tensor-bundle checkpoint directory (train_test_GSC.py:842-845)."""
    from blindshadowremoval_amd import tf_bundle
    from blindshadowremoval_amd.fsrnet import Config, FSRNet
    w = init_weights(2)
    tf_bundle.write_bundle(str(tmp_path / "ckpt-94"), w)
    cfg = Config(0)
    cfg.CHECKPOINT_DIR = str(tmp_path)
    fsr = FSRNet(cfg)
    assert fsr._restore() == 94
    torch.manual_seed(1)
    inp, uv = torch.rand(1, 256, 256, 3), torch.rand(1, 256, 256, 3)
    from parity_util import run_and_compare
    run_and_compare(fsr.gen, w, inp, uv)
    empty = Config(0)
    empty.CHECKPOINT_DIR = str(tmp_path / "none")
    with pytest.raises(RuntimeError, match="no generator weights"):
        FSRNet(empty)._restore()
