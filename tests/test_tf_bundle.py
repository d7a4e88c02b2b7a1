"""TF tensor-bundle reader: round trip through our writer, plus the reference's real ckpt-94.index
when /root/reference is mounted (build container only)."""
import glob
import json
import os

import numpy as np
import pytest

from blindshadowremoval_amd import tf_bundle
from blindshadowremoval_amd.weights import init_weights


def test_round_trip(tmp_path):
    w = init_weights(9)
    prefix = str(tmp_path / "ckpt-7")
    tf_bundle.write_bundle(prefix, w)
    assert tf_bundle.latest_checkpoint(str(tmp_path)) == prefix
    inv = tf_bundle.generator_inventory(prefix + ".index")
    assert {k: tuple(v.shape) for k, v in w.items()} == {k: tuple(v) for k, v in inv.items()}
    back = tf_bundle.load_generator_weights(prefix)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])


def test_missing_data_shard_and_bad_magic(tmp_path):
    w = {"conv1/conv/bias": np.zeros(32, np.float32)}
    prefix = str(tmp_path / "ckpt-1")
    tf_bundle.write_bundle(prefix, w)
    os.remove(prefix + ".data-00000-of-00001")
    with pytest.raises(FileNotFoundError):
        tf_bundle.load_generator_weights(prefix)
    with open(prefix + ".index", "wb") as f:
        f.write(b"\0" * 100)
    with pytest.raises(ValueError):
        tf_bundle.read_index(prefix + ".index")
    assert tf_bundle.latest_checkpoint(str(tmp_path / "nope")) is None


@pytest.mark.skipif(not os.path.isdir("/root/reference/log"), reason="reference checkpoint indices are only in the build container")
def test_reference_index_matches_fixture(golden_dir):
    with open(os.path.join(golden_dir, "gsc_ckpt94_inventory.json")) as f:
        fix = json.load(f)["gsc"]
    idx = glob.glob("/root/reference/log/*reweight-gradients/ckpt-94.index")[0]
    inv = tf_bundle.generator_inventory(idx)
    assert {k: list(v) for k, v in inv.items()} == fix["variables"]
    # optimizer slots and discriminators are present in the index but not part of the generator inventory
    allkeys = tf_bundle.read_index(idx)
    assert any(".OPTIMIZER_SLOT" in k for k in allkeys) and any(k.startswith("discriminator_1/") for k in allkeys)
