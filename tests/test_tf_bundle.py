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


@pytest.mark.skipif(not os.path.isdir("/root/reference/log"), reason="reference checkpoint indices are only in the build container")
def test_reader_follows_the_reference_index_offsets(tmp_path):
    """The reference ships ckpt-94.index without its data shard.  A synthetic shard of the size the REAL index implies (seeded
    floats at every byte) is read through the real index: every generator tensor must be exactly the bytes at the [offset,
    offset + size) the index names, in its shape — so the reader is exercised on an index written by TensorFlow, not only on
    files of this repo's own writer; the entries themselves must be float32, non-overlapping and of size 4 * prod(shape)."""
    import shutil
    idx = glob.glob("/root/reference/log/*reweight-gradients/ckpt-94.index")[0]
    entries = {k: e for k, e in tf_bundle.read_index(idx).items() if k}
    spans = sorted((e.offset, e.offset + e.size) for e in entries.values())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "tensor byte ranges overlap"
    total = spans[-1][1]
    rng = np.random.default_rng(94)
    blob = rng.standard_normal(total // 4 + 1).astype("<f4").tobytes()[:total]
    prefix = str(tmp_path / "ckpt-94")
    shutil.copy(idx, prefix + ".index")
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(blob)
    w = tf_bundle.load_generator_weights(prefix)
    inv = tf_bundle.generator_inventory(idx)
    assert set(w) == set(inv) and len(w) == 258
    checked = 0
    for key, e in entries.items():
        name = key.replace("generator/", "", 1).replace("/.ATTRIBUTES/VARIABLE_VALUE", "")
        if name in w and key.startswith("generator/") and ".OPTIMIZER_SLOT" not in key:
            assert e.dtype == 1 and e.size == 4 * int(np.prod(e.shape)), (key, e)
            want = np.frombuffer(blob[e.offset:e.offset + e.size], "<f4").reshape(e.shape)
            np.testing.assert_array_equal(w[name], want)
            checked += 1
    assert checked == 258
