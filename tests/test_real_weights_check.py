"""tools/real_weights_check.py: the hook for the day a trained checkpoint exists — exercised on a tensor bundle
written from init_weights: restore through tf_bundle, three dtypes on the sample + UCB items, range guard / activation / cross-mode report."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("real_weights_check", os.path.join(ROOT, "tools", "real_weights_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tool_refuses_an_index_without_its_data_shard(tmp_path):
    """Upstream ships ckpt-94.index without ckpt-94.data-00000-of-00001: the tool must say so (tf_bundle names the missing shard), not invent
    weights — checked on a bundle written here whose data shard is then removed.  Runs without a GPU: the refusal comes before any is needed."""
    import glob
    from blindshadowremoval_amd import tf_bundle
    from blindshadowremoval_amd.weights import init_weights
    tf_bundle.write_bundle(str(tmp_path / "ckpt-94"), init_weights(2))
    for f in glob.glob(str(tmp_path / "ckpt-94.data-*")):
        os.remove(f)
    with pytest.raises((FileNotFoundError, OSError, ValueError), match="data"):
        _tool().run(str(tmp_path), [], None, 1)
    with pytest.raises(SystemExit, match="no checkpoint"):
        _tool().run(str(tmp_path / "nothing_here"), [], None, 1)


@pytest.mark.gpu
def test_report_on_a_bundle_written_from_seeded_weights(tmp_path):
    from blindshadowremoval_amd import tf_bundle
    from blindshadowremoval_amd.weights import init_weights
    tf_bundle.write_bundle(str(tmp_path / "ckpt-3"), init_weights(1))
    out = str(tmp_path / "report.json")
    rc = _tool().main([str(tmp_path), "--limit", "6", "--json", out])
    rep = json.load(open(out))
    assert rc == 0 and rep["f32x3_usable"] and rep["f16_usable"] and rep["items"] == 6 and rep["variables"] == 258
    assert rep["checkpoint"].endswith("ckpt-3")
    d = rep["dtypes"]
    assert set(d) == {"f32", "f32x3", "f16"} and all(v["range_guard"] == "ok" and v["outputs_finite"] for v in d.values())
    assert 1.0 < d["f32"]["max_activation"] < 65504.0 and d["f32"]["fp16_headroom"] > 10
    if d["f32x3"]["bmask_flips_vs_f32"] == 0:
        assert max(d["f32x3"]["max_abs_diff_vs_f32"].values()) <= 1e-4
    if d["f16"]["bmask_flips_vs_f32"] == 0:
        assert max(d["f16"]["max_abs_diff_vs_f32"].values()) <= 4e-3
