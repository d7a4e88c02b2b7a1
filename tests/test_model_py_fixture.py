"""The oracle (and, under -m gpu, the HIP path) against tests/golden/model_py_*.npz: outputs of the reference's OWN
`model.py` / `model_with_TSM.py` source executed over a TensorFlow stand-in (tools/make_model_fixture.py).

This pins the WIRING the reference's text states — layer order, concat orders (model.py:238,244,245,252,259,267), the tail
zero-pad (:105-112), token order (:36-54), the threshold (:256), the ShareLayer reshape/stack (model_with_TSM.py:204-229), the
attribute tree behind the checkpoint names — NOT TensorFlow's op arithmetic: the stand-in's ops are oracle/np_loops.py's
KAT-checked primitives in float64 (parity stays "unpinned" for SURVEY A.1-A.7).  The torch oracle shares no code with that
stand-in, so an agreement to ~1e-5 means two independent restatements driven by two different control flows (ours, the
reference's) coincide."""
import os
import sys

import numpy as np
import pytest
import torch

from blindshadowremoval_amd.weights import init_weights
from oracle.gsc_oracle import GeneratorOracle, GeneratorTSMOracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 5e-5          # fp32 torch oracle vs float64 stand-in through ~40 layers (measured 3e-6)
NAMES = ("gs", "con_rgb", "mask22", "dif")


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: z[k] for k in z.files}


def _check_backend(z):
    """Fixtures made by `tools/make_model_fixture.py --backend tf` (real TensorFlow executing the reference's model.py) carry a
    "backend" key "tf-<version>"; the stand-in ones carry "standin-np_loops" or (older files) no key.  Anything else is a
    fixture this test does not know how to interpret."""
    if "backend" in z:
        b = str(z["backend"])
        assert b.startswith("tf-") or b == "standin-np_loops", b


def _f(a):
    return torch.from_numpy(np.asarray(a, np.float32))


@pytest.mark.parametrize("fixture", ["model_py_gsc_64.npz", "model_py_gsc_256.npz"])
def test_oracle_reproduces_reference_model_py(fixture):
    z = _load(fixture)
    assert float(z["min_abs_d32_minus_thr"]) > 2e-5          # the fixture's threshold decisions are not marginal (oracle error ~3e-6)
    oracle, pr = GeneratorOracle(init_weights(int(z["weights_seed"]))), {}
    out = oracle(_f(z["inputs"]), _f(z["uv"]), probes=pr)
    _check_backend(z)
    assert float(np.abs(pr["d32"].numpy() - z["d32"]).max()) < TOL and np.array_equal(pr["bmask"].numpy(), z["bmask"])
    for o, n in zip(out, NAMES):
        assert o.shape == z[n].shape, n
        assert float(np.abs(o.numpy() - z[n]).max()) < TOL, n


@pytest.mark.parametrize("fixture", ["model_py_tsm_64.npz", "model_py_tsm_256.npz"])
def test_tsm_oracle_reproduces_reference_model_with_tsm_py(fixture):
    z = _load(fixture)
    assert float(z["min_abs_d32_minus_thr"]) > 2e-5
    oracle, pr = GeneratorTSMOracle(init_weights(int(z["weights_seed"]), variant="tsm")), {}
    out = oracle(_f(z["inputs"]), _f(z["uv"]), _f(z["reg"]), int(z["frame"]), True, probes=pr)
    _check_backend(z)
    assert float(np.abs(pr["d32"].numpy() - z["d32"]).max()) < TOL
    assert np.array_equal(pr["bmask"].numpy(), z["bmask"])
    for o, n in zip(out, NAMES):
        assert float(np.abs(o.numpy() - z[n]).max()) < TOL, n


@pytest.mark.gpu
def test_hip_reproduces_reference_model_py():
    """The HIP path itself against the reference-source fixture (256x256: the C ABI needs W % 256 == 0)."""
    from blindshadowremoval_amd import Generator
    z = _load("model_py_gsc_256.npz")
    gen = Generator().load_weights(init_weights(int(z["weights_seed"])))
    out = [o.cpu().numpy() for o in gen(_f(z["inputs"]).cuda(), _f(z["uv"]).cuda())]
    assert np.array_equal(gen.probe("bmask").cpu().numpy(), z["bmask"])
    assert float(np.abs(gen.probe("d32").cpu().numpy() - z["d32"]).max()) < 1e-3
    for o, n in zip(out, NAMES):
        assert float(np.abs(o - z[n]).max()) < 1e-3, n
    gen.close()


@pytest.mark.gpu
def test_hip_tsm_reproduces_reference_model_with_tsm_py():
    from blindshadowremoval_amd import GeneratorTSM
    z = _load("model_py_tsm_256.npz")
    gen = GeneratorTSM().load_weights(init_weights(int(z["weights_seed"]), variant="tsm"))
    out = [o.cpu().numpy() for o in gen(_f(z["inputs"]).cuda(), _f(z["uv"]).cuda(), _f(z["reg"]).cuda(), int(z["frame"]), True)]
    assert np.array_equal(gen.probe("bmask").cpu().numpy(), z["bmask"])
    assert float(np.abs(gen.probe("d32").cpu().numpy() - z["d32"]).max()) < 1e-3
    for o, n in zip(out, NAMES):
        assert float(np.abs(o - z[n]).max()) < 1e-3, n
    gen.close()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="imports the reference's model.py (build container only)")
@pytest.mark.parametrize("variant", ["gsc", "tsm"])
def test_tf_backend_driver_dry_run_over_mock_tensorflow(variant, monkeypatch):
    """`tools/make_model_fixture.py --backend tf` cannot run here (no TensorFlow), but its DRIVER can: over a mock `tensorflow` module
    that has Keras' variable mechanics (variables created on the first call, `model.variables`, `.assign`) and the stand-in's
    arithmetic, the tf code path — build by one forward, assignment through the checkpoint attribute paths incl. `res_stack/<i>/...`,
    the every-variable-assigned check, the keyword call of the reference's call sites, the d32 spy on `tf.image.resize` — must
    reproduce the committed stand-in fixture exactly.  This is a test of the tool's logic, not a pin of TensorFlow's arithmetic."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_model_fixture", os.path.join(root, "tools", "make_model_fixture.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    monkeypatch.setenv("BSR_MOCK_TF", "1")
    z = _load("model_py_%s_64.npz" % variant)
    w = init_weights(int(z["weights_seed"]), variant=variant)
    inp, uv = z["inputs"].astype(np.float32), z["uv"].astype(np.float32)
    if variant == "gsc":
        args = (inp, uv, None, 1, False)
        outs, probes, backend = tool.run_reference_tf("model.py", w, args)
    else:
        args = (inp, uv, z["reg"].astype(np.float32), int(z["frame"]), True, 1, False)
        outs, probes, backend = tool.run_reference_tf("model_with_TSM.py", w, args)
    assert backend == "tf-mock"
    assert getattr(sys.modules.get("tensorflow"), "__version__", "") != "mock"          # the mock does not outlive the call
    for o, n in zip(outs, NAMES):
        assert np.array_equal(o.astype(np.float32), z[n]), n
    assert np.array_equal(probes["d32"].astype(np.float32), z["d32"])
    # a checkpoint name that does not exist in the model, or a model variable that no name covers, must fail loudly
    bad = dict(w)
    bad.pop("conv1/conv/bias")
    with pytest.raises(RuntimeError, match="variables"):
        tool.run_reference_tf("model.py" if variant == "gsc" else "model_with_TSM.py", bad, args)
