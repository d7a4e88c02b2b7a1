"""Generate tests/golden/model_py_*.npz — the reference's OWN model source executed over a TensorFlow stand-in.

Runs IN THE BUILD CONTAINER ONLY (it imports /root/reference/model.py and /root/reference/model_with_TSM.py; nothing of them is
stored in the repo or travels to the GPU box — only inputs/outputs do).

WHAT THIS PINS: the WIRING of the hot path as the reference's text states it — `Generator.call` (model.py:228-290: layer order, the
concat orders at :238,:244,:245,:252,:259,:267, the strict-greater threshold at :256, x_hole at :258), `ResBottleneck.call`
(:98-113: conv/BN/LeakyReLU order, NonLocal before the skip, channel zero-pad on the tail :105-112), `NonLocalBlock.call` (:23-61:
token order of the reshapes :36-54, theta x phi^T, softmax axis, no 1/sqrt(d), BN after `w`, residual), `Conv.call` / `ConvT.call`
(:139-177), the Keras attribute tree the checkpoint names follow (:198-226), and for the TSM variant
`ShareLayer.call` (model_with_TSM.py:204-229: warp -> [1, frame, ...] reshape -> max|mean concat -> stack -> unwarp, the tf.cond)
with `Generator.call` (:261-325) and the reference's own `warp.tf_batch_map_offsets` / `tf_batch_map_coordinates`
(warp.py:71-165, executed from its source over the same stand-in).

WHAT THIS DOES NOT PIN: TensorFlow's op arithmetic.  `tensorflow` is not installable here, so every tf.* / keras op the reference
calls is provided below with the semantics of SURVEY.md Appendix A, built on the known-answer-tested primitives of
oracle/np_loops.py (float64 explicit loops; NOT oracle/gsc_oracle.py, so that the torch oracle the GPU tests use is checked
against an independent form driven by the reference's own control flow).  Parity therefore stays "unpinned" for A.1-A.7.

Usage:  python tools/make_model_fixture.py [--backend standin|tf] [gsc64 tsm64 gsc256 tsm256]
        (default: all four cases over the stand-in; its 256x256 cases take a few minutes: the primitives are Python loops)

--backend tf  — THE PIN THAT IS MISSING HERE.  On any machine with TensorFlow 2.3 (README.md:11 of the reference; tensorflow_addons
and cv2 are stubbed if absent, they are imported by model.py / warp.py but never executed on this path) the same cases run the
reference's model.py / model_with_TSM.py UNMODIFIED over REAL TensorFlow: `Generator()` is built by one forward, `init_weights(seed)`
is assigned variable by variable through the checkpoint attribute paths (train_test_GSC.py:143-148), `Generator.call(...,
training=False)` runs on the same seeded inputs and the SAME npz keys are written, plus `backend = "tf-<version>"`.  The tests
(tests/test_model_py_fixture.py) need no change: with such files in tests/golden the oracle and the HIP path are compared with
TensorFlow's own op arithmetic (SAME padding, Conv2DTranspose scatter / crop, BN epsilon, LeakyReLU alpha, half-pixel resize:
SURVEY A.1-A.7) and DESIGN.md's "parity unpinned" can be struck.  It cannot run in the build container (no TensorFlow wheel, no
network) — `--backend tf` fails loudly there.
"""
import contextlib
import importlib.util
import inspect
import io
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import np_loops as P                                   # noqa: E402
from blindshadowremoval_amd.weights import init_weights            # noqa: E402


# ------------------------------------------------------------------------------------------------ tensors
class T(np.ndarray):
    """numpy array with the two EagerTensor methods the reference calls (`get_shape`, `.numpy`); `.shape` is a tuple, so
    `b, h, w, c = x.shape` and `x.shape[-1]` work as on a TF tensor."""

    def get_shape(self):
        return self.shape

    def numpy(self):
        return np.asarray(self)


def t(a, dtype=None):
    return np.asarray(a, dtype=dtype).view(T)


def _dtype(d):
    return {"float32": np.float64, "int32": np.int64}.get(d, d) if isinstance(d, str) else d


# ------------------------------------------------------------------------------------------------ keras stand-in
_TRAINING_CTX = []          # Keras call context: a layer called without `training` inherits the enclosing call's value


class Layer:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        sig = inspect.signature(self.call)
        if "training" in sig.parameters:
            bound = sig.bind_partial(*args, **kwargs)
            if "training" not in bound.arguments:
                kwargs["training"] = _TRAINING_CTX[-1] if _TRAINING_CTX else None
                training = kwargs["training"]
            else:
                training = bound.arguments["training"]
        else:
            training = _TRAINING_CTX[-1] if _TRAINING_CTX else None
        _TRAINING_CTX.append(training)
        try:
            return self.call(*args, **kwargs)
        finally:
            _TRAINING_CTX.pop()


def _need(layer, *names):
    for n in names:
        if getattr(layer, n, None) is None:
            raise RuntimeError("%s called without its variable '%s' (checkpoint name mismatch?)" % (type(layer).__name__, n))
        layer._used = True


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", name=None, **kw):
        assert padding == "same" and not kw
        self.filters, self.ksize, self.strides = filters, _pair(kernel_size), _pair(strides)
        assert self.strides[0] == self.strides[1]
        self.kernel = self.bias = None

    def call(self, x):
        _need(self, "kernel", "bias")
        assert self.kernel.shape == self.ksize + (x.shape[-1], self.filters), (self.kernel.shape, x.shape)     # HWIO
        return t(P.conv2d_same(np.asarray(x), self.kernel, self.bias, self.strides[0]))


class Conv2DTranspose(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", **kw):
        assert padding == "same" and not kw and _pair(kernel_size) == (3, 3) and _pair(strides) == (2, 2)
        self.filters = filters
        self.kernel = self.bias = None

    def call(self, x):
        _need(self, "kernel", "bias")
        assert self.kernel.shape == (3, 3, self.filters, x.shape[-1]), (self.kernel.shape, x.shape)           # [kh,kw,out,in]
        return t(P.conv2d_transpose_same(np.asarray(x), self.kernel, self.bias))


class BatchNormalization(Layer):
    def __init__(self, **kw):
        assert not kw
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def call(self, x, training=None):
        _need(self, "gamma", "beta", "moving_mean", "moving_variance")
        assert training is False, "the test paths run training=False (train_test_GSC.py:404,856): moving statistics"
        return t(P.batchnorm(np.asarray(x), self.gamma, self.beta, self.moving_mean, self.moving_variance))


class LeakyReLU(Layer):
    def __init__(self, **kw):
        assert not kw                    # Keras default alpha = 0.3

    def call(self, x):
        return t(P.lrelu(np.asarray(x)))


class _Unused(Layer):
    def call(self, *a, **k):
        raise RuntimeError("%s is constructed but never called on the inference path" % type(self).__name__)


class MaxPool2D(_Unused):
    pass


class Dropout(_Unused):
    pass


# ------------------------------------------------------------------------------------------------ tf.* stand-in
def _softmax(x, axis=-1):
    x = np.asarray(x, np.float64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return t(e / e.sum(axis=axis, keepdims=True))


def _resize(x, size, **kw):
    assert not kw
    return t(P.resize_bilinear(np.asarray(x), int(size[0]), int(size[1])))


def _greater(a, b):
    return t(np.asarray(a).astype(np.float32) > np.float32(b))          # fp32 compare, strict (model.py:256)


def _cast(x, dtype):
    d = _dtype(dtype)
    a = np.asarray(x)
    if np.issubdtype(np.dtype(d), np.integer) and np.issubdtype(a.dtype, np.floating):
        assert np.all(a == np.trunc(a))             # the reference only casts floor()/ceil() results to int32
    return t(a.astype(d))


def _gather_nd(params, indices):
    idx = np.asarray(indices)
    return t(np.asarray(params)[tuple(idx[..., k] for k in range(idx.shape[-1]))])


def _split(x, n, axis=0):
    return [t(p) for p in np.split(np.asarray(x), n, axis=axis)]


def make_tf_module():
    tf = types.ModuleType("tensorflow")
    keras = types.ModuleType("tensorflow.keras")
    layers = types.ModuleType("tensorflow.keras.layers")
    for cls in (Layer, Conv2D, Conv2DTranspose, BatchNormalization, LeakyReLU, MaxPool2D, Dropout):
        setattr(layers, cls.__name__, cls)
    keras.layers, keras.Model = layers, Layer
    tf.keras = keras
    tf.float32, tf.int32 = "float32", "int32"
    tf.reshape = lambda x, shape: t(np.reshape(np.asarray(x), [int(s) for s in shape]))
    tf.transpose = lambda x, perm: t(np.transpose(np.asarray(x), perm))
    tf.matmul = lambda a, b: t(np.matmul(np.asarray(a, np.float64), np.asarray(b, np.float64)))
    tf.concat = lambda xs, axis: t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.stack = lambda xs, axis=0: t(np.stack([np.asarray(x) for x in xs], axis=axis))
    tf.zeros = lambda shape: t(np.zeros([int(s) for s in shape], np.float64))
    tf.tanh = lambda x: t(np.tanh(np.asarray(x)))
    tf.greater, tf.cast, tf.split, tf.gather_nd = _greater, _cast, _split, _gather_nd
    tf.stop_gradient = lambda x: x
    tf.reduce_max = lambda x, axis: t(np.max(np.asarray(x), axis=axis))
    tf.reduce_mean = lambda x, axis: t(np.mean(np.asarray(x), axis=axis))
    tf.cond = lambda pred, a, b: a() if bool(pred) else b()
    tf.shape = lambda x: np.asarray(x).shape
    tf.range = lambda n: t(np.arange(int(n)))
    tf.meshgrid = lambda a, b, indexing="xy": [t(m) for m in np.meshgrid(np.asarray(a), np.asarray(b), indexing=indexing)]
    tf.expand_dims = lambda x, axis: t(np.expand_dims(np.asarray(x), axis))
    tf.tile = lambda x, reps: t(np.tile(np.asarray(x), [int(r) for r in reps]))
    tf.clip_by_value = lambda x, lo, hi: t(np.clip(np.asarray(x), lo, hi))
    nn = types.ModuleType("tensorflow.nn")
    nn.softmax = _softmax
    nn.relu = lambda x: t(np.maximum(np.asarray(x), 0))
    tf.nn = nn
    image = types.ModuleType("tensorflow.image")
    image.resize = _resize
    image.rgb_to_grayscale = lambda x: t(P.gray(np.asarray(x)))
    tf.image = image
    math = types.ModuleType("tensorflow.math")
    math.floor = lambda x: t(np.floor(np.asarray(x)))
    math.ceil = lambda x: t(np.ceil(np.asarray(x)))
    tf.math = math
    mods = {"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.keras.layers": layers}

    class _Any(types.ModuleType):                       # tensorflow_addons, cv2: imported, never executed on this path
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            raise RuntimeError("%s.%s is not on the inference path and has no stand-in" % (self.__name__, name))
    for name in ("tensorflow_addons", "cv2"):
        mods[name] = _Any(name)
    return mods


def import_reference(module_file):
    """Import /root/reference/<module_file> (and its `from warp import ...`) with the stand-ins in sys.modules."""
    mods = make_tf_module()
    saved = {k: sys.modules.get(k) for k in list(mods) + ["warp", "scipy.ndimage.interpolation"]}
    sys.modules.update(mods)
    if "scipy.ndimage.interpolation" not in sys.modules or sys.modules["scipy.ndimage.interpolation"] is None:
        import scipy.ndimage as ndi
        shim = types.ModuleType("scipy.ndimage.interpolation")          # removed from recent scipy; warp.py:4 imports it at top level
        shim.map_coordinates = ndi.map_coordinates
        sys.modules["scipy.ndimage.interpolation"] = shim
    sys.path.insert(0, REF)
    try:
        sys.modules.pop("warp", None)
        spec = importlib.util.spec_from_file_location("ref_" + module_file.replace(".py", ""), os.path.join(REF, module_file))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def assign_weights(gen, weights):
    """tf.train.Checkpoint(generator=gen) names variables by the attribute path (train_test_GSC.py:143-148):
    `res_stack/3/non_local/g/kernel` -> gen.res_stack[3].non_local.g.kernel.  Every variable must land on a stand-in layer
    that the forward then really uses."""
    touched = []
    for name, val in weights.items():
        obj = gen
        parts = name.split("/")
        for p in parts[:-1]:
            obj = obj[int(p)] if isinstance(obj, (list, tuple)) else getattr(obj, p)
        if not hasattr(obj, parts[-1]):
            raise RuntimeError("reference layer %s has no variable slot '%s' (%s)" % (type(obj).__name__, parts[-1], name))
        setattr(obj, parts[-1], np.asarray(val, np.float64))
        obj._used = False
        touched.append((name, obj))
    return touched


def run_reference(module_file, weights, call_args, call_kwargs=None):
    mod = import_reference(module_file)
    gen = mod.Generator()
    touched = assign_weights(gen, weights)
    with contextlib.redirect_stdout(io.StringIO()):                 # model_with_TSM.py:210-214 prints shapes
        outs = gen(*call_args, **(call_kwargs or {}))
    unused = sorted({n for n, o in touched if not o._used})
    if unused:
        raise RuntimeError("variables never used by the reference's forward: %s" % unused[:6])
    return [np.asarray(o) for o in outs]


# ------------------------------------------------------------------------------------------------ mock TensorFlow (dry run of the tf backend)
class Variable:
    """What the tf backend touches of a tf.Variable: shape, assign(), ref(), numpy(); array-like for the stand-in arithmetic."""

    def __init__(self, shape):
        self.value = np.zeros(shape, np.float64)

    @property
    def shape(self):
        return self.value.shape

    def assign(self, v):
        v = np.asarray(v)
        assert v.shape == self.value.shape, (v.shape, self.value.shape)
        self.value = v.astype(np.float64)

    def ref(self):
        return id(self)

    def numpy(self):
        return self.value

    def __array__(self, dtype=None, copy=None):
        return self.value if dtype is None else self.value.astype(dtype)


def make_mock_tf_module():
    """The stand-in of make_tf_module() with Keras' VARIABLE mechanics added — layers create their variables on the first call,
    `model.variables` walks the attribute tree (lists included), `tf.constant`, `tf.__version__` — so that run_reference_tf's driver
    logic (build by one forward, assignment through the checkpoint attribute paths, the every-variable-assigned check, the keyword
    call, the d32 spy on tf.image.resize) can be exercised WITHOUT TensorFlow (tests/test_model_py_fixture.py, BSR_MOCK_TF=1).  Same
    arithmetic as the stand-in, so the outputs must equal the committed stand-in fixtures bit for bit."""
    mods = make_tf_module()
    tf, layers = mods["tensorflow"], mods["tensorflow.keras.layers"]

    def _walk(obj, seen, out):
        if id(obj) in seen:
            return
        seen.add(id(obj))
        if isinstance(obj, Variable):
            out.append(obj)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                _walk(o, seen, out)
        elif isinstance(obj, Layer):
            for v in vars(obj).values():
                _walk(v, seen, out)

    class MockLayer(Layer):
        @property
        def variables(self):
            out = []
            _walk(self, set(), out)
            return out

    class MConv2D(MockLayer, Conv2D):
        def call(self, x):
            if self.kernel is None:
                self.kernel, self.bias = Variable(self.ksize + (x.shape[-1], self.filters)), Variable((self.filters,))
            return t(P.conv2d_same(np.asarray(x), np.asarray(self.kernel), np.asarray(self.bias), self.strides[0]))

    class MConv2DTranspose(MockLayer, Conv2DTranspose):
        def call(self, x):
            if self.kernel is None:
                self.kernel, self.bias = Variable((3, 3, self.filters, x.shape[-1])), Variable((self.filters,))
            return t(P.conv2d_transpose_same(np.asarray(x), np.asarray(self.kernel), np.asarray(self.bias)))

    class MBatchNormalization(MockLayer, BatchNormalization):
        def call(self, x, training=None):
            if self.gamma is None:
                c = (x.shape[-1],)
                self.gamma, self.beta, self.moving_mean, self.moving_variance = Variable(c), Variable(c), Variable(c), Variable(c)
                self.gamma.assign(np.ones(c))
                self.moving_variance.assign(np.ones(c))
            assert training is False
            return t(P.batchnorm(np.asarray(x), *(np.asarray(v) for v in (self.gamma, self.beta, self.moving_mean, self.moving_variance))))

    class MLeakyReLU(MockLayer, LeakyReLU):
        pass

    for name, cls in (("Layer", MockLayer), ("Conv2D", MConv2D), ("Conv2DTranspose", MConv2DTranspose), ("BatchNormalization", MBatchNormalization),
                      ("LeakyReLU", MLeakyReLU)):
        setattr(layers, name, cls)
    tf.keras.Model = MockLayer
    tf.constant = lambda a, dtype=None: t(np.asarray(a, np.float64))
    tf.__version__ = "mock"
    return mods


# ------------------------------------------------------------------------------------------------ real-TensorFlow backend
def run_reference_tf(module_file, weights, call_args):
    """The reference's module UNMODIFIED over real TensorFlow.  Returns ([gs, con_rgb, mask22, dif], {"d32": ...}, "tf-<version>").
    BSR_MOCK_TF=1 runs the same driver over make_mock_tf_module() (a dry run of THIS function's logic, not a pin)."""
    mock = os.environ.get("BSR_MOCK_TF") == "1"
    if mock:
        saved = {k: sys.modules.get(k) for k in ("tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow_addons", "cv2")}
        sys.modules.update(make_mock_tf_module())
    try:
        return _run_reference_tf(module_file, weights, call_args)
    finally:
        if mock:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v


def _run_reference_tf(module_file, weights, call_args):
    try:
        import tensorflow as tf
    except ImportError as e:                                            # the build container: no TensorFlow
        raise SystemExit("--backend tf needs TensorFlow (the reference pins 2.3.0, README.md:11): %s" % e)
    for name in ("tensorflow_addons", "cv2"):                           # imported at module top (model.py:2, warp.py:2), never executed here
        try:
            importlib.import_module(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
    if "scipy.ndimage.interpolation" not in sys.modules:
        try:
            importlib.import_module("scipy.ndimage.interpolation")
        except ImportError:
            import scipy.ndimage as ndi
            shim = types.ModuleType("scipy.ndimage.interpolation")
            shim.map_coordinates = ndi.map_coordinates
            sys.modules["scipy.ndimage.interpolation"] = shim
    sys.path.insert(0, REF)
    try:
        sys.modules.pop("warp", None)
        spec = importlib.util.spec_from_file_location("reftf_" + module_file.replace(".py", ""), os.path.join(REF, module_file))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REF)
    gen = mod.Generator()
    tens = [tf.constant(np.asarray(a, np.float32)) if isinstance(a, np.ndarray) else a for a in call_args]
    # keyword form, exactly as the reference's call sites write it (train_test_GSC.py:871; train_with_TSM.py:676)
    if len(tens) == 5:                                                  # model.py: (inputs, uv, reg, chuck, training)
        args, kwargs = tens[:3], {"chuck": tens[3], "training": tens[4]}
    else:                                                               # model_with_TSM.py: (inputs, uv, reg, frame, share, chuck, training)
        args, kwargs = tens[:3], {"frame": tens[3], "share": tens[4], "chuck": tens[5], "training": tens[6]}
    with contextlib.redirect_stdout(io.StringIO()):
        gen(*args, **kwargs)                                            # Keras creates the variables on the first call
    assigned = set()
    for name, val in weights.items():
        obj = gen
        parts = name.split("/")
        for p in parts[:-1]:
            obj = obj[int(p)] if p.isdigit() else getattr(obj, p)      # res_stack/<i>/... : Keras tracks the Python list as a ListWrapper
        var = getattr(obj, parts[-1])                                   # kernel / bias / gamma / beta / moving_mean / moving_variance
        if tuple(var.shape) != tuple(np.shape(val)):
            raise RuntimeError("%s: reference variable %s vs init_weights %s" % (name, tuple(var.shape), np.shape(val)))
        var.assign(np.asarray(val, np.float32))
        assigned.add(var.ref() if hasattr(var, "ref") else id(var))
    have = {(v.ref() if hasattr(v, "ref") else id(v)) for v in gen.variables}
    if have != assigned:
        raise RuntimeError("%d generator variables, %d assigned: the checkpoint attribute paths do not cover the model" % (len(have), len(assigned)))
    probes = {}
    orig = tf.image.resize

    def spy(x, size, *a, **k):
        y = orig(x, size, *a, **k)
        if x.shape[-1] == 1:                                            # the only 1-channel resize is d32 (model.py:256)
            probes["d32"] = y.numpy()
        return y
    tf.image.resize = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            outs = gen(*args, **kwargs)
    finally:
        tf.image.resize = orig
    return [np.asarray(o.numpy(), np.float64) for o in outs], {k: np.asarray(v) for k, v in probes.items()}, "tf-" + tf.__version__


BACKEND = "standin"


def record_probes(module_file, weights, call_args):
    """-> (outputs, {"d32"}, backend label): over real TensorFlow (--backend tf) or over the stand-in (default)."""
    if BACKEND == "tf":
        return run_reference_tf(module_file, weights, [np.asarray(a) if isinstance(a, np.ndarray) else a for a in call_args])
    outs, probes = record_probes_standin(module_file, weights, call_args)
    return outs, probes, "standin-np_loops"


def record_probes_standin(module_file, weights, call_args):
    """Second run that also captures the 32x32 threshold input: tf.image.resize of a 1-channel map is only called for d32."""
    probes = {}
    orig = P.resize_bilinear

    def spy(x, oh, ow):
        y = orig(x, oh, ow)
        if x.shape[-1] == 1:
            probes["d32"] = np.asarray(y)
        return y
    P.resize_bilinear = spy
    try:
        outs = run_reference(module_file, weights, call_args)
    finally:
        P.resize_bilinear = orig
    return outs, probes


def synthetic_inputs(seed, B, S):
    """Inputs stored as float16-exact float32 values so the fixture stays small and regenerable bit for bit."""
    rng = np.random.default_rng(seed)
    inp = rng.random((B, S, S, 3)).astype(np.float16).astype(np.float32)
    uv = rng.random((B, S, S, 3)).astype(np.float16).astype(np.float32)
    uv[:, :, : S // 8] = 0                       # real uv maps are zero outside the landmark hull
    return inp, uv


def make_gsc(S, B, seed, path):
    w = init_weights(1)
    inp, uv = synthetic_inputs(seed, B, S)
    (gs, con_rgb, mask22, dif), pr, backend = record_probes("model.py", w, (t(inp, np.float64), t(uv, np.float64), None, 1, False))
    d32 = pr["d32"]
    np.savez_compressed(path, backend=backend, weights_seed=1, input_seed=seed, inputs=inp.astype(np.float16), uv=uv.astype(np.float16),
                        gs=gs.astype(np.float32), con_rgb=con_rgb.astype(np.float32), mask22=mask22.astype(np.float32), dif=dif.astype(np.float32),
                        d32=d32.astype(np.float32), bmask=(d32.astype(np.float32) > np.float32(0.1)).astype(np.float32),
                        min_abs_d32_minus_thr=np.float32(np.abs(d32 - 0.1).min()))
    print(path, "gs", gs.shape, "min|d32-0.1| = %.3e" % np.abs(d32 - 0.1).min(), "bmask mean %.3f" % (d32 > 0.1).mean())


def make_tsm(S, frame, seed, path):
    w = init_weights(1, variant="tsm")
    inp, uv = synthetic_inputs(seed, frame, S)
    rng = np.random.default_rng(seed + 100)
    import torch
    coarse = torch.from_numpy((rng.random((frame, 6, 5, 5)) - 0.5) * 0.3)
    reg = torch.nn.functional.interpolate(coarse, size=(S, S), mode="bicubic", align_corners=True).permute(0, 2, 3, 1).numpy()
    reg[..., 2] = 0
    reg[..., 5] = 0
    reg = reg.astype(np.float16).astype(np.float32)
    args = (t(inp, np.float64), t(uv, np.float64), t(reg, np.float64), frame, True, 1, False)
    (gs, con_rgb, mask22, dif), pr, backend = record_probes("model_with_TSM.py", w, args)
    d32 = pr["d32"]
    np.savez_compressed(path, backend=backend, weights_seed=1, input_seed=seed, frame=frame, inputs=inp.astype(np.float16), uv=uv.astype(np.float16),
                        reg=reg.astype(np.float16), gs=gs.astype(np.float32), con_rgb=con_rgb.astype(np.float32),
                        mask22=mask22.astype(np.float32), dif=dif.astype(np.float32), d32=d32.astype(np.float32),
                        bmask=(d32.astype(np.float32) > np.float32(0.1)).astype(np.float32),
                        min_abs_d32_minus_thr=np.float32(np.abs(d32 - 0.1).min()))
    print(path, "gs", gs.shape, "min|d32-0.1| = %.3e" % np.abs(d32 - 0.1).min(), "bmask mean %.3f" % (d32 > 0.1).mean())


if __name__ == "__main__":
    gold = os.path.join(ROOT, "tests", "golden")
    argv = sys.argv[1:]
    if "--backend" in argv:
        i = argv.index("--backend")
        BACKEND = argv[i + 1]
        del argv[i:i + 2]
        if BACKEND not in ("standin", "tf"):
            raise SystemExit("--backend must be standin or tf")
    which = argv or ["gsc64", "tsm64", "gsc256", "tsm256"]
    if "gsc64" in which:
        make_gsc(64, 2, 11, os.path.join(gold, "model_py_gsc_64.npz"))
    if "tsm64" in which:
        make_tsm(64, 2, 12, os.path.join(gold, "model_py_tsm_64.npz"))
    if "gsc256" in which:
        make_gsc(256, 1, 13, os.path.join(gold, "model_py_gsc_256.npz"))
    if "tsm256" in which:
        make_tsm(256, 2, 15, os.path.join(gold, "model_py_tsm_256.npz"))
