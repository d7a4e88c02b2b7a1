"""The C-ABI library loads on a CPU-only box and exports every symbol include/bsr_hip.h declares.
No compute is launched here (no GPU)."""
import ctypes
import os
import re

import pytest

from blindshadowremoval_amd import _lib
from blindshadowremoval_amd.build import LIB_PATH, build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build_library()            # no-op when the in-tree .so is fresh
    assert os.path.isfile(LIB_PATH)
    return _lib.load()


def test_header_symbols_are_exported(lib):
    with open(os.path.join(ROOT, "include", "bsr_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(bsr_[a-z_]+)\s*\(", text))
    assert declared == set(_lib.EXPORTS), "binding and header disagree: %s" % (declared ^ set(_lib.EXPORTS))
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.bsr_abi_version() == _lib.ABI_VERSION


def test_workspace_bytes_is_pure_host_arithmetic(lib):
    one = lib.bsr_workspace_bytes(1, 256, 256)
    assert 80 * 2 ** 20 < one < 120 * 2 ** 20          # ~24 M floats of activations per image
    assert lib.bsr_workspace_bytes(32, 256, 256) <= 32 * one
    assert lib.bsr_workspace_bytes(0, 256, 256) == 0


def test_create_rejects_bad_blobs_before_touching_the_gpu(lib):
    h = ctypes.c_void_p()
    bad = (ctypes.c_char * 64)()
    rc = lib.bsr_create(ctypes.byref(h), 0, ctypes.cast(bad, ctypes.c_void_p), 64, 0)
    assert rc == 2 and b"magic" in lib.bsr_last_error()
    rc = lib.bsr_create(ctypes.byref(h), 0, ctypes.cast(bad, ctypes.c_void_p), 4, 0)
    assert rc == 2
    rc = lib.bsr_create(ctypes.byref(h), 0, ctypes.cast(bad, ctypes.c_void_p), 64, 7)
    assert rc == 1 and b"F32" in lib.bsr_last_error()
    rc = lib.bsr_create(ctypes.byref(h), 0, None, 64, 0)
    assert rc == 1
    with pytest.raises(RuntimeError, match="code 1"):
        _lib.check(rc, "bsr_create")
    assert lib.bsr_forward(None, None, None, 1, 256, 256, None, None, None, None, None) == 1
    assert lib.bsr_debug_attention(None, None, 1, 1024, None) == 1


def test_header_is_plain_c():
    """include/bsr_hip.h must be consumable by a C compiler (the boundary is a C ABI, no C++ / torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "bsr_hip.h")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
