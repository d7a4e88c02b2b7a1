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


def test_a_stale_library_is_refused(lib, tmp_path):
    """The binary is bound to its sources: build.py compiles source_sha16() into the library, _lib.load() compares it with the tree's.
    A deliberately stale .so — the real library with the embedded hash overwritten, as if it had been built from other kernel sources
    and shipped as an artefact — must raise, not load."""
    import subprocess
    import sys
    from blindshadowremoval_amd.build import library_sha16, source_sha16
    sha = source_sha16()
    assert _lib.source_sha() == sha == library_sha16() and len(sha) == 16          # the fresh build carries the tree's hash
    blob = open(LIB_PATH, "rb").read()
    assert blob.count(sha.encode()) >= 1
    stale = tmp_path / "libbsr_hip.so"
    stale.write_bytes(blob.replace(sha.encode(), b"0123456789abcdef"))
    assert library_sha16(str(stale)) == "0123456789abcdef"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from blindshadowremoval_amd import build, _lib\n"
            "build.LIB_PATH = _lib.LIB_PATH = %r\n"
            "try:\n"
            "    _lib.load()\n"
            "except RuntimeError as e:\n"
            "    print('REFUSED', e)\n"
            "else:\n"
            "    print('LOADED')\n" % (ROOT, str(stale)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout and "STALE" in r.stdout and "0123456789abcdef" in r.stdout, r.stdout + r.stderr
    # and is_stale() sees it too (so build_library() would recompile), whatever the file times say
    code2 = ("import sys; sys.path.insert(0, %r)\n"
             "from blindshadowremoval_amd import build\n"
             "build.LIB_PATH = %r\n"
             "print('STALE' if build.is_stale() else 'FRESH')\n" % (ROOT, str(stale)))
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300)
    assert r.stdout.strip() == "STALE", r.stdout + r.stderr


def test_host_helper_library_exports_its_header():
    """include/bsr_host.h (the loaders' plain-C helper, libbsr_host.so): consumable by a C compiler, every declared symbol exported."""
    import ctypes
    import re
    import shutil
    import subprocess
    from blindshadowremoval_amd import build
    hdr = os.path.join(ROOT, "include", "bsr_host.h")
    gcc = shutil.which("gcc")
    if gcc:
        res = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
    names = re.findall(r"\b(bsr_[a-z0-9_]+)\s*\(", open(hdr).read())
    assert set(names) == {"bsr_png_unfilter", "bsr_inflate_zlib", "bsr_host_source_sha"}
    lib = ctypes.CDLL(build.build_host_library())
    for n in names:
        assert hasattr(lib, n), n
    lib.bsr_host_source_sha.restype = ctypes.c_char_p
    assert lib.bsr_host_source_sha().decode() == build.host_source_sha16()
