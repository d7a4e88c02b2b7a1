"""Build libbsr_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libbsr_hip.so")
SOURCES = ["csrc/bsr_api.hip"]
# the loaders' host-side helper (plain C, gcc, no GPU code: PNG scanline reconstruction for the worker processes) — its own small
# library so that a worker can load it without bringing the HIP runtime in
HOST_LIB_PATH = os.path.join(PKG_DIR, "libbsr_host.so")
HOST_SOURCES = ["hostsrc/png_unfilter.c", "hostsrc/inflate.c"]


def _deps():
    """Everything the library is compiled from: every kernel header under csrc/ plus the public C header."""
    import glob
    return (sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.h"))) + sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.hip")))
            + [os.path.join(PKG_DIR, "..", "include", "bsr_hip.h")])


def source_sha16() -> str:
    """Hash of the kernel sources: profiles/*_pmc_traffic.json records it so bench.py can tell a stale traffic figure."""
    import hashlib
    h = hashlib.sha256()
    for f in _deps():
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    if extra_flags():
        h.update(b"flags\0" + " ".join(extra_flags()).encode())
    return h.hexdigest()[:16]


def extra_flags():
    """Extra hipcc flags of an EXPERIMENT build (env BSR_EXTRA_FLAGS, e.g. "-DBSR_NL_DIAG=4"): part of the source hash, so a library
    built with them only loads while the variable still says so (scratch/ab_build.sh); empty for every product build."""
    return os.environ.get("BSR_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def library_sha16(path: str = None) -> str:
    """The source hash compiled into a built library, read from the FILE (the bytes behind its "BSR_SRC_SHA=" tag) — nothing is
    loaded into the process (a dlopen here, before torch is imported, would bring a second HIP runtime in); '' if there is none."""
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(b"BSR_SRC_SHA=")
    if i < 0:
        return ""
    tail = blob[i + 12:i + 12 + 32].split(b"\0", 1)[0]
    return tail.decode(errors="replace")


def is_stale() -> bool:
    """True when the library is missing or was compiled from other sources than the tree holds now.  The embedded hash decides
    (a checkout or a copy resets mtimes); the mtime test only saves loading the library when it is obviously fresh."""
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    if any(os.path.getmtime(f) > t for f in _deps()):
        return True
    return library_sha16() != source_sha16()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into blindshadowremoval_amd/libbsr_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    # several ranks of one node may get here together (bench.py --gpus N, torchrun): one of them compiles, the others wait on
    # the lock and find the library fresh
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB_PATH
            tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
            # the hash of everything the library is compiled from goes INTO the binary (bsr_source_sha()): _lib.load() refuses a
            # library whose hash is not the tree's, so a stale .so can neither pass the tests nor produce a bench line
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
                   '-DBSR_SRC_SHA="%s"' % source_sha16()] + extra_flags() + ["-o", tmp] + [os.path.join(PKG_DIR, s) for s in SOURCES]
            res = subprocess.run(cmd, cwd=PKG_DIR, capture_output=True, text=True)
            if verbose or res.returncode != 0:
                print(" ".join(cmd))
                print(res.stdout + res.stderr)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed building libbsr_hip.so:\n" + res.stderr[-4000:])
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def host_source_sha16() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in HOST_SOURCES:
        with open(os.path.join(PKG_DIR, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def host_library_sha16(path: str = None) -> str:
    try:
        with open(path or HOST_LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(b"BSR_HOST_SHA=")
    return blob[i + 13:i + 13 + 32].split(b"\0", 1)[0].decode(errors="replace") if i >= 0 else ""


def build_host_library(force: bool = False) -> str:
    """gcc -O3 -> blindshadowremoval_amd/libbsr_host.so (in-tree, bound to its source by an embedded hash like libbsr_hip.so)."""
    if not force and host_library_sha16() == host_source_sha16():
        return HOST_LIB_PATH
    import fcntl
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler: libbsr_host.so cannot be built")
    with open(HOST_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and host_library_sha16() == host_source_sha16():
                return HOST_LIB_PATH
            tmp = "%s.%d.tmp" % (HOST_LIB_PATH, os.getpid())
            cmd = [cc, "-O3", "-shared", "-fPIC", '-DBSR_HOST_SHA="%s"' % host_source_sha16(), "-o", tmp] + [os.path.join(PKG_DIR, s) for s in HOST_SOURCES]
            res = subprocess.run(cmd, cwd=PKG_DIR, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("building libbsr_host.so failed:\n" + res.stderr[-4000:])
            os.replace(tmp, HOST_LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_LIB_PATH
