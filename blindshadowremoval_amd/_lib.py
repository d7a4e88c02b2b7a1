"""ctypes binding of libbsr_hip.so (C ABI: include/bsr_hip.h).  There is NO CPU fallback: if the
library is missing the import of the HIP path fails loudly."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

from . import build as _build
from .build import LIB_PATH

ABI_VERSION = 8
NUM_CLASSES = 7
CLASS_NAMES = ("conv3x3", "convT3x3", "conv1x1", "attention", "conv7", "glue", "convT3x3_ni2")

_lib: Optional[ctypes.CDLL] = None

# every symbol include/bsr_hip.h declares
EXPORTS = ("bsr_create", "bsr_forward", "bsr_forward_tsm", "bsr_workspace_bytes", "bsr_reserve", "bsr_probe", "bsr_set_timing",
           "bsr_get_timing", "bsr_timing_launches", "bsr_timing_entry", "bsr_handle_workspace_bytes", "bsr_debug_attention", "bsr_debug_attention_dtype", "bsr_debug_attention_qw", "bsr_debug_split_qkv", "bsr_clock_trace", "bsr_debug_attention_split", "bsr_destroy", "bsr_last_error", "bsr_abi_version", "bsr_check_range", "bsr_prep_rows", "bsr_png_unfilter", "bsr_forward_packed", "bsr_source_sha", "bsr_peek_range", "bsr_png_file_bytes", "bsr_png_scratch_bytes", "bsr_png_encode", "bsr_png_encode_figs", "bsr_ucb_post_scratch_bytes", "bsr_ucb_post")


def load() -> ctypes.CDLL:
    """Load the in-tree library (after torch, so both share one HIP runtime) and declare signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError("libbsr_hip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "— the HIP path has no fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; our DT_NEEDED then resolves to the same runtime)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    # The binary is bound to its sources: build.py compiles the hash of csrc/* + include/bsr_hip.h into it.  A library built from
    # other sources than this tree holds (the .so is git-ignored and travels as a built artefact) is refused — tests and bench lines
    # can then only ever describe the kernels that are in the tree.
    try:
        lib.bsr_source_sha.restype = ctypes.c_char_p
        built = (lib.bsr_source_sha() or b"").decode()
    except AttributeError:
        built = "<none: built before the hash was embedded>"
    want = _build.source_sha16()
    if built != want:
        raise RuntimeError("libbsr_hip.so is STALE: it was compiled from kernel sources %s, the tree holds %s — rebuild "
                           "(`python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback" % (built, want))
    c_f, c_i, c_v, c_sz = ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.bsr_abi_version.restype = c_i
    lib.bsr_last_error.restype = ctypes.c_char_p
    lib.bsr_create.argtypes = [ctypes.POINTER(c_v), c_i, c_v, c_sz, c_i]
    lib.bsr_create.restype = c_i
    lib.bsr_forward.argtypes = [c_v, c_v, c_v, c_i, c_i, c_i, c_v, c_v, c_v, c_v, c_v]
    lib.bsr_forward.restype = c_i
    lib.bsr_forward_packed.argtypes = [c_v, c_v, c_v, c_i, c_i, c_i, c_v, c_v, c_v, c_v]
    lib.bsr_forward_packed.restype = c_i
    lib.bsr_forward_tsm.argtypes = [c_v, c_v, c_v, c_v, c_i, c_i, c_i, c_i, c_i, c_v, c_v, c_v, c_v, c_v]
    lib.bsr_forward_tsm.restype = c_i
    lib.bsr_workspace_bytes.argtypes = [c_i, c_i, c_i]
    lib.bsr_workspace_bytes.restype = c_sz
    lib.bsr_reserve.argtypes = [c_v, c_i, c_i, c_i]
    lib.bsr_reserve.restype = c_i
    lib.bsr_probe.argtypes = [c_v, ctypes.c_char_p, c_v, c_sz, ctypes.POINTER(c_i * 4), c_v]
    lib.bsr_probe.restype = c_i
    lib.bsr_set_timing.argtypes = [c_v, c_i]
    lib.bsr_set_timing.restype = c_i
    lib.bsr_get_timing.argtypes = [c_v, ctypes.POINTER(ctypes.c_float * NUM_CLASSES), ctypes.POINTER(c_i * NUM_CLASSES)]
    lib.bsr_get_timing.restype = c_i
    lib.bsr_timing_launches.argtypes = [c_v]
    lib.bsr_timing_launches.restype = c_i
    lib.bsr_timing_entry.argtypes = [c_v, c_i, ctypes.c_char_p, c_sz, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(c_i)]
    lib.bsr_timing_entry.restype = c_i
    lib.bsr_handle_workspace_bytes.argtypes = [c_v, c_i, c_i, c_i]
    lib.bsr_handle_workspace_bytes.restype = c_sz
    lib.bsr_debug_attention.argtypes = [c_v, c_v, c_i, c_i, c_v]
    lib.bsr_debug_attention.restype = c_i
    lib.bsr_debug_attention_dtype.argtypes = [c_v, c_v, c_i, c_i, c_i, c_v]
    lib.bsr_debug_attention_dtype.restype = c_i
    lib.bsr_debug_attention_qw.argtypes = [c_v, c_v, c_i, c_i, c_i, c_v]
    lib.bsr_clock_trace.argtypes = [c_i, c_v, c_i, c_i, c_v, c_v, c_v]
    lib.bsr_clock_trace.restype = c_i
    lib.bsr_debug_split_qkv.argtypes = [c_v, c_v, c_i, c_i, c_v]
    lib.bsr_debug_split_qkv.restype = c_i
    lib.bsr_debug_attention_split.argtypes = [c_v, c_v, c_i, c_i, c_i, c_v]
    lib.bsr_debug_attention_split.restype = c_i
    lib.bsr_debug_attention_qw.restype = c_i
    lib.bsr_prep_rows.argtypes = [c_i, c_v, c_sz, c_sz, c_sz, c_i, c_i, c_v, c_v, c_v]
    lib.bsr_prep_rows.restype = c_i
    lib.bsr_png_unfilter.argtypes = [c_i, c_v, c_sz, c_sz, c_i, c_v]
    lib.bsr_png_unfilter.restype = c_i
    lib.bsr_check_range.argtypes = [c_v, c_v]
    lib.bsr_check_range.restype = c_i
    lib.bsr_png_file_bytes.argtypes = [c_i, c_i]
    lib.bsr_png_file_bytes.restype = c_sz
    lib.bsr_png_scratch_bytes.argtypes = [c_i]
    lib.bsr_png_scratch_bytes.restype = c_sz
    lib.bsr_png_encode.argtypes = [c_i, c_v, c_i, c_i, c_i, c_v, c_sz, c_v, c_v]
    lib.bsr_png_encode.restype = c_i
    lib.bsr_png_encode_figs.argtypes = [c_i, c_i, c_v, c_v, c_v, c_v, c_v, c_v, c_i, c_i, c_i, c_v, c_sz, c_v, c_v]
    lib.bsr_png_encode_figs.restype = c_i
    lib.bsr_ucb_post_scratch_bytes.argtypes = [c_i, c_i]
    lib.bsr_ucb_post_scratch_bytes.restype = c_sz
    lib.bsr_ucb_post.argtypes = [c_i, c_v, c_v, c_v, c_i, c_i, c_v, c_v, c_v, c_v, c_v, c_v]
    lib.bsr_ucb_post.restype = c_i
    lib.bsr_peek_range.argtypes = [c_v]
    lib.bsr_peek_range.restype = c_i
    lib.bsr_destroy.argtypes = [c_v]
    lib.bsr_destroy.restype = None
    if lib.bsr_abi_version() != ABI_VERSION:
        raise RuntimeError("libbsr_hip.so ABI %d != binding ABI %d: rebuild" % (lib.bsr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def source_sha() -> str:
    """The source hash the LOADED library carries (== build.source_sha16(), or load() would have refused it)."""
    return (load().bsr_source_sha() or b"").decode()


class RangeError(RuntimeError):
    """BSR_ERR_RANGE: an activation did not fit fp16 in a forward of a 16-bit-mode handle (include/bsr_hip.h, bsr_check_range)."""


ERR_RANGE = 5


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().bsr_last_error()
        raise (RangeError if rc == ERR_RANGE else RuntimeError)("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
