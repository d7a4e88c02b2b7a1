"""PNG files of the loops' figure strips built on the device (csrc/png_kernels.h, bsr_png_encode): the `cv2.imwrite` of
`Logging.save_img` (/root/reference/utils.py:196-204) up to the write() itself.  uint8 [B,H,W,3] strips on the GPU -> [B, file_bytes]
uint8 on the GPU, each row one complete PNG file (stored deflate, checksums computed on the device); the loops copy that to pinned
memory with the batch's other outputs and the host only writes the bytes.  No CPU fallback: the library must be loaded."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib


def file_bytes(h: int, w: int) -> int:
    n = int(_lib.load().bsr_png_file_bytes(int(h), int(w)))
    if n == 0:
        raise ValueError("no PNG geometry for %dx%d (W <= 5461, H <= 65535)" % (h, w))
    return n


class StripEncoder:
    """Reusable encoder for strips of one shape on one device (keeps its checksum scratch and an output buffer pool of one)."""

    def __init__(self, device: int):
        self.device = int(device)
        self._scratch: Optional[torch.Tensor] = None

    def encode(self, strips: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """strips: uint8 [B,H,W,3] on this encoder's GPU (contiguous).  Returns uint8 [B, file_bytes(H, W)] on the GPU; asynchronous on
        the current stream.  ``out``: an optional preallocated destination of that shape."""
        if strips.dtype != torch.uint8 or strips.dim() != 4 or strips.shape[3] != 3:
            raise TypeError("StripEncoder.encode takes a uint8 [B,H,W,3] tensor, got %s %s" % (strips.dtype, tuple(strips.shape)))
        if not strips.is_cuda or strips.device.index != self.device:
            raise ValueError("strips must live on cuda:%d" % self.device)
        strips = strips.contiguous()
        b, h, w, _ = strips.shape
        n = file_bytes(h, w)
        if out is None:
            out = torch.empty((b, n), dtype=torch.uint8, device=strips.device)
        elif out.shape != (b, n) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != strips.device:
            raise ValueError("out must be a contiguous uint8 [%d, %d] tensor on %s" % (b, n, strips.device))
        lib = _lib.load()
        need = int(lib.bsr_png_scratch_bytes(b))
        if self._scratch is None or self._scratch.numel() * 8 < need:
            self._scratch = torch.empty((max(need, 4096) + 7) // 8, dtype=torch.int64, device=strips.device)
        with torch.cuda.device(self.device):
            rc = lib.bsr_png_encode(self.device, ctypes.c_void_p(strips.data_ptr()), b, h, w, ctypes.c_void_p(out.data_ptr()), n,
                                    ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "bsr_png_encode")
        return out
