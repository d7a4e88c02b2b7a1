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
            self._scratch = torch.zeros((max(need, 4096) + 7) // 8, dtype=torch.int64, device=strips.device)      # zero ONCE: every call leaves it zero (ABI 7)
        with torch.cuda.device(self.device):
            rc = lib.bsr_png_encode(self.device, ctypes.c_void_p(strips.data_ptr()), b, h, w, ctypes.c_void_p(out.data_ptr()), n,
                                    ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "bsr_png_encode")
        return out

    def encode_figs(self, figs, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The PNG files of the strips `Logging.strips_on_device(figs)` would make, without making them (bsr_png_encode_figs): ``figs`` is
        the list of figures of one strip, each a float32 CUDA tensor [B,H,Wf,1|3] — any view whose pixels are a fixed number of floats
        apart, e.g. a channel slice of the packed rows — or a tuple (tensor, multiplier | None, scale): a one-channel multiplier image of
        the same [B,H,Wf,1] extent and a scalar, applied in that order (test_step_FFHQ's `mask_pred * face * 2`).  Byte for byte the
        files of ``encode(strips_on_device(...))``.  Returns None when a figure's layout is not one the kernel addresses (the caller then
        takes the strip path)."""
        spec = []
        for f in figs:
            t, m, sc = (f if isinstance(f, tuple) else (f, None, 1.0))
            spec.append((t, m, float(sc)))
        t0 = spec[0][0]
        if t0.dim() != 4:
            return None
        b, h, wf = int(t0.shape[0]), int(t0.shape[1]), int(t0.shape[2])

        def pix_stride(t, ch):
            """floats between neighbouring pixels when [B,H,Wf] are laid out densely in that stride, else None"""
            if t.dtype != torch.float32 or not t.is_cuda or t.device.index != self.device or tuple(t.shape[:3]) != (b, h, wf) or t.shape[3] != ch:
                return None
            ps = t.stride(2)
            if ps < ch or t.stride(3) != 1 or t.stride(1) != wf * ps or t.stride(0) != h * wf * ps:
                return None
            return int(ps)
        n = len(spec)
        if n < 1 or n > 8:
            return None
        ptrs, muls, scales, chans, pss, mss = [], [], [], [], [], []
        for t, m, sc in spec:
            ch = int(t.shape[3]) if t.dim() == 4 else 0
            ps = pix_stride(t, ch) if ch in (1, 3) else None
            ms = pix_stride(m, 1) if m is not None and m.dim() == 4 else (None if m is not None else 1)
            if ps is None or ms is None:
                return None
            ptrs.append(t.data_ptr()); muls.append(m.data_ptr() if m is not None else 0); scales.append(sc); chans.append(ch); pss.append(ps); mss.append(ms)
        nbytes = file_bytes(h, n * wf)
        if out is None:
            out = torch.empty((b, nbytes), dtype=torch.uint8, device=t0.device)
        lib = _lib.load()
        need = int(lib.bsr_png_scratch_bytes(b))
        if self._scratch is None or self._scratch.numel() * 8 < need:
            self._scratch = torch.zeros((max(need, 4096) + 7) // 8, dtype=torch.int64, device=t0.device)           # zero ONCE: every call leaves it zero (ABI 7)
        P = (ctypes.c_void_p * n)
        with torch.cuda.device(self.device):
            rc = lib.bsr_png_encode_figs(self.device, n, P(*ptrs), P(*muls), (ctypes.c_float * n)(*scales), (ctypes.c_int * n)(*chans), (ctypes.c_int * n)(*pss),
                                         (ctypes.c_int * n)(*mss), b, h, wf, ctypes.c_void_p(out.data_ptr()), nbytes, ctypes.c_void_p(self._scratch.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "bsr_png_encode_figs")
        return out
