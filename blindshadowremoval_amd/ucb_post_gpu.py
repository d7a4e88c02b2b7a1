"""`FSRNet.test_step`'s per-item post-processing (/root/reference/train_test_GSC.py:424-748) for a whole batch ON THE DEVICE:
binding of bsr_ucb_post (csrc/ucb_kernels.h).  blindshadowremoval_amd/ucb_post.py is the host statement of the same steps — every
threshold / mask / component decision of the two is bit-identical (tests/test_ucb_post_gpu.py); this module has no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .ucb_post import MASK_DIRS

from .prep import MASK_ORDER, read_masks_u8      # noqa: F401  (re-exported: the mask order of bsr_ucb_post and the reader the loaders use)

assert MASK_ORDER == tuple(MASK_DIRS)
STATUS_TEXT = {1: "a segmentation mask the reference takes a bounding box of (nose / mouth / forehead / face) is empty after the resize",
               2: "the crop box is larger than the image or empty"}


class UcbPostDevice:
    """Reusable runner for one device: keeps its scratch and output buffers."""

    def __init__(self, device: int):
        self.device = int(device)
        self._scratch: Optional[torch.Tensor] = None

    def run(self, rows10: torch.Tensor, masks: torch.Tensor, boxes: torch.Tensor, want_figs: bool = False):
        """rows10: [B,S,S,10] float32 (input 3 | gt 3 | con_rgb 3 | dif 1), masks: [B,7,S,S] uint8, boxes: [B,4] float32 — all on this
        device.  -> (losses [B,2] float32 = ssim | psnr, strips [B,S,7S,3] uint8, figs [B,7,S,S,3] float32 | None, status [B] int32), on
        the device, asynchronous on the current stream.  Check `status` (raise_for_status) once it is on the host."""
        dev = torch.device("cuda", self.device)
        for name, t, dt, nd in (("rows10", rows10, torch.float32, 4), ("masks", masks, torch.uint8, 4), ("boxes", boxes, torch.float32, 2)):
            if not isinstance(t, torch.Tensor) or t.dtype != dt or t.dim() != nd or t.device != dev:
                raise TypeError("%s must be a %s tensor with %d dims on %s" % (name, dt, nd, dev))
        rows10, masks, boxes = rows10.contiguous(), masks.contiguous(), boxes.contiguous()
        b, s = rows10.shape[0], rows10.shape[1]
        if rows10.shape != (b, s, s, 10) or masks.shape != (b, 7, s, s) or boxes.shape != (b, 4):
            raise ValueError("shapes: rows10 [B,S,S,10], masks [B,7,S,S], boxes [B,4]; got %s %s %s" % (tuple(rows10.shape), tuple(masks.shape), tuple(boxes.shape)))
        lib = _lib.load()
        need = int(lib.bsr_ucb_post_scratch_bytes(b, s))
        if need == 0:
            raise ValueError("bsr_ucb_post supports S in {32, 64, 128, 256} (reference: 256), got %d" % s)
        if self._scratch is None or self._scratch.numel() < need + 256:
            self._scratch = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        base = self._scratch.data_ptr()
        base += (-base) % 256
        losses = torch.empty((b, 2), dtype=torch.float32, device=dev)
        strips = torch.empty((b, s, 7 * s, 3), dtype=torch.uint8, device=dev)
        figs = torch.empty((b, 7, s, s, 3), dtype=torch.float32, device=dev) if want_figs else None
        status = torch.empty((b,), dtype=torch.int32, device=dev)
        with torch.cuda.device(self.device):
            rc = lib.bsr_ucb_post(self.device, ctypes.c_void_p(rows10.data_ptr()), ctypes.c_void_p(masks.data_ptr()), ctypes.c_void_p(boxes.data_ptr()),
                                  b, s, ctypes.c_void_p(losses.data_ptr()), ctypes.c_void_p(strips.data_ptr()),
                                  ctypes.c_void_p(figs.data_ptr()) if figs is not None else None, ctypes.c_void_p(status.data_ptr()),
                                  ctypes.c_void_p(base), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "bsr_ucb_post")
        return losses, strips, figs, status


def raise_for_status(status: Sequence[int], names: Optional[Sequence[str]] = None) -> None:
    """The reference raises (numpy's min of an empty array) where a mask it needs is empty: so does the device path, by item."""
    for j, st in enumerate(status):
        if int(st) != 0:
            raise ValueError("UCB post-processing of item %s: %s" % (names[j] if names is not None else j, STATUS_TEXT.get(int(st), "status %d" % int(st))))
